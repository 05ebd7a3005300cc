// Block-level pre-verify pass: walks a marshalled common.Block once, extracts every
// (identity, message, signature) the validators will ask bccsp to verify, and verifies them in ONE fused launch.
//
//   creator signature      identity = SignatureHeader.creator, message = Envelope.payload, signature = Envelope.signature
//                          (core/common/validation/msgvalidation.go:258-298, checkSignatureFromCreator :26-64)
//   endorsement signature  identity = Endorsement.endorser, message = proposal_response_payload || Endorsement.endorser,
//                          signature = Endorsement.signature   (core/common/validation/statebased/validator_keylevel.go:246-258)
//
// The block buffer itself is the message arena (FABGPU_IDB_SPANS); each action's proposal_response_payload is a shared
// prefix hashed once; every identity seen is imported once (x509 -> P-256 point -> fabgpu_p256_key_register: the msp
// identity cache + BCCSP.KeyImport of the reference, msp/cache/cache.go, msp/mspimpl.go:408-421) so that the block runs on
// the keyed kernels.  What stays with the Go validators: identity validation (chain, revocation, OUs), endorsement policy
// evaluation, MVCC.  This pass only answers "would identity.Verify return nil?" for every tuple and seeds the verdict memo.
//
// Wire format: protobuf field numbers of github.com/hyperledger/fabric-protos-go (common/common.proto, peer/transaction.proto,
// peer/proposal_response.proto, msp/identities.proto) - not in the reference tree; the outer layers are pinned by the
// reference's own block fixtures (tests/test_block_prepass.py).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

#include "block_walk_core.h"

namespace fab {
namespace bccsp {

// TUPLE_BLOCK_SIG: an orderer's signature over the block (BlockMetadataIndex_SIGNATURES), the SignedData MCS.VerifyBlock hands to
// the BlockValidation policy (internal/peer/gossip/mcs.go:166-193):
//     identity = SignatureHeader.creator of MetadataSignature.signature_header
//     message  = Metadata.value || MetadataSignature.signature_header || protoutil.BlockHeaderBytes(block.Header)
// BlockHeaderBytes is the ASN.1 DER of {Number INTEGER, PreviousHash OCTET STRING, DataHash OCTET STRING}
// (protoutil/blockutils.go:38-58) - bytes that are NOT in the marshalled block.  The walker therefore writes each such message
// into ParsedBlock::tail, and the tuple's suffix span addresses it at offset tail_base + k of a VIRTUAL arena
// block || zero padding up to tail_base || tail  (fabgpu_identity_batch.tail).  tx = BLOCK_LEVEL_TX for these tuples.
// (Span, BlockTuple, BlockHashCheck, TUPLE_* and HASH_*: block_walk_core.h - shared with the device walker.)
// per-tuple outcome: 0..4 = the device status codes of include/fabgpu.h, plus
enum : uint8_t {
    TUPLE_ST_BAD_DER = 5,        // UnmarshalECDSASignature fails / r,s <= 0: identity.Verify returns an error
    TUPLE_ST_NEEDS_SW = 6,       // identity is not a PEM x509 certificate with a P-256 key (idemix, other curves): bccsp/sw decides
    TUPLE_ST_EMPTY_SIG = 7,      // empty signature
    TUPLE_ST_SKIPPED = 8,        // the caller asked the pass not to verify this kind of tuple (block signatures)
};
// per-transaction summary
enum : uint8_t {
    TX_ALL_SIGNATURES_VALID = 0,
    TX_BAD_CREATOR_SIGNATURE = 1,   // TxValidationCode_BAD_CREATOR_SIGNATURE territory
    TX_BAD_ENDORSEMENT = 2,         // at least one endorsement signature does not verify
    TX_NOT_UNDERSTOOD = 3,          // envelope / payload / transaction did not parse as expected: left to the Go validators
    TX_NEEDS_SW = 4,                // some identity must be verified by bccsp/sw
    TX_BAD_TXID = 5,                // ChannelHeader.tx_id != hex(SHA-256(nonce || creator)): TxValidationCode_BAD_PROPOSAL_TXID
    TX_BAD_PROPOSAL_HASH = 6,       // an action's SHA-256(channel header || action header || proposal payload) != prp.proposal_hash
};
// The two other SHA-256 computations ValidateTransaction makes per endorser transaction:
//   HASH_TXID           protoutil.CheckTxID (protoutil/proputils.go:357-375, called at core/common/validation/msgvalidation.go:288):
//                       SHA-256(SignatureHeader.nonce || SignatureHeader.creator), compared with ChannelHeader.tx_id as lowercase hex
//   HASH_PROPOSAL       protoutil.GetProposalHash2 (protoutil/txutils.go:431-447, called at msgvalidation.go:233-241), per action:
//                       SHA-256(Header.channel_header || TransactionAction.header || ChaincodeActionPayload.chaincode_proposal_payload),
//                       compared with ProposalResponsePayload.proposal_hash

struct ParsedBlock {
    uint32_t n_tx = 0;
    std::vector<uint8_t> tx_type;         // ChannelHeader.type per envelope (-> 255 if not parsed)
    std::vector<uint8_t> tx_understood;
    std::vector<Span> prefixes;
    std::vector<BlockTuple> tuples;
    std::vector<BlockHashCheck> hash_checks;   // endorser transactions only
    std::string first_channel_id;         // of envelope 0 (fixture pin)
    // block level (MCS.VerifyBlock): header fields, the orderer signature messages (see TUPLE_BLOCK_SIG)
    bool has_header = false;
    uint64_t number = 0;
    Span previous_hash, data_hash;        // BlockHeader{2 previous_hash, 3 data_hash}
    Span data;                            // the BlockData message: BlockDataHash = SHA-256 of its concatenated entries
    uint32_t n_block_sigs = 0;
    bool block_sigs_understood = false;   // metadata[SIGNATURES] parsed (possibly to zero signatures)
    uint32_t tail_base = 0;
    std::vector<uint8_t> tail;
    // A ParsedBlock that is handed to ParseBlock again keeps its storage (and that of the per-worker parts below): a provider
    // that parses block after block does not allocate - and page-fault in - a few MB per block.
    std::vector<std::unique_ptr<ParsedBlock>> parts;   // scratch of the threaded walk: one per chunk of envelopes
    void reset() {
        n_tx = 0;
        tx_type.clear(); tx_understood.clear(); prefixes.clear(); tuples.clear(); hash_checks.clear(); first_channel_id.clear();
        has_header = false; number = 0; previous_hash = Span(); data_hash = Span(); data = Span();
        n_block_sigs = 0; block_sigs_understood = false; tail_base = 0; tail.clear();
    }
};

// Pure parsing (no device): false only if the outer Block / BlockData framing is broken.
// BlockData of 1 MiB and more is walked on up to max_threads (<= 16) worker threads while the calling thread lists the envelopes.
bool ParseBlock(const uint8_t* block, size_t len, ParsedBlock& out, int max_threads = 8);
// The host's share of a DEVICE-side walk (block_walk_kernels.hip): outer framing, (offset, length) of every envelope in env_spans, and
// the block-level fields of `out` (header, data span, tail, n_tx = envelopes listed) with the orderers' signature tuples in block_sigs.
// out.tuples / prefixes / hash_checks / tx_type stay empty: the device fills its own copies.  false: as ParseBlock.
// payload_spans (optional): (start, end) of every envelope's Envelope.payload - what its creator signed - so that the device can start
// hashing it before it has walked anything ((0, 0) where the envelope does not yield one).
bool OutlineBlock(const uint8_t* block, size_t len, ParsedBlock& out, std::vector<uint32_t>& env_spans, std::vector<BlockTuple>& block_sigs,
                  std::vector<uint32_t>* payload_spans = nullptr);
// worker threads the pass gives the walk: 8, or FABGPU_PASS_WALK_THREADS (experiments)
int WalkThreads();
// SerializedIdentity{mspid, id_bytes = PEM x509} -> uncompressed P-256 point.  false: not such an identity.
bool IdentityToP256(const uint8_t* ident, size_t len, uint8_t qx[32], uint8_t qy[32]);
// DER x509 certificate -> P-256 SubjectPublicKeyInfo point (exposed for tests against the reference's certificate fixtures)
bool CertDerToP256(const uint8_t* der, size_t len, uint8_t qx[32], uint8_t qy[32]);
bool PemToDer(const uint8_t* pem, size_t len, std::vector<uint8_t>& der);
// protoutil.BlockHeaderBytes (protoutil/blockutils.go:38-58): DER of SEQUENCE{INTEGER number, OCTET STRING previous_hash, OCTET STRING data_hash}
void BlockHeaderBytes(uint64_t number, const uint8_t* prev, size_t prev_len, const uint8_t* data_hash, size_t dh_len, std::vector<uint8_t>& out);
// does the 32-byte digest equal what the block says (hex string for HASH_TXID, raw bytes for HASH_PROPOSAL)?
bool HashCheckMatches(const uint8_t* block, const BlockHashCheck& hc, const uint8_t digest[32]);
// SerializedIdentity{mspid, id_bytes = msp.SerializedIdemixIdentity{1 nym_x, 2 nym_y, 3 ou, 4 role, 5 proof}} (what
// idemixidentity.Serialize writes, msp/idemixmsp.go:605-640) -> MSP id and the 32-byte pseudonym coordinates.
// false: not such an identity (or coordinates of another size: those stay with bccsp/idemix).
bool IdentityToIdemixNym(const uint8_t* ident, size_t len, std::string& mspid, uint8_t nx[32], uint8_t ny[32]);

}  // namespace bccsp
}  // namespace fab
