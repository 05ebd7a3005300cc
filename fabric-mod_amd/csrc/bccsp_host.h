// C++ host mirror of the reference's BCCSP surface for the block-validation signature path.
// Same names, argument meaning and error text as bccsp/bccsp.go:90-134 (Hash / Verify / KeyImport),
// bccsp/utils/ecdsa.go and msp/identities.go:169-196; every verdict comes from the GPU through
// include/fabgpu.h.  The Go provider shown in INTEGRATION.md is the production binding; this layer is what
// the parity tests drive (Go is not installed in the build image).
#pragma once
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

#include <map>
#include <mutex>
#include <thread>

#include "../../include/fabgpu.h"
#include "block_prepass.h"

namespace fab {
namespace bccsp {

struct Error {  // Go `error`: empty == nil
    std::string msg;
    bool nil = true;
    Error() {}
    explicit Error(const std::string& m) : msg(m), nil(false) {}
    bool ok() const { return nil; }
};

struct BigInt {  // *big.Int as produced by encoding/asn1: two's complement, minimal
    std::vector<uint8_t> twos;
    bool negative = false;
    bool is_zero() const;
    int sign() const;
    std::vector<uint8_t> magnitude() const;
    std::string decimal() const;
    bool fits256() const;
    void to_be32(uint8_t* out) const;
};

Error UnmarshalECDSASignature(const uint8_t* raw, size_t len, BigInt& R, BigInt& S);  // bccsp/utils/ecdsa.go:43-67
bool IsLowS(const BigInt& S);                                                          // bccsp/utils/ecdsa.go:84-92
bool PublicKeyOnCurve(const uint8_t* qx32, const uint8_t* qy32);
void HashToInt(const uint8_t* digest, size_t len, uint8_t* e32);
extern const char* HALF_ORDER_DECIMAL;

struct ECDSAPublicKey {  // bccsp/sw/ecdsakey.go:72-117 (X, Y only)
    uint8_t x[32], y[32];
    bool on_curve = false;
};
struct HashOpts {  // bccsp/hashopts.go:20-70
    std::string algorithm;  // "SHA256" is the only family on this path (msp/identities.go:216-224)
};
struct VerifyItem {
    const ECDSAPublicKey* key;
    const uint8_t* sig;
    size_t siglen;
    const uint8_t* digest;
    size_t dlen;
};
struct VerifyResult {  // (valid bool, err error)
    bool valid = false;
    Error err;
    bool needs_sw = false;        // tuple the GPU provider refuses to decide (off-curve key)
    bool infrastructure = false;  // device failure: caller falls back to bccsp/sw
};
struct IdentityItem {
    const ECDSAPublicKey* key;
    const uint8_t* msg;
    size_t msglen;
    const uint8_t* sig;
    size_t siglen;
};

struct BlockVerdicts {
    uint32_t n_tx = 0;
    std::vector<uint8_t> tx_flags;        // TX_* per transaction
    std::vector<uint8_t> tx_type;         // ChannelHeader.type (255: envelope not parsed)
    std::vector<uint32_t> tuple_tx;       // per tuple: owning transaction
    std::vector<uint8_t> tuple_kind;      // TUPLE_CREATOR / TUPLE_ENDORSEMENT
    std::vector<uint8_t> tuple_status;    // device status 0..4 or TUPLE_ST_*
    uint32_t distinct_identities = 0;     // identities of this block that were not in the cache yet
    double ms_gates = 0, ms_upload_wait = 0, ms_device = 0;   // where the pass spent its time (host clock)
};

class GPUCSP {
   public:
    static Error New(const fabgpu_cfg* cfg, std::unique_ptr<GPUCSP>& out);
    ~GPUCSP();
    // device_table: also build the key's comb table on the device (fabgpu_p256_key_register) - what the provider does for a
    // key imported through BCCSP.KeyImport; false for keys merely unmarshalled from a flat batch.
    Error KeyImport(const uint8_t* qx32, const uint8_t* qy32, ECDSAPublicKey& out, bool device_table = false) const;
    Error Hash(const uint8_t* msg, size_t len, const HashOpts* opts, std::vector<uint8_t>& digest) const;
    VerifyResult Verify(const ECDSAPublicKey* k, const uint8_t* sig, size_t siglen, const uint8_t* digest, size_t dlen) const;
    Error VerifyBatch(const std::vector<VerifyItem>& items, std::vector<VerifyResult>& results) const;
    Error IdentityVerifyBatch(const std::vector<IdentityItem>& items, std::vector<std::string>& out) const;
    // Block-level pre-verify pass (block_prepass.h): one fused launch for every creator / endorsement signature of the block.
    Error PreVerifyBlock(const uint8_t* block, size_t len, BlockVerdicts& out) const;
    struct BlockUpload;
    // up: an upload of the block started ahead (StartBlockUpload); joined right before the submission.  nullptr: the block
    // travels with the submission.
    Error PreVerifyParsed(const uint8_t* block, const ParsedBlock& parsed, BlockVerdicts& out, BlockUpload* up = nullptr) const;
    // Starts the upload of a block on a helper thread (blocks of 4 MiB and more) so that it travels while the caller parses
    // and gates; join() returns the token for PreVerifyParsed (0 if nothing was staged).
    struct BlockUpload {
        std::thread th;
        uint64_t token = 0;
        int rc = -1;
        uint64_t join() {
            if (th.joinable()) th.join();
            return rc == 0 ? token : 0;
        }
        ~BlockUpload() { if (th.joinable()) th.join(); }
    };
    void StartBlockUpload(BlockUpload& up, const uint8_t* block, size_t len) const;
    // An idemix MSP of the channel (msp/idemixmsp.go:99-173 Setup): its creators' pseudonym signatures are then verified by the
    // pre-verify pass too.  ipk_raw: marshalled idemix.IssuerPublicKey.  Returns the device issuer id, or -1 (not accelerated).
    int64_t RegisterIdemixMSP(const std::string& mspid, const uint8_t* ipk_raw, size_t len) const;
    fabgpu_ctx* ctx() const { return ctx_; }

   private:
    explicit GPUCSP(fabgpu_ctx* c) : ctx_(c) {}
    fabgpu_ctx* ctx_;
    // identity cache of the pre-verify pass (msp/cache/cache.go): SerializedIdentity bytes -> P-256 key + device key id
    struct CachedIdentity {
        bool p256 = false;
        uint8_t qx[32], qy[32];
        int64_t key_id = -1;
    };
    mutable std::mutex idmu_;
    mutable std::map<std::string, CachedIdentity> idcache_;
    mutable std::map<std::string, int64_t> idemix_msps_;   // mspid -> device issuer id (guarded by idmu_)
    // scratch of the pre-verify pass, reused from block to block (guarded by pass_mu_)
    struct PassScratch {
        struct Gated {
            uint8_t qx[32], qy[32], r[32], s[32];   // idemix tuple: qx, qy = pseudonym; r, s = ProofC, ProofSSk
            uint8_t srn[32], nonce[32];              // idemix only: ProofSRNym, Nonce
            int64_t key_id;                          // idemix: issuer id
            bool submit, nym;
        };
        std::vector<Gated> gt;
        std::vector<uint32_t> sub, ids, off, pre_idx;
        std::vector<uint8_t> qx, qy, r, s;
    };
    mutable std::mutex pass_mu_;
    mutable PassScratch ps_;
};

}  // namespace bccsp
}  // namespace fab
