// The block-level pre-verify pass ON THE DEVICE (private interface between bccsp_host.cpp, fabgpu_api.hip and block_walk_kernels.hip;
// the C ABI above it is unchanged: fabgpu_csp_block_preverify / _preverify2 of include/fabgpu_bccsp.h take this route for blocks it
// can serve and the host walk for the rest).
//
// What moved: the envelope walk (core/common/validation/msgvalidation.go:258-298, statebased/validator_keylevel.go:246-258 tuples;
// protoutil CheckTxID / GetProposalHash2 inputs), the signature gates of the common DER shape (bccsp/utils/ecdsa.go:43-92), the
// identity -> key lookup (the msp identity cache, msp/cache/cache.go), the submission arrays of the fused launch, the comparison of the
// TxID / proposal-hash digests and the per-transaction flags.  What the host keeps: the outer framing and the serial list of envelope
// starts (block_prepass.h OutlineBlock), the orderers' block-signature tuples with their tail, the identity cache itself (decoding a
// certificate nobody has seen yet), the verdict memo.
//
// Data layout in HBM for one pass (all sized by the counts of the first kernel):
//     block bytes              as uploaded by fabgpu_arena_stage (read-only; tail behind it at tail_base)
//     env_spans   u32[2 n_env]      (offset, length) of every envelope                                   H2D  8 B / tx
//     counts      uint4[n_env]      tuples, prefixes, hash checks, gathered bytes per envelope            device only
//     bases       uint4[n_env]      exclusive prefix sums of the above                                    device only
//     tuples      BlockTuple[n]     44 B records                                                          D2H only when the caller wants spans / memo
//     off2 u32[2n]  pre_idx u32[n]  key_id u32[n]  qx qy r s u8[32 n]  gate_st u8[n]                       device only (the fused launch reads them)
//     pre_off2    u32[2 n_pre]      checks BlockHashCheck[n_chk]  gather_spans u32[6 n_chk]  gather_off u32[n_chk + 1]
//     verdict words, status bytes, digests 32 n, gather digests 32 n_chk                                   digests D2H only when wanted
//     tx_mask u32[n_tx] -> tx_flags u8[n_tx], tx_type u8[n_tx]                                             D2H  2 B / tx
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "block_walk_core.h"

struct fabgpu_ctx;

namespace fab {

// One identity the provider has met: the host's cache entry as the device sees it.
struct DevIdEntry {
    uint64_t hash;        // walk::id_hash_host of the SerializedIdentity bytes
    uint32_t off, len;    // those bytes in the table's byte arena
    int32_t key_id;       // >= 0: a registered comb table (fabgpu_p256_key_register)
    uint32_t p256;        // 1: (qx, qy) is an on-curve P-256 key; 0: identity.Verify needs bccsp/sw (TUPLE_ST_NEEDS_SW)
    uint8_t qx[32], qy[32];
};
static_assert(sizeof(DevIdEntry) == 88, "uploaded as raw bytes");

// summary words of one pass (the status kernel adds up what the gate and identity-decode kernels noted per tuple, one atomic per
// wavefront; one small copy device -> host at the end)
struct WalkSummary {
    uint32_t n_unknown_identity;   // tuples whose identity is not in the device table: the device decodes their certificates itself
    uint32_t n_undecided;          // ... of which it could not decide (a PEM body beyond its buffer): the one case the host repairs
    uint32_t n_outline_differs;    // creator messages that are not the span the host's outline named: this pass does not answer
    uint32_t n_submitted;          // tuples for the device to decide
    uint32_t n_unkeyed_creator;    // submitted creator tuples without a registered comb table (any: the fresh-key kernel serves that launch)
    uint32_t n_unkeyed_other;      // the same for endorsements and block signatures
    uint32_t n_hashed_creator;     // tuples the device hashed and decided, by launch class (status kernel)
    uint32_t n_hashed_other;
    uint32_t n_nym;                // idemix creators whose pseudonym signature went to the nym kernel
    uint32_t n_general_der;        // signatures that took the general DER parser (statistics)
    uint32_t n_learn;              // (identity kernel) identities offered to the provider's cache in learn[] (distinct by table hash)
    uint32_t pad[1];
};
// An idemix MSP the provider knows (GPUCSP::RegisterIdemixMSP): creators serialized under this MSP id sign with pseudonym signatures, which
// the device route verifies itself since round 3 (the gate kernel recognises msp.SerializedIdemixIdentity and unmarshals the
// idemix.NymSignature; the nym kernels of idemix_kernels.hip run beside the ECDSA launches over rows indexed by creator rank).
constexpr uint32_t WALK_IDEMIX_MSPS_MAX = 16;
constexpr uint32_t WALK_IDEMIX_MSPID_MAX = 120;
struct DevIdemixMsp {
    uint32_t len;                  // of the MSP id
    int32_t issuer;                // device issuer id (fabgpu_idemix_issuer_register)
    uint8_t id[WALK_IDEMIX_MSPID_MAX];
};
static_assert(sizeof(DevIdemixMsp) == 128, "uploaded as raw bytes");

// An identity the device decoded and the provider may want in its cache (and, once it has been named often enough, with a comb table):
// a slot found by open addressing from the table hash (eight probes) - a block offers at most WALK_LEARN_SLOTS new identities, whoever
// finds eight other identities in a row shows up again in the next block.
constexpr uint32_t WALK_LEARN_SLOTS = 128;
struct WalkLearn {
    uint64_t tag;                  // 0: empty; else the identity's table hash | 1
    uint32_t off, len;             // the SerializedIdentity bytes in the block (valid once `ready` is set)
    uint32_t ready;                // 1: qx, qy = its P-256 key; 2: a certificate without a P-256 key (identity.Verify needs bccsp/sw)
    uint32_t hits;                 // tuples of this block that named it
    uint8_t qx[32], qy[32];
};
static_assert(sizeof(WalkSummary) == 48 && sizeof(WalkLearn) == 88, "copied as raw bytes");

// what the memo kernels made: entries, key bytes, and whether the keys outgrew the room the caller gave (then there is no memo for this block)
struct WalkMemoTotals {
    uint32_t n, overflow;       // entries written (candidates), 1: the keys did not fit
    uint64_t bytes;             // of keys
    uint32_t live, pad;         // entries that got a slot (hashed and decided)
};
struct WalkTotals {
    uint32_t tuples, prefixes, checks, creators;   // creators: envelopes that yield tuples (each yields exactly one creator tuple, its first)
    uint64_t gather_bytes;
};

// device pointers of one pass (block_walk_kernels.hip launchers; every pointer is device memory)
struct WalkArrays {
    const uint8_t* block = nullptr;
    uint32_t block_len = 0;          // bytes of the marshalled block
    uint32_t arena_len = 0;          // block + tail extent (spans of block-signature tuples reach into the tail)
    const uint32_t* env_spans = nullptr;
    uint32_t n_env = 0;
    uint4* counts = nullptr;
    bccsp::walk::EnvStash* stash = nullptr;   // optional, n_env slots: the count kernel keeps what it found, the emit kernel copies it (block_walk_core.h)
    uint4* bases = nullptr;
    uint32_t* cbase = nullptr;       // per envelope: creator tuples of the envelopes before it
    WalkTotals* totals = nullptr;
    uint8_t* tx_type = nullptr;
    uint8_t* tx_understood = nullptr;
    // sized after the totals are known
    bccsp::BlockTuple* tuples = nullptr;
    uint32_t n_tuples = 0;           // device-emitted + appended block signatures
    uint32_t* pre_off2 = nullptr;
    bccsp::BlockHashCheck* checks = nullptr;
    uint32_t* gather_spans = nullptr;
    uint32_t* gather_off = nullptr;
    const uint32_t* payload_spans = nullptr;   // per envelope, from the host's outline: (start, end) of Envelope.payload
    uint8_t* digest_env = nullptr;       // per envelope: SHA-256 of that payload, started before anything was walked
    uint32_t early_creator_hash = 0;     // the creators' launch reads digests that came from digest_env (the gate kernel cross-checks the spans)
    uint32_t* creator_spans = nullptr;   // (start, end) of every creator tuple's message, in creator order (split submissions hash them early)
    uint32_t* id_idx = nullptr;
    uint32_t* off2 = nullptr;
    uint32_t* pre_idx = nullptr;
    uint32_t* key_id = nullptr;
    uint8_t *qx = nullptr, *qy = nullptr, *r = nullptr, *s = nullptr;
    uint8_t* gate_st = nullptr;
    uint8_t* tflags = nullptr;           // per tuple: what the gate / identity kernels note for the summary (the status kernel adds them up)
    // idemix creators (null / 0: no idemix MSP is registered).  Rows by CREATOR RANK (cbase): six 32-byte columns, issuer, message span;
    // creators that are not idemix keep an all-zero row the nym kernel answers "not decided" for, and nobody reads.
    const DevIdemixMsp* idemix_msps = nullptr;
    uint32_t n_idemix_msps = 0;
    uint8_t* nym_fields = nullptr;       // 6 columns x 32 n_creators: nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce
    uint32_t* nym_issuer = nullptr;      // n_creators
    int32_t* nym_issuer_out = nullptr;   // n_creators: the issuer id of an ACTIVE row, -1 otherwise (the memo binds entries to the issuer)
    uint32_t* nym_spans = nullptr;       // 2 n_creators: (start, end) of the signed message = the envelope payload
    // The nym launch runs over the idemix creators' rows only, through a row -> rank list (walk_nym_pack_kernel): a block's 10 000 creator
    // rows with 2 000 idemix ones among them would otherwise occupy every SIMD beside the ECDSA launches.  nym_slot[rank] = the launch's row
    // for that creator, nym_cap = rows launched.
    uint32_t* nym_slot = nullptr;        // n_creators
    uint32_t nym_cap = 0;
    uint32_t gate_mode = 0;              // walk_gate_kernel: 0 every tuple, 1 the creators' tuples only, 2 the others only
    const uint8_t* nym_status = nullptr; // results of the nym launch, by the LAUNCH's row (nym_cap + 64 bytes)
    WalkLearn* learn = nullptr;          // WALK_LEARN_SLOTS slots, zeroed per pass
    // Row of tuple i in the submission arrays.  Plain: row = i.  Split (a block of 32 769 .. 65 536 tuples): the creator tuples - long
    // messages (the whole envelope payload), no shared prefix - take rows [0, n_creators) and run as a launch of their own with two
    // lanes per signature, next to the endorsements' launch (rows n_creators ..) with one: the chip's 1024 SIMDs hold both.
    uint32_t* row_of = nullptr;
    uint32_t n_dev_tuples = 0;       // tuples the walk emitted (the appended block signatures follow)
    uint32_t n_creators = 0;
    uint32_t split = 0;
    WalkSummary* summary = nullptr;
    // identity table
    const uint32_t* id_slots = nullptr;
    uint32_t id_mask = 0;
    const DevIdEntry* id_entries = nullptr;
    const uint8_t* id_bytes = nullptr;
    uint64_t id_seed = 0;            // the provider's table-hash seed (walk::id_hash_host)
    // results
    const uint64_t* verdict_bits = nullptr;    // of the launch over rows [split ? n_creators : 0, n)
    const uint64_t* verdict_bits_c = nullptr;  // split: of the creators' launch
    // the eight-lane kernels' second phase runs as ONE launch over all rows: bit `row` of this bitmap is the verdict of a class while
    // its flag is set (a class that had to be verified again - keys carried along - goes back to its own bitmap above)
    const uint64_t* verdict_bits_all = nullptr;
    uint32_t all_creators = 0, all_others = 0;
    const uint8_t* dev_status = nullptr;       // by row
    const uint8_t* row_digests = nullptr;      // by row (null: not wanted)
    uint8_t* tuple_digests = nullptr;          // by tuple
    uint32_t* summary_parts = nullptr;         // ceil(n_tuples / 256) rows of WalkSummary's words: the status kernel's workgroups, summed by the finish kernel
    // The block's verdict memo, built on the device (walk_memo_*_kernel) in the layout of bccsp_host.h's BlockMemo: framed keys back to
    // back, their offsets, a status byte per entry and an open-addressed slot table (entry index + 1; the slot hash is GPUCSP::MemoHash).
    uint32_t* memo_slots = nullptr;
    uint32_t memo_mask = 0;
    uint32_t* memo_key_off = nullptr;          // entries + 1
    uint8_t* memo_keys = nullptr;
    uint32_t memo_keys_cap = 0;
    uint8_t* memo_status = nullptr;            // by entry (255: a candidate that was not decided - it has no slot)
    uint8_t* memo_digests = nullptr;           // 32 bytes by entry
    uint32_t* memo_ent = nullptr;              // by tuple: its entry, ~0 = none
    void* memo_tiles = nullptr;                // scratch of the entry scan: 24 bytes per tile of 2 048 tuples (walk_memo_tile_*_kernel)
    // ... and its DIGEST memo (bccsp.Hash of bytes this pass hashed): by entry the two spans of the signed message (prefix offset, length,
    // suffix offset, length - into the block / its tail), and a second slot table of the same size over walk::msg_fingerprint of the
    // message's bytes (entry index + 1).  Every candidate gets a slot, early; a lookup skips entries whose status says "not decided".
    uint32_t* memo_hspans = nullptr;           // 4 per entry (null: no digest memo)
    uint32_t* memo_hslots = nullptr;
    struct WalkMemoTotals* memo_totals = nullptr;
    const uint8_t* issuer_hashes = nullptr;    // 32 bytes per idemix MSP of idemix_msps: ipk.Hash of its issuer (a pseudonym entry is bound to it)
    uint8_t* tuple_qxy = nullptr;              // by tuple, 64 bytes: the key of a P-256 identity, zeros otherwise (null: not wanted)
    uint8_t* tuple_status = nullptr;
    uint8_t* tuple_hashed = nullptr;
    const uint8_t* gather_digests = nullptr;
    uint32_t* tx_mask = nullptr;
    uint8_t* tx_flags = nullptr;
};

// Where the kernels write what the HOST waits for: pinned, host-mapped memory (device pointers to it), each block of results closed by a
// 32-bit flag that takes the pass's sequence number once everything before it is visible to the host.  The host polls the flag instead
// of queueing device-to-host copies and synchronising the stream: a copy command plus a stream wake-up is ~40 us per round trip and
// a pass used to end with six of them (profiles/r02_device_walk_timeline.txt: 652 -> 714 us, and 81 -> 119 us for the totals).
struct WalkHostOut {
    uint32_t* flag = nullptr;             // <- seq when the arrays below are complete
    uint32_t seq = 0;
    uint32_t* done = nullptr;             // DEVICE memory: workgroups of the last kernel that have finished (zeroed per pass)
    WalkSummary* summary = nullptr;
    WalkLearn* learn = nullptr;           // WALK_LEARN_SLOTS
    uint8_t *tx_flags = nullptr, *tx_type = nullptr, *tx_understood = nullptr;   // n_env each
    uint8_t *tuple_status = nullptr, *tuple_hashed = nullptr;                    // n_tuples each
    uint32_t* id_idx = nullptr;                                                  // n_tuples
    WalkMemoTotals* memo_totals = nullptr;                                       // optional
};
// counts, tx_type, tx_understood; then the scan, which also writes the totals to host_totals and then seq to host_flag (host-mapped)
hipError_t launch_walk_count(const WalkArrays& a, WalkTotals* host_totals, uint32_t* host_flag, uint32_t seq, hipStream_t st);
hipError_t launch_walk_emit(const WalkArrays& a, const WalkTotals& t, hipStream_t st);   // tuples, prefixes, checks, gather spans / offsets
hipError_t launch_walk_gate(const WalkArrays& a, hipStream_t st);                        // identity lookup / certificate decode + gates + submission arrays
// TEST HOOK: the device's identity decoder over n SerializedIdentity byte strings (spans = (start, end) pairs into arena) -> code
// (0 P-256 key, 1 not such an identity, 2 undecided), key (64 bytes each, zero unless code == 0)
hipError_t launch_walk_idfix_probe(uint32_t n, const void* arena, const void* spans, void* code, void* key, hipStream_t st);
// the idemix creators among [0, n_creators), counted off: nym_slot[rank] = its row in the nym launch, gather[row] = rank for row < cap
hipError_t launch_walk_nym_pack(const WalkArrays& a, uint32_t* gather, uint32_t cap, hipStream_t st);
// the verdict memo of the pass: keys, entry indices and offsets as soon as the gates are through; digests, statuses and slots behind the status kernel
hipError_t launch_walk_memo_early(const WalkArrays& a, hipStream_t st, hipEvent_t scanned = nullptr);   // scanned: recorded behind the scan, in front of the key bytes
hipError_t launch_walk_memo_index(const WalkArrays& a, hipStream_t st);   // the digest memo's index (needs the scan; its own stream)
hipError_t launch_walk_memo_late(const WalkArrays& a, hipStream_t st);
hipError_t launch_walk_status_checks(const WalkArrays& a, uint32_t n_checks, hipStream_t st);   // tuple statuses + digest comparisons (one launch)
// per-transaction flags and everything the host reads, written to host-mapped memory; the last workgroup raises h.flag
hipError_t launch_walk_finish(const WalkArrays& a, const WalkHostOut& h, hipStream_t st);
// the two above as ONE single-workgroup launch, for a block of a few hundred transactions whose caller wants no arrays copied between them
bool walk_small_finish_fits(const WalkArrays& a, uint32_t n_checks);
hipError_t launch_walk_status_finish_small(const WalkArrays& a, uint32_t n_checks, const WalkHostOut& h, hipStream_t st);
hipError_t launch_walk_creator_digests(const WalkArrays& a, void* row_digests, hipStream_t st);   // digest_env -> the creators' digest rows
// TEST HOOK: the wavefront form of the signature gate over n signatures (device pointers; spans = (start, end) pairs into arena)
hipError_t launch_walk_gate_probe(uint32_t n, const void* arena, const void* spans, void* code, void* r, void* s, hipStream_t st);    // statuses, digest comparisons, per-transaction flags

// ---- host side (fabgpu_api.hip) ----
// Replace the device's identity table (entries + their bytes); the table is rebuilt by the provider whenever its cache changes.
// Entries are placed in the order given (put the ones that matter most first); one that would land WALK_ID_PROBE_MAX or more slots from
// its home is left out (the device then decodes that identity itself: slower, never wrong).  `seed`: what the entries' hashes were made with.
int walk_idtab_set(fabgpu_ctx* ctx, uint32_t n, const DevIdEntry* entries, const uint8_t* bytes, size_t nbytes, uint64_t seed);

constexpr int WALK_DECLINED = 100;   // not an error: this block is for the host walk (why: WalkRequest::declined_why)

struct WalkCounts {
    uint32_t n_tx = 0, n_tuples = 0, n_prefixes = 0, n_checks = 0, n_creators = 0;
};
// host arrays the pass fills; asked for through WalkRequest::sizes once the counts are known (null = not wanted)
struct WalkOut {
    uint8_t* tx_flags = nullptr;          // n_tx
    uint8_t* tx_type = nullptr;           // n_tx
    uint8_t* tx_understood = nullptr;     // n_tx
    uint8_t* tuple_status = nullptr;      // n_tuples
    uint8_t* tuple_hashed = nullptr;      // n_tuples
    bccsp::BlockTuple* tuples = nullptr;  // n_tuples
    uint32_t* id_idx = nullptr;           // n_tuples: index into the entries of walk_idtab_set; 0xFFFFFFFE: decoded by the device, 0xFFFFFFFF: none
    uint8_t* tuple_qxy = nullptr;         // 64 n_tuples: key of a P-256 identity, zeros otherwise
    int32_t* nym_issuer = nullptr;        // n_creators (WalkCounts::n_creators): issuer id of an idemix creator's row by creator rank, -1 otherwise
    uint8_t* tuple_digest = nullptr;      // 32 n_tuples
    bccsp::Span* prefixes = nullptr;      // n_prefixes            (tests)
    bccsp::BlockHashCheck* checks = nullptr;   // n_checks         (tests)
    // The block's verdict memo as the DEVICE builds it (round 3), into PINNED host memory of the caller (walk_pinned_alloc), in the layout
    // GPUCSP::MemoLookup reads; all null = not wanted.  Afterwards WalkRequest::memo_n entries are in place (0: none - nothing to remember,
    // or the keys did not fit memo_keys_cap: that block simply has no memo, its signatures are verified by bccsp/sw).
    uint32_t* memo_slots = nullptr;       // memo_slot_cap entries, a power of two >= 2 n_tuples
    uint32_t memo_slot_cap = 0;
    uint32_t* memo_key_off = nullptr;     // n_tuples + 1
    uint8_t* memo_keys = nullptr;         // memo_keys_cap bytes (< 2^32): copied ahead; keys that need more arrive through WalkRequest::memo_grow
    size_t memo_keys_cap = 0;
    uint8_t* memo_status = nullptr;       // n_tuples
    uint8_t* memo_digests = nullptr;      // 32 n_tuples: the digest of entry e (the keys end with the digest's length field)
    uint32_t* memo_hspans = nullptr;      // optional, with memo_hslots: 4 n_tuples - the digest memo's message spans by entry ...
    uint32_t* memo_hslots = nullptr;      // ... and its slot table, memo_slot_cap entries
};
struct WalkRequest {
    uint64_t stage_token = 0;             // the block, uploaded with fabgpu_arena_stage
    const uint32_t* payload_spans = nullptr;   // host, optional: OutlineBlock's payload spans, 2 per envelope (the creators' hashes start on them)
    size_t block_len = 0;
    const uint32_t* env_spans = nullptr;  // host
    uint32_t n_env = 0;
    // host, optional: what walk_count_kernel would find - 4 per envelope (tuples, prefixes, hash checks, gathered bytes clamped to 2^32 - 1)
    // and the two per-envelope bytes - counted by the caller with the same walk_envelope<CountEmitter> while the block was still on its
    // way up.  The pass then starts at the emit kernel: no count kernel, no scan, no wait for totals.  (The emit kernel checks every
    // envelope's counts against its own walk: WalkSummary::n_outline_differs.)
    const uint32_t* host_counts = nullptr;
    const uint8_t* host_tx_type = nullptr;
    const uint8_t* host_tx_understood = nullptr;
    const bccsp::BlockTuple* block_sigs = nullptr;   // appended behind the device's tuples
    uint32_t n_block_sigs = 0;
    const uint8_t* tail = nullptr;
    uint32_t tail_base = 0, tail_len = 0;
    const DevIdemixMsp* idemix_msps = nullptr;   // host: the idemix MSPs whose creators the pass verifies (at most WALK_IDEMIX_MSPS_MAX)
    uint32_t n_idemix_msps = 0;
    const uint8_t* idemix_issuer_hashes = nullptr;   // host, 32 bytes per entry of idemix_msps: ipk.Hash (memo entries of pseudonym signatures)
    bool walk_only = false;               // stop after the walk (tests: the device walker against the host walker)
    void* user = nullptr;
    bool (*sizes)(void* user, const WalkCounts& c, WalkOut& out) = nullptr;   // false: the caller has no room (FABGPU_ETOOBIG)
    // the memo's keys need `bytes` > WalkOut::memo_keys_cap: pinned room for all of them (they are copied there whole), or null: no memo
    uint8_t* (*memo_grow)(void* user, size_t bytes) = nullptr;
    // out
    WalkSummary summary = {};
    bool keyed_creators = false, keyed_others = false;   // which launch classes ran on registered comb tables
    uint32_t relaunched = 0;              // launches repeated because the prediction "everybody is registered" did not hold
    WalkLearn* learn_out = nullptr;       // host, optional: WALK_LEARN_SLOTS records (tag == 0 or ready == 0: empty)
    uint32_t memo_n = 0;                  // entries the device wrote into WalkOut::memo_* (0: none) ...
    uint32_t memo_live = 0;               // ... of which so many have a slot (were hashed and decided)
    uint64_t memo_bytes = 0;
    const char* declined_why = "";
    double ms_walk = 0, ms_gate = 0, ms_verify = 0;
};
// FABGPU_OK, WALK_DECLINED, or a negative FABGPU_E*
int walk_block_pass(fabgpu_ctx* ctx, WalkRequest& rq);
// The block's bytes in HOST memory the context owns, kept after the upload (fabgpu_arena_stage copies a block through pinned staging
// anyway: with `keep` that staging buffer comes out of a small pool and stays with the caller until host_copy_release).  What the
// digest memo compares a bccsp.Hash caller's bytes with: nothing of the CALLER's block is retained past the call (cgo pointer rules).
// p == nullptr after the call: the pool had no buffer to spare (the upload itself went through; the block simply has no digest memo).
struct HostCopy {
    fabgpu_ctx* ctx = nullptr;
    int idx = -1;
    const uint8_t* p = nullptr;
    size_t len = 0;
};
int arena_stage_keep(fabgpu_ctx* ctx, const void* arena, size_t len, uint64_t* token, HostCopy* keep);
void host_copy_release(HostCopy* c);
// at most `blocks` kept copies per context (default 8; each as large as its block)
void host_copy_limit(fabgpu_ctx* ctx, uint32_t blocks);
void host_copy_preallocate(fabgpu_ctx* ctx, size_t block_bytes, uint32_t n);
void host_copy_stats(fabgpu_ctx* ctx, uint64_t* held, uint64_t* bytes_held, uint64_t* refused);
// allocate now what `slots` overlapping passes over blocks of up to these sizes will need on this device (best effort)
int walk_preallocate(fabgpu_ctx* ctx, size_t block_bytes, uint32_t n_tx, uint32_t n_tuples, int slots);
double walk_warm_copies(fabgpu_ctx* ctx, void* const* pinned, const size_t* bytes, int n);
// a registered key's comb table built apart from its registration (fabgpu_api.hip)
size_t key_table_words();
bool key_table_build(const uint8_t* qx32, const uint8_t* qy32, int32_t* out);
// n keys (qxy: n x 64 bytes X || Y, all on the curve) on one context, their comb tables built on the device (keytab_kernels.hip)
int key_register_batch(fabgpu_ctx* ctx, int n, const uint8_t* qxy, uint32_t* key_ids);
int64_t gtab_compare_with_host(fabgpu_ctx* ctx);   // TEST HOOK support: first differing word of the generator comb vs the host builder's, -1 = identical
int64_t key_tables16_check(fabgpu_ctx* ctx, uint32_t key_id);   // TEST HOOK support: waits for the 16-bit key tables, cross-checks one against the key's 8-bit table
int key_table_copy(fabgpu_ctx* ctx, uint32_t key_id, int32_t* out);   // TEST HOOK support: a registered key's device table, to the host
int key_register_many_prebuilt(fabgpu_ctx* const* ctxs, int n, const uint8_t* qx32, const uint8_t* qy32, const int32_t* table, uint32_t* key_ids);
// the duration of the last timed launch of a FABGPU_FLAG_TIME_KERNELS context (ms; < 0: none) - read by the test-hook library
float ctx_last_kernel_ms(fabgpu_ctx* ctx);
// TEST HOOK: the idemix four-lane form orders its side launch BEHIND the commitment launch while this is on
void ctx_test_nym_side_after(fabgpu_ctx* ctx, bool on);
// pinned host memory for WalkOut::memo_* (hipHostMalloc / hipHostFree; nullptr when there is none to be had)
void* walk_pinned_alloc(fabgpu_ctx* ctx, size_t bytes);
void walk_pinned_free(fabgpu_ctx* ctx, void* p);
// TEST HOOK: the device's (wavefront) signature gate over n signatures = arena[spans[2i], spans[2i+1]) in host memory -> code (as
// walk::gate_sig_any), r, s (32 bytes each; zero unless code == GATE_SUBMIT)
int walk_gate_probe(fabgpu_ctx* ctx, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* r, uint8_t* s);
// TEST HOOK: the device's identity decoder over n identities in host memory -> code, key (64 bytes each)
int walk_idfix_probe(fabgpu_ctx* ctx, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* key);

}  // namespace fab
