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

#include <deque>
#include <list>
#include <map>
#include <array>
#include <atomic>
#include <mutex>

#include "reader_lock.h"
#include <shared_mutex>
#include <thread>
#include <unordered_map>

#include "../../include/fabgpu.h"
#include "block_prepass.h"
#include "block_walk_dev.h"
#include "coalescer.h"

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
    double ms_gates = 0, ms_upload_wait = 0, ms_device = 0, ms_nym = 0, ms_memo = 0;   // where the pass spent its time (host clock)
    double ms_post = 0;                   // device route: bookkeeping behind the device phase (cache hits, learned identities, comb tables, memo)
    // What a consumer needs to attach each verdict to the BYTES it was computed over (never to a position):
    std::vector<uint8_t> tuple_digest;    // 32 per tuple: SHA-256 of the signed message as the fused kernel computed it; zero unless the
                                          // device hashed the message (tuple_hashed[i] == 1)
    std::vector<uint8_t> tuple_hashed;
    std::vector<uint8_t> tuple_qxy;       // 64 per tuple: the P-256 key the identity carries (zero: none / not P-256)
    std::vector<int64_t> tuple_nym_issuer;   // empty, or per tuple: the device issuer id an idemix pseudonym signature was verified under (-1: not one)
    uint32_t n_block_sigs = 0;            // TUPLE_BLOCK_SIG tuples (the last ones)
    uint8_t block_sigs_understood = 0;
    uint32_t memo_seeded = 0;             // entries this pass added to the verdict memo
    size_t n_keyed = 0;                   // submitted tuples that went through per-key device tables (the rest carried their keys)
    uint32_t n_device_decoded = 0;        // device route: tuples whose identity was not in the device's table (certificate decoded on the device)
};

// Options of one pass.
struct PassOptions {
    bool want_digests = false;            // fill BlockVerdicts::tuple_digest (costs 32 B per tuple of D2H)
    bool seed_memo = false;               // remember (key, signature, digest) -> status under `block_seq` (implies want_digests)
    bool block_sigs = true;               // verify the orderers' block signatures too (MCS.VerifyBlock's SignedData)
    uint64_t block_seq = 0;
    size_t tail_cap = (size_t)-1;         // room the caller has for the block-signature tail (device route: checked before anything is launched)
};

// What the `GPU:` section of the BCCSP configuration carries (go/bccsp/factory/gpufactory.go GPUOpts -> include/fabgpu_bccsp.h
// fabgpu_csp_opts; pattern: bccsp/pkcs11/conf.go:70-84).  The reference has ONE process-global BCCSP (bccsp/factory/factory.go:41-55,
// handed to every channel's validator at core/peer/peer.go:337-355), so one provider owns every device of the node.
struct ProviderOptions {
    std::vector<int32_t> devices;      // HIP ordinals, one device context each; an ordinal may repeat (several contexts on one GPU).
                                       // empty: every visible device
    uint32_t ctx_flags = 0;            // FABGPU_FLAG_* for every context
    uint32_t concurrent_passes = 0;    // per device: what that many overlapping block passes need (staging slots, pinned memo tables,
                                       // per-pass scratch) is allocated when the provider is made; 0 = when passes first overlap
    size_t expect_block_bytes = 0;     // sizes the pre-allocation (0: 64 MiB)
    uint32_t expect_tuples = 0;        // (0: 65 536)
    // switches that used to be environment variables.  0 = the default, > 0 on, < 0 off - a zeroed struct is all defaults.
    int64_t pass_stage_min_bytes = 0;  // > 0: blocks of at least this many bytes are uploaded ahead of the pass (default: every block, with the device walk)
    int pass_device_walk = 0;          // < 0: every block takes the host walk (default on)
    int pass_device_memo = 0;          // < 0: the verdict memo is seeded on the host (SeedMemo) also on the device route (default on)
    int pass_host_counts = 0;          // > 0: count the envelopes' tuples on the host while the block travels (default off)
    int pass_skip_hash_checks = 0;     // > 0: no TxID / proposal-hash digests (A/B timing only; default off)
    int pass_timing = 0;               // > 0: stage breakdown of every pass on stderr (default off)
    int pass_hash_memo = 0;            // < 0: memo-seeding passes keep no host copy of their block and bccsp.Hash is never answered from the
                                       // digest memo (default on: HashLookup)
    uint32_t hash_memo_blocks = 0;     // per device: host copies of blocks kept at a time for the digest memo (0: 8; at most 64)
};
constexpr int kMaxProviderDevices = 64;   // contexts per provider (8 GPUs x up to 8 contexts each)

class GPUCSP {
   public:
    static Error New(const fabgpu_cfg* cfg, std::unique_ptr<GPUCSP>& out);              // one context on cfg->device
    static Error New(const ProviderOptions& opts, std::unique_ptr<GPUCSP>& out);        // one context per entry of opts.devices
    ~GPUCSP();
    // ---- the device pool ----
    int n_devices() const { return (int)devs_.size(); }
    fabgpu_ctx* ctx_of(int d) const { return devs_[(size_t)d]->ctx; }
    int device_ordinal(int d) const { return devs_[(size_t)d]->ordinal; }
    // Which context a block pass named `block_seq` runs on: the one with the fewest passes in flight, ties broken round the ring
    // starting at block_seq mod G (pass_route.h) - consecutive blocks and the channels of a peer spread over the node.
    int RouteBlock(uint64_t block_seq) const;
    // passes[d] = block passes context d has served since construction (n_devices() entries)
    void PassesPerDevice(uint64_t* passes) const;
    // options that may change while the provider lives (tests, A/B runs); returns the previous value, INT64_MIN for an unknown name
    int64_t SetOption(const std::string& name, int64_t value) const;
    int64_t GetOption(const std::string& name) const;
    // device_table: also build the key's comb table on the device (fabgpu_p256_key_register) - what the provider does for a
    // key imported through BCCSP.KeyImport; false for keys merely unmarshalled from a flat batch.
    Error KeyImport(const uint8_t* qx32, const uint8_t* qy32, ECDSAPublicKey& out, bool device_table = false) const;
    Error Hash(const uint8_t* msg, size_t len, const HashOpts* opts, std::vector<uint8_t>& digest) const;
    // bccsp.Hash(msg, &bccsp.SHA256Opts{}) for bytes a memo-seeding pass has ALREADY hashed on the device - the `digest = Hash(msg)` half of
    // identity.Verify (msp/identities.go:173-181 -> bccsp/sw/impl.go:177-194), which the unchanged validators call once per signature.
    // The pass keeps the block's bytes in host memory of its own until the block is evicted; each signed message is indexed by a
    // fingerprint of a few sampled bytes (slot choice only), and the stored digest is handed out ONLY when every byte of `msg` equals the
    // bytes the device hashed (memcmp, piecewise for prp || endorser).  0: hit, digest32 filled.  1: miss - the caller hashes on the CPU
    // (always a correct answer).  Never an infrastructure error.
    int HashLookup(const uint8_t* msg, size_t len, uint8_t* digest32) const;
    // digest memo counters: lookups answered / left to the CPU, host copies of blocks held right now and their bytes, passes that
    // wanted a copy and found the pool exhausted
    void HashMemoStats(uint64_t* hits, uint64_t* misses, uint64_t* blocks_held, uint64_t* bytes_held, uint64_t* refused) const;
    VerifyResult Verify(const ECDSAPublicKey* k, const uint8_t* sig, size_t siglen, const uint8_t* digest, size_t dlen) const;
    Error VerifyBatch(const std::vector<VerifyItem>& items, std::vector<VerifyResult>& results) const;
    Error IdentityVerifyBatch(const std::vector<IdentityItem>& items, std::vector<std::string>& out) const;
    // The same two one-signature verbs for callers that arrive MANY AT A TIME on their own threads (orderer Broadcast handlers behind
    // SigFilter, validator goroutines on memo misses): blocking, same answers, calls in flight together share a launch (coalescer.h).
    VerifyResult VerifyCoalesced(const ECDSAPublicKey* k, const uint8_t* sig, size_t siglen, const uint8_t* digest, size_t dlen) const;
    // identity.Verify(msg, sig): "" (nil) or the error text; *infrastructure = true when the device failed (no verdict)
    std::string IdentityVerifyCoalesced(const ECDSAPublicKey* k, const uint8_t* msg, size_t msglen, const uint8_t* sig, size_t siglen,
                                        bool* infrastructure) const;
    void CoalescerConfigure(uint32_t window_us, uint32_t max_batch) const;
    void CoalescerStats(uint64_t* calls, uint64_t* launches, uint64_t* largest_batch) const;
    // Block-level pre-verify pass (block_prepass.h): one fused launch for every creator / endorsement signature of the block.
    Error PreVerifyBlock(const uint8_t* block, size_t len, BlockVerdicts& out, const PassOptions& opt = PassOptions()) const;
    struct BlockUpload;
    // up: an upload of the block started ahead (StartBlockUpload); joined right before the submission.  nullptr: the block
    // travels with the submission.
    Error PreVerifyParsed(const uint8_t* block, const ParsedBlock& parsed, BlockVerdicts& out, BlockUpload* up = nullptr,
                          const PassOptions& opt = PassOptions()) const;
    // The same pass with the envelope walk, the signature gates and the identity lookup ON THE DEVICE (block_walk_dev.h): the host lists
    // the envelopes, the block is read where StartBlockUpload put it, only flags (and, when asked for, tuple records / digests) come
    // back.  Serves blocks whose identities the provider has already met and whose signatures have the common DER shape; anything else
    // is DECLINED (returns 1, *why says why, nothing was written) and the caller takes ParseBlock + PreVerifyParsed, which also learns
    // the new identities.  0: done (`out` as PreVerifyParsed fills it; parsed.tuples too with WANT_TUPLES); < 0: FABGPU_E* - with
    // FABGPU_ETOOBIG parsed.n_tx / *n_tuples hold what the caller must make room for.
    enum : unsigned { WANT_TUPLES = 1, WANT_QXY = 2 };   // what the caller reads besides flags and statuses: tuple records; keys
    int PreVerifyBlockOnDevice(const uint8_t* block, size_t len, ParsedBlock& parsed, BlockVerdicts& out, BlockUpload& up, const PassOptions& opt,
                               unsigned want, uint32_t cap_tx, uint32_t cap_tuples, uint32_t* n_tuples, const char** why) const;
    // The device walker alone (tests: it must produce what ParseBlock produces, record for record): fills `parsed` like ParseBlock does,
    // except first_channel_id.  0 done, 1 declined, < 0 FABGPU_E*.
    int WalkBlockOnDevice(const uint8_t* block, size_t len, ParsedBlock& parsed, const char** why) const;
    // option pass_device_walk = 0 keeps every block on the host walk (A/B runs); default on
    bool DeviceWalkEnabled() const;
    // ---- verdict memo (second half) ----
    // The pass answers in advance the question the unchanged Go validators will ask one signature at a time:
    // bccsp.Verify(k, signature, digest) (msp/identities.go:188).  An entry is keyed on exactly those three byte strings - the
    // key's (X, Y), the DER signature as it sits in the block, the digest the DEVICE computed over the bytes it verified - each
    // length-framed, so a verdict can only ever be found again by a caller holding the same key, signature and digest.
    // Lookup: 0 = hit (*status = tuple status: 0 valid; 1 / 2 / 3 / 5 = the reference rejects, its exact error text comes from
    // bccsp/sw on that one tuple), 1 = miss (ask bccsp/sw).  Bounded: at most memo_capacity entries; the oldest BLOCK goes first.
    // issuer_hash32 != nullptr: the entry of an idemix pseudonym signature (qx, qy = Nym) verified under the issuer key with that ipk.Hash
    int MemoLookup(const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen, const uint8_t* digest, size_t dlen, uint8_t* status,
                   const uint8_t* issuer_hash32 = nullptr) const;
    size_t MemoEvictBlock(uint64_t block_seq) const;          // the validator wrapper calls this when Validate(block) returned
    size_t MemoHasBlock(uint64_t block_seq) const;            // entries still held under block_seq (0: never seeded, evicted, or aged out)
    void MemoStats(uint64_t* entries, uint64_t* hits, uint64_t* misses, uint64_t* evicted) const;
    void MemoSetCapacity(size_t max_entries) const;
    // identity cache bounds (msp/cache/cache.go keeps 100 deserialized identities; the pass sees every client certificate too)
    void SetIdentityCacheLimits(size_t max_identities, size_t max_registered_keys, uint32_t register_after_hits) const;
    size_t IdentityCacheSize() const;
    // device-route statistics since construction: launches repeated after a wrong "everybody is registered" prediction, tuples whose
    // certificate the device decoded itself, identities that entered the cache that way, signatures that took the general DER parser
    void PassStats(uint64_t out[4]) const;
    // Starts the upload of a block on a helper thread (blocks of 4 MiB and more) so that it travels while the caller parses
    // and gates; join() returns the token for PreVerifyParsed (0 if nothing was staged).
    struct BlockUpload {
        std::thread th;
        uint64_t token = 0;
        int rc = -1;
        int dev = 0;                 // the context (index into the provider's pool) the block travels to: the pass runs there
        bool started = false;        // an upload was started (join() may already have been called)
        bool routed = false;         // dev was chosen by RouteBlock and counts as a pass in flight on it until the upload object dies
        const GPUCSP* owner = nullptr;
        const uint8_t* block = nullptr;   // what was uploaded (a retry of the same call finds its upload again: bccsp_capi.cpp)
        size_t len = 0;
        uint64_t seq = 0;
        HostCopy copy;               // keep_host_copy: the block's bytes in memory the device context owns (p == nullptr: none) - handed to
                                     // the block's memo table when the pass publishes one, released with the upload object otherwise
        uint64_t join() {
            if (th.joinable()) th.join();
            return rc == 0 ? token : 0;
        }
        BlockUpload() {}
        BlockUpload(const BlockUpload&) = delete;
        BlockUpload& operator=(const BlockUpload&) = delete;
        ~BlockUpload();
    };
    // Chooses the pass's device (RouteBlock) and, for blocks of pass_stage_min_bytes and more, starts the upload to it.
    // keep_host_copy: the upload leaves the block's bytes in host memory of the context (BlockUpload::copy) for the digest memo.
    void StartBlockUpload(BlockUpload& up, const uint8_t* block, size_t len, uint64_t block_seq = 0, bool keep_host_copy = false) const;
    bool HashMemoEnabled() const;
    // An idemix MSP of the channel (msp/idemixmsp.go:99-173 Setup): its creators' pseudonym signatures are then verified by the
    // pre-verify pass too.  ipk_raw: marshalled idemix.IssuerPublicKey.  Returns the device issuer id, or -1 (not accelerated).
    int64_t RegisterIdemixMSP(const std::string& mspid, const uint8_t* ipk_raw, size_t len, const std::string& channel = std::string()) const;
    // Registers an idemix issuer on EVERY device of the pool, under one lock, so that its id is the same everywhere (ids are handed
    // out in order of registration per context); -1: not accelerated.  ipk_raw: marshalled idemix.IssuerPublicKey.
    int64_t ImportIdemixIssuer(const uint8_t* ipk_raw, size_t len, std::string* err = nullptr) const;
    fabgpu_ctx* ctx() const { return devs_[0]->ctx; }          // the first context (tests, single-device callers)
    // the context a flat batch (Verify / VerifyBatch / IdentityVerifyBatch / a coalesced launch) runs on: round the ring
    fabgpu_ctx* flat_ctx() const { return devs_[flat_rr_.fetch_add(1, std::memory_order_relaxed) % devs_.size()]->ctx; }
    const ProviderOptions& options() const { return opts_; }

   private:
    GPUCSP() {}
    // One device context of the pool and the provider's per-device state: the device's copy of the identity cache and its version.
    struct Dev {
        fabgpu_ctx* ctx = nullptr;
        int ordinal = 0;
        std::atomic<uint32_t> in_flight{0};                 // passes routed here that have not finished
        std::atomic<uint64_t> passes{0};                    // passes served
        // The device's copy of the identity cache (block_walk_dev.h walk_idtab_set): rebuilt whenever id_version_ moved.  Passes hold
        // idtab_rw shared from the version check until they have translated the device's identity indices back; a rebuild holds it exclusively.
        std::shared_timed_mutex idtab_rw;
        std::atomic<uint64_t> idtab_version{0};
        std::vector<uint64_t> idtab_host;                   // serial of the cache entry behind index k of the device table
        std::vector<uint8_t> idtab_bytes;                   // scratch of the rebuild (3 MB when the cache is full: not reallocated per version)
        std::vector<DevIdEntry> idtab_ents;
    };
    std::vector<std::unique_ptr<Dev>> devs_;
    mutable std::atomic<uint64_t> flat_rr_{0};
    mutable ProviderOptions opts_;
    mutable std::mutex opt_mu_;
    // A P-256 key's comb table on every device of the pool (the table is built once, fabgpu_p256_key_register_many): the common key id,
    // or -1 when the devices disagree about it / a device failed (the fresh-key kernels then serve that key: always correct).
    int64_t RegisterKeyOnAllDevices(const uint8_t* qx32, const uint8_t* qy32, const int32_t* prebuilt_table = nullptr) const;
    mutable std::mutex reg_mu_;                             // registrations take turns: ids stay the same on every device
    // Registrations that reached SOME devices of the pool and failed on another (ENOMEM, a context's table limit): replayed, in order,
    // before anything new is installed anywhere - so a transient failure heals and a lasting one stops further installs instead of
    // leaving the per-device id counters offset for good.  Guarded by reg_mu_.
    struct PendingKey { uint8_t qx[32], qy[32]; };
    mutable std::vector<PendingKey> pending_keys_;
    mutable std::vector<std::string> pending_issuers_;      // marshalled IssuerPublicKey bytes
    mutable bool reg_failure_logged_ = false, issuer_mismatch_logged_ = false;
    mutable uint32_t heal_attempts_ = 0;                    // failed replays of the front pending entry (reg_mu_)
    mutable std::atomic<uint64_t> reg_dropped_{0}, reg_id_mismatches_{0};   // GetOption("registrations_dropped" / "registration_id_mismatches")
    bool HealPendingRegistrationsLocked() const;            // true: nothing is pending any more
    void Preallocate() const;
    // identity cache of the pre-verify pass (msp/cache/cache.go): SerializedIdentity bytes -> P-256 key + device key id
    struct CachedIdentity {
        bool p256 = false;
        uint8_t qx[32], qy[32];
        int64_t key_id = -1;
        uint32_t hits = 0;              // tuples that named this identity (drives device-table registration)
        bool registering = false;       // some thread is building the device table right now
        uint64_t table_hash = 0;        // walk::id_hash_host of the identity bytes under idtab_seed_ (computed once, when it enters the cache)
        uint64_t serial = 0;            // unique per cache entry: how the device table's indices find their way back (idserial_)
    };
    // Bounded LRU (the reference's msp cache is one: msp/cache/cache.go:14-18, second_chance.go).  Identities come out of
    // UNVALIDATED blocks, so neither host memory nor device tables may grow with what a block names: at most id_max_ cached
    // identities; a device comb table (640 KiB, ~6 ms of host work) only for an identity that was named id_register_after_ times,
    // built outside idmu_, and at most id_max_registered_ of them - everybody else verifies on the fresh-key kernels.
    typedef std::list<std::pair<std::string, CachedIdentity>> IdList;
    mutable std::mutex idmu_;
    mutable IdList idlru_;
    mutable std::unordered_map<std::string, IdList::iterator> idcache_;
    mutable std::unordered_map<uint64_t, IdList::iterator> idserial_;
    mutable uint64_t id_next_serial_ = 1;
    // (idmu_ held) a new cache entry at the front of the LRU list; evicts what no longer fits
    void InsertIdentityLocked(std::string&& key, CachedIdentity ci, bool evict_now = true) const;
    mutable size_t id_max_ = 4096, id_max_registered_ = 256, id_registered_ = 0;
    mutable uint32_t id_register_after_ = 64;
    // Every device of the pool holds a copy of this cache (Dev::idtab_*), rebuilt before a pass on that device whenever id_version_ moved.
    mutable std::atomic<uint64_t> id_version_{1};
    const uint64_t idtab_seed_ = MakeSeed();      // the device table's hash is keyed per provider (block_walk_core.h id_hash_host)
    static uint64_t MakeSeed();
    int SyncDeviceIdentityTable(Dev& dv) const;
    mutable std::atomic<uint64_t> pass_relaunches_{0}, pass_decoded_{0}, pass_learned_{0}, pass_general_der_{0};
    void EvictIdentitiesLocked() const;
    void RegisterQueued(const std::vector<std::string>& to_register) const;
    bool RegisterKeysOnAllDevices(const std::vector<std::pair<std::string, CachedIdentity>>& keys, std::vector<int64_t>& ids) const;
    void SeedMemo(const uint8_t* block, const ParsedBlock& pb, BlockVerdicts& out, const PassOptions& opt, std::vector<uint32_t>& sel_scratch, int gate_max,
                  BlockUpload* up = nullptr) const;
    // verdict memo
    // One table per BLOCK (seeded once by the pass, dropped whole when the block's validation returns): an open-addressed index over
    // length-framed keys stored back to back - no allocation per entry, filled by the pass's worker threads in parallel (a
    // 10 000-transaction block seeds 40 000 entries; a node-per-entry map cost more than the device call).  Lookups take a shared lock.
    struct BlockMemo {
        uint64_t seq = 0;
        uint64_t gen = 0;                                 // unique per publication (the lookups' per-thread hints name a table by it)
        uint32_t n = 0, mask = 0;
        std::unique_ptr<std::atomic<uint32_t>[]> slots;   // entry index + 1; 0 = empty
        std::vector<uint32_t> key_off;                    // n + 1 offsets into keys
        std::unique_ptr<uint8_t[]> keys;                  // framed keys: X || Y || u32 len || sig || u32 len || digest (not zero-filled)
        size_t keys_cap = 0, slots_cap = 0;
        std::vector<uint8_t> status;                      // n
        // What a lookup reads: the arrays above (a table SeedMemo built on the host) or - a table the DEVICE built, block_walk_dev.h
        // WalkOut::memo_* - the same four arrays inside `pin`, pinned host memory the pass copied them into.
        const uint32_t* slots_v = nullptr;
        const uint32_t* key_off_v = nullptr;
        const uint8_t* keys_v = nullptr;
        const uint8_t* status_v = nullptr;
        // a table the device built: its keys end with the digest's LENGTH field and the 32-byte digests sit in an array of their own, by
        // entry (they are the one part of a key that only exists once the verify launches are through; everything else travels beside
        // them); it may hold entries without a slot (candidates that were not decided: status 255) - n counts the ones with a slot
        const uint8_t* digests_v = nullptr;
        uint32_t n_entries = 0;
        // The block's DIGEST memo (HashLookup): the block's bytes in host memory of a device context (`copy`; the orderers' signature
        // messages, which are not in the block, in `tail` at virtual offset tail_base), per entry the two spans of its signed message,
        // and a second slot table over walk::msg_fingerprint of the message.  Device-built tables: hslots_v / hspans_v point into `pin`;
        // host-built ones into the two vectors.  No copy, no index: every lookup misses.
        const uint32_t* hslots_v = nullptr;
        const uint32_t* hspans_v = nullptr;
        std::vector<uint32_t> hslots, hspans;
        HostCopy copy;
        std::vector<uint8_t> tail;
        uint32_t tail_base = 0;
        void ReleaseCopy() {
            host_copy_release(&copy);
            hslots_v = hspans_v = nullptr;
        }
        void* pin = nullptr;
        size_t pin_cap = 0;
        void* pin_keys = nullptr;                         // room of its own for the keys of a block whose signatures are far longer than usual
        size_t pin_keys_cap = 0;
        fabgpu_ctx* pin_ctx = nullptr;
        ~BlockMemo();
    };
    void PublishMemo(const std::shared_ptr<BlockMemo>& bm) const;   // push under the lock, oldest blocks out while over capacity
    mutable std::vector<std::shared_ptr<BlockMemo>> memo_free_;    // evicted tables, recycled: 7 MB of fresh pages per block otherwise
    mutable size_t memo_free_max_ = 4, scratch_free_max_ = 4;      // (both grow with the pool and with ProviderOptions::concurrent_passes)
    mutable BigReaderLock memo_mu_;                       // readers (every bccsp.Verify of every validator thread) share no cache line: reader_lock.h
    mutable std::deque<std::shared_ptr<BlockMemo>> memo_blocks_;   // oldest first
    mutable size_t memo_cap_ = (size_t)1 << 18;
    mutable ShardedCounter memo_hits_, memo_misses_;      // (bumped per lookup by every validator thread)
    mutable std::atomic<uint64_t> memo_evicted_{0};
    static size_t MemoPinLayout(uint32_t n_tuples, uint32_t n_creators, uint32_t* slot_cap, size_t* keys_cap, size_t* total, size_t* offs7 = nullptr);
    mutable ShardedCounter hash_hits_, hash_misses_;
    static size_t MemoKeyBytes(size_t siglen, size_t dlen, bool nym) { return 1 + (nym ? 32 : 0) + 64 + 4 + siglen + 4 + dlen; }
    static void MemoKeyWrite(uint8_t* out, const uint8_t* issuer_hash32, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen,
                             const uint8_t* digest, size_t dlen);
    mutable std::map<int64_t, std::array<uint8_t, 32>> idemix_issuer_hash_;   // device issuer id -> ipk.Hash (guarded by idmu_)
    static uint64_t MemoHash(const uint8_t* sig, size_t siglen, const uint8_t* digest, size_t dlen);
    mutable std::map<std::string, int64_t> idemix_msps_;   // mspid -> device issuer id, or -2 while channels disagree (guarded by idmu_)
    mutable std::map<std::string, std::map<std::string, int64_t>> idemix_msp_channels_;   // mspid -> channel -> that channel's latest issuer id (idmu_)
    // scratch of the pre-verify pass, reused from block to block: a pass leases one set (a peer's channels run passes side by side)
    struct PassScratch {
        struct Gated {
            uint8_t qx[32], qy[32], r[32], s[32];   // idemix tuple: qx, qy = pseudonym; r, s = ProofC, ProofSSk
            uint8_t srn[32], nonce[32];              // idemix only: ProofSRNym, Nonce
            int64_t key_id;                          // idemix: issuer id
            bool submit, nym;
        };
        std::vector<Gated> gt;
        std::vector<uint32_t> sub, ids, off, pre_idx;
        std::vector<uint8_t> qx, qy, r, s, dig, st, hash_digests;
        std::vector<uint32_t> pre_off, gsp;
        std::vector<uint64_t> bits;
        // the block's idemix creators (they ride in the ECDSA submission): tuple indices, message spans, issuer ids, six field columns, results
        std::vector<uint32_t> nym_idx, nym_sp, nym_iss;
        std::vector<uint8_t> nym_fields, nym_st;
        std::vector<uint64_t> nym_bits;
        // the device walk: envelope list, block-signature tuples, identity indices per tuple
        std::vector<uint32_t> env_spans, payload_spans, id_idx;
        std::vector<uint32_t> env_counts;          // 4 per envelope: what walk_count_kernel would find (counted here while the block travels)
        std::vector<uint8_t> env_type, env_understood;
        std::vector<BlockTuple> block_sigs;
        std::vector<WalkLearn> learn;              // identities the device decoded and offers to the cache
        std::vector<int32_t> nym_issuer_rank;      // device route: issuer id of an idemix creator's row, by creator rank (-1: not one)
    };
    struct CoReqV : CoalescedBase {
        VerifyItem item;
        VerifyResult res;
    };
    struct CoReqI : CoalescedBase {
        IdentityItem item;
        std::string out;
        bool infra = false;
    };
    mutable Coalescer<CoReqV> co_verify_;
    mutable Coalescer<CoReqI> co_identity_;
    mutable std::mutex pass_mu_;                                      // guards scratch_free_
    mutable std::vector<std::unique_ptr<PassScratch>> scratch_free_;
};

}  // namespace bccsp
}  // namespace fab
