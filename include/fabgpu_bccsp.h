/*
 * fabgpu_bccsp.h - flat C view of the C++ host mirror (fabric-mod_amd/csrc/bccsp_host.h) so that the
 * Python parity tests (and any other FFI) can drive the provider with the reference's vocabulary:
 *   bccsp/sw/impl.go:177-194   CSP.Hash        -> fabgpu_csp_hash
 *   bccsp/sw/impl.go:247-270   CSP.Verify      -> fabgpu_csp_verify / fabgpu_csp_verify_batch
 *   msp/identities.go:169-196  identity.Verify -> fabgpu_csp_identity_verify_batch
 * Error strings are the Go `error` text ("" == nil).  All functions return 0, or a negative FABGPU_E* when the
 * device failed (the caller then falls back to bccsp/sw).  The production binding is cgo over fabgpu.h
 * (INTEGRATION.md); this header adds no new capability.
 */
#ifndef FABGPU_BCCSP_H
#define FABGPU_BCCSP_H
#include "fabgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fabgpu_csp fabgpu_csp;

int fabgpu_csp_new(const fabgpu_cfg* cfg, fabgpu_csp** out, char* err, size_t errcap);
/* ONE provider over SEVERAL devices.  The reference has one process-global BCCSP (bccsp/factory/factory.go:41-55: GetDefault), handed to
 * every channel's validator (core/peer/peer.go:337-355), and validates its channels side by side (core/committer/txvalidator/v20/
 * validator.go:194-210): whatever drives more than one GPU has to sit BEHIND that one object.  fabgpu_csp_new2 makes one device context
 * per entry of `devices` (an ordinal may repeat: several contexts on one GPU); every block pass is routed to the context with the fewest
 * passes in flight (ties: round the ring from block_seq mod G), the identity cache and the verdict memo stay ONE on the host -
 * fabgpu_csp_memo_lookup does not care which device verified - and registered keys / idemix issuers get their tables on every device
 * under the same id.  Flat batches (verify_batch, identity_verify_batch, coalesced launches) take turns round the pool.
 * This struct is what `GPU:` of the BCCSP configuration carries (go/bccsp/factory/gpufactory.go GPUOpts; pattern bccsp/pkcs11/conf.go:70-84);
 * `size` = sizeof(fabgpu_csp_opts) as the caller compiled it (the struct may grow at its end). */
typedef struct fabgpu_csp_opts {
    uint32_t size;
    int32_t n_devices;             /* 0: every visible gfx950 device */
    const int32_t* devices;        /* n_devices HIP ordinals, or NULL = 0 .. n_devices-1 */
    uint32_t ctx_flags;            /* FABGPU_FLAG_* for every context */
    uint32_t concurrent_passes;    /* per device: what this many overlapping block passes need - staging slots, pinned memo tables, pass
                                      arrays, kept block copies (concurrent_passes + 2 of them), a slab of key tables - is allocated NOW, not
                                      when passes first overlap (0: on demand).  Any value > 0 makes ALL THREE staging slots of a device
                                      (3 x expect_block_bytes of device memory): uploads walk round all of them even when passes never overlap */
    uint64_t expect_block_bytes;   /* sizes that pre-allocation (0: 64 MiB) */
    uint32_t expect_tuples;        /* (0: 65 536) */
    /* switches that were environment variables through round 3: 0 = the default, > 0 on, < 0 off (a zeroed struct is all defaults) */
    int32_t pass_device_walk;      /* < 0: every block takes the host walk (default: the walk runs on the device) */
    int64_t pass_stage_min_bytes;  /* > 0: only blocks of at least this many bytes are uploaded ahead of their pass (default: every block) */
    int32_t pass_device_memo;      /* < 0: the verdict memo is seeded on the host also on the device route (default: built by the device) */
    int32_t pass_host_counts;      /* > 0: the envelopes' tuples are counted on the host while the block travels (default off) */
    int32_t pass_timing;           /* > 0: stage breakdown of every pass on stderr (default off) */
    int32_t pass_hash_memo;        /* < 0: no digest memo - memo-seeding passes keep no host copy of their block and fabgpu_csp_hash_lookup
                                      always misses (default on) */
    uint32_t hash_memo_blocks;     /* per device: host copies of blocks the digest memo keeps at a time, each as large as its block, pinned
                                      (0: 8; at most 64).  A block beyond that simply has no digest memo: its messages are hashed on the CPU */
} fabgpu_csp_opts;
int fabgpu_csp_new2(const fabgpu_csp_opts* opts, fabgpu_csp** out, char* err, size_t errcap);
void fabgpu_csp_free(fabgpu_csp* csp);
fabgpu_ctx* fabgpu_csp_ctx(fabgpu_csp* csp);            /* the first device's context */
int fabgpu_csp_device_count(fabgpu_csp* csp);            /* contexts in the pool */
fabgpu_ctx* fabgpu_csp_ctx_of(fabgpu_csp* csp, int d);
/* passes[d] = block passes context d has served; returns the number of contexts (cap must be at least that) */
int fabgpu_csp_passes_per_device(fabgpu_csp* csp, uint64_t* passes, int cap);
int fabgpu_csp_route_block(fabgpu_csp* csp, uint64_t block_seq);   /* where a pass named block_seq would go right now */
/* the switches above on a living provider (tests, A/B runs): "pass_device_walk", "pass_stage_min_bytes", "pass_device_memo",
 * "pass_host_counts", "pass_skip_hash_checks", "pass_timing", "pass_hash_memo" - same convention: 0 the default, > 0 on / the threshold, < 0 off; get also
 * answers "n_devices" and two counters: "registrations_dropped" (key / issuer tables that could not be brought onto every device of the
 * pool after three attempts - those identities verify on the fresh-key kernels / bccsp/idemix) and "registration_id_mismatches".
 * FABGPU_EINVAL: no such option. */
int fabgpu_csp_set_option(fabgpu_csp* csp, const char* name, int64_t value, int64_t* previous);
int fabgpu_csp_get_option(fabgpu_csp* csp, const char* name, int64_t* value);

/* BCCSP.KeyImport for a P-256 public key (bccsp/sw/keyimport.go:103-134; pattern bccsp/pkcs11/pkcs11.go:148-179): checks
 * curve membership and registers the key's comb table on the device (fabgpu_p256_key_register), so that batches whose
 * signers were all imported this way run on the keyed kernels.  err: "" or the Go error text. */
int fabgpu_csp_key_import(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, int* on_curve, char* err, size_t errcap);

/* alg == NULL mirrors opts == nil. */
int fabgpu_csp_hash(fabgpu_csp* csp, const uint8_t* msg, size_t len, const char* alg, uint8_t* digest32, char* err, size_t errcap);

/* BCCSP.Hash for bytes a memo-seeding block pass has ALREADY hashed on the device: the `digest, err := id.msp.bccsp.Hash(msg, hashOpt)` half
 * of identity.Verify (msp/identities.go:173-181 -> bccsp/sw/impl.go:177-194), which the unchanged validators call once per creator /
 * endorsement signature over bytes the pass hashed a moment earlier - 100 MB per 10 000-transaction block.  A pass with
 * FABGPU_PASS_SEED_MEMO keeps the block's bytes in host memory THE LIBRARY owns (the pinned buffer the upload went through anyway; never the
 * caller's buffer) until fabgpu_csp_memo_evict_block, and indexes every signed message by a fingerprint of a few sampled bytes.  The
 * fingerprint only chooses where to look: the stored digest is returned ONLY when every byte of `msg` equals the bytes the device hashed
 * (memcmp; piecewise for prp || endorser, validator_keylevel.go:246-258) - a flipped byte anywhere, another length, an evicted block, a
 * message the device did not hash: miss.
 * Returns 0: hit, digest32 = SHA-256(msg) as the device computed it (the digest the block's verdict-memo entries are keyed on);
 * 1: miss - the caller computes the digest itself (bccsp/sw), always a correct answer.  Never an infrastructure error.  Messages shorter
 * than 64 bytes always miss (one SHA-256 block costs less than the call).  Callers map *bccsp.SHA256Opts (what msp/identities.go:216-224
 * selects for the SHA2 family) to this entry and every other HashOpts to bccsp/sw. */
int fabgpu_csp_hash_lookup(fabgpu_csp* csp, const uint8_t* msg, size_t len, uint8_t* digest32);
/* digest memo counters since the provider was made: lookups answered / left to the CPU; host copies of blocks held right now, their
 * (pinned) bytes, and passes that wanted a copy when the pool was exhausted (any pointer may be NULL) */
int fabgpu_csp_hash_memo_stats(fabgpu_csp* csp, uint64_t* hits, uint64_t* misses, uint64_t* blocks_held, uint64_t* bytes_held, uint64_t* refused);

/* qx == NULL mirrors k == nil.  *valid = 0/1; err = Go error text or "".  *flags bit0: tuple must be decided by bccsp/sw. */
int fabgpu_csp_verify(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen,
                      const uint8_t* digest, size_t dlen, int* valid, int* flags, char* err, size_t errcap);

/* The two one-signature verbs for callers that arrive MANY AT A TIME, each on its own thread: the orderer's Broadcast handlers behind
 * SigFilter (orderer/common/msgprocessor/sigfilter.go:50-80 -> identity.Verify, msp/identities.go:169-196; one goroutine per client
 * stream), the validator pool's bccsp.Verify calls that miss the verdict memo (core/committer/txvalidator/v20/validator.go:198-208).
 * Blocking, same answers and error texts as fabgpu_csp_verify / fabgpu_csp_identity_verify_batch; calls that are in flight at the same
 * moment share ONE launch (a caller that finds the device busy queues behind the running launch and travels with the next; a caller
 * that finds it idle waits window_us - default 50 - for company).  A device failure is FABGPU_ELAUNCH for every caller of that launch,
 * never a verdict.  err == "" means nil. */
int fabgpu_csp_verify_coalesced(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen,
                                const uint8_t* digest, size_t dlen, int* valid, int* flags, char* err, size_t errcap);
int fabgpu_csp_identity_verify_coalesced(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* msg, size_t msglen,
                                         const uint8_t* sig, size_t siglen, char* err, size_t errcap);
int fabgpu_csp_coalescer_configure(fabgpu_csp* csp, uint32_t window_us, uint32_t max_batch);
int fabgpu_csp_coalescer_stats(fabgpu_csp* csp, uint64_t* calls, uint64_t* launches, uint64_t* largest_batch);

/* n keys (n x 32), ragged signatures and digests; valid: n bytes; errs: n * errstride chars (NUL terminated, truncated). */
int fabgpu_csp_verify_batch(fabgpu_csp* csp, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* sig_arena,
                            const uint32_t* sig_off, const uint8_t* dig_arena, const uint32_t* dig_off, uint8_t* valid,
                            char* errs, size_t errstride);

/* identity.Verify(msg, sig) for n (key, msg, sig) triples; errs[i] == "" means nil. */
int fabgpu_csp_identity_verify_batch(fabgpu_csp* csp, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* msg_arena,
                                     const uint32_t* msg_off, const uint8_t* sig_arena, const uint32_t* sig_off, char* errs,
                                     size_t errstride);

/* ---- block-level pre-verify pass (extensions/validation/validation.go:48-64 is the hook) ----
 * One fused launch for every signature of a marshalled common.Block:
 *   creator:      SignatureHeader.creator over Envelope.payload            (core/common/validation/msgvalidation.go:258-298)
 *   endorsements: Endorsement.endorser over prp || Endorsement.endorser    (.../statebased/validator_keylevel.go:246-258)
 * Identities (PEM x509, P-256) are imported once and cached (device comb tables), the block buffer is the message arena,
 * each proposal_response_payload is hashed once.  The two other SHA-256 checks of ValidateTransaction ride along in the same
 * submission: CheckTxID (protoutil/proputils.go:366-375) and the proposal hash of every action (protoutil/txutils.go:431-447,
 * core/common/validation/msgvalidation.go:233-241).  tx_flags[t], in the order the reference would reject: 3 not understood
 * (left to the Go validators), 1 creator signature bad, 5 TxID does not match, 6 a proposal hash does not match, 2 an
 * endorsement signature bad, 4 an identity needs bccsp/sw, 0 everything checks.
 * tuple_status[i]: 0..4 as in fabgpu.h, 5 signature does not unmarshal, 6 identity needs bccsp/sw, 7 empty signature.
 * Returns FABGPU_ETOOBIG with *n_tx / *n_tuples set when the caller's arrays are too small. */
int fabgpu_csp_block_preverify(fabgpu_csp* csp, const uint8_t* block, size_t len, uint32_t* n_tx, uint8_t* tx_flags, uint8_t* tx_type,
                               uint32_t cap_tx, uint32_t* n_tuples, uint32_t* tuple_tx, uint8_t* tuple_kind, uint8_t* tuple_status,
                               uint32_t cap_tuples);
/* ---- the pass, second form: verdicts tied to the BYTES they were computed over, the orderers' block signatures, the verdict memo ----
 * Everything fabgpu_csp_block_preverify returns, plus per tuple: the spans of (identity, prefix, suffix, signature) in the virtual arena
 * block || padding || tail (see fabgpu_block_tuples), the SHA-256 digest of the signed message AS THE DEVICE COMPUTED IT, and the
 * P-256 key the identity carries.  A consumer must key on these (never on tuple position): a crafted envelope could make another
 * parser see other bytes; the walker refuses envelopes with repeated singular fields for the same reason (tx_flags = 3).
 * Tuples of kind 2 (tx = 0xFFFFFFFF, listed last) are the orderers' signatures over the block - the SignedData MCS.VerifyBlock hands
 * to the BlockValidation policy (internal/peer/gossip/mcs.go:166-193).  BlockDataHash (one serial SHA-256 over the whole BlockData,
 * protoutil/blockutils.go:65-68) stays with the caller: a single SHA-256 stream cannot be parallelised.
 * FABGPU_PASS_SEED_MEMO: every tuple the device hashed and decided is remembered under `block_seq` as
 * (key X||Y, signature bytes, digest) -> status; fabgpu_csp_memo_lookup answers the bccsp.Verify(k, sig, digest) calls the unchanged
 * validators make afterwards (msp/identities.go:188) and fabgpu_csp_memo_evict_block drops the block's entries when Validate returns. */
#define FABGPU_PASS_SEED_MEMO 1u
#define FABGPU_PASS_NO_BLOCK_SIGS 2u /* do not verify the orderers' block signatures (their tuples report status 8) */
typedef struct fabgpu_block_pass {
    /* in */
    const uint8_t* block;
    size_t len;
    uint64_t block_seq;
    uint32_t flags;              /* FABGPU_PASS_* */
    uint32_t cap_tx, cap_tuples; /* capacity of the arrays below */
    /* out: counts (always set; FABGPU_ETOOBIG when a capacity is too small - nothing was launched) */
    uint32_t n_tx, n_tuples, n_block_sigs, memo_seeded;
    uint32_t tail_base, tail_len;
    uint8_t block_sigs_understood; /* 1: metadata[SIGNATURES] parsed (kind-2 tuples are complete) */
    /* out: arrays, caller-allocated, any may be NULL */
    uint8_t* tx_flags;           /* cap_tx */
    uint8_t* tx_type;            /* cap_tx */
    uint32_t* tuple_tx;          /* cap_tuples */
    uint8_t* tuple_kind;         /* cap_tuples: 0 creator, 1 endorsement, 2 block signature */
    uint8_t* tuple_status;       /* cap_tuples */
    uint32_t* tuple_spans;       /* cap_tuples x 8 */
    uint8_t* tuple_digest;       /* cap_tuples x 32; zero where tuple_hashed[i] == 0 */
    uint8_t* tuple_hashed;       /* cap_tuples */
    uint8_t* tuple_qxy;          /* cap_tuples x 64 */
    uint8_t* tail;               /* tail_cap bytes: the block-signature messages (optional) */
    uint32_t tail_cap;
    /* out: submitted tuples that went through per-key device tables.  Host walk: all of them, or none (one identity without a table
     * sends the block down the fresh-key kernel).  Device walk: per launch class - the creators' launch and the launch of everybody
     * else each take the tables when every one of their submitted tuples has one. */
    uint32_t n_keyed;
    /* out (ABI v5), device walk: tuples whose identity the provider had never met - their certificates were decoded on the device
     * (msp/mspimpl.go:408-421 deserialization: PEM, x509 SubjectPublicKeyInfo, curve membership) and offered to its identity cache */
    uint32_t n_device_decoded;
    /* out (ABI v6): where this pass spent its time, host clock, milliseconds - [0] the outline of the block and the identity table's
     * sync (host walk: the gates), [1] waiting for the block's upload to finish, [2] the device phase (count .. finish, the verify
     * launches inside), [3] bookkeeping behind it (cache, learned identities, memo publication); [4] the device context that served it */
    float ms_stage[4];
    int32_t device_context;
} fabgpu_block_pass;
int fabgpu_csp_block_preverify2(fabgpu_csp* csp, fabgpu_block_pass* pass);
/* FABGPU_ETOOBIG and the retry (both forms of the pass).  The block's upload starts before its shape is known; when the caller's arrays
 * turn out too small the pass returns FABGPU_ETOOBIG with the counts set, NOTHING was launched, and the library keeps the finished upload
 * for the retry.  Contract:
 *  - by the time FABGPU_ETOOBIG is returned the upload has been waited for: the library never reads `block` after a call has returned
 *    (cgo: no Go pointer is retained), so the caller may free or reuse the buffer;
 *  - a retry finds the kept upload only if it passes the same pointer, length and block_seq, a fingerprint of the buffer (its first and
 *    last KiB and 64 samples between them) is unchanged, and it comes within one second; any other call uploads afresh (correct, just
 *    slower).  The caller MUST NOT modify the buffer between the attempts: the fingerprint is a guard against an allocator handing the
 *    same address to another block, not a proof that every byte is the one that was uploaded;
 *  - a caller that will not retry calls fabgpu_csp_block_pass_abandon (returns 1 if an upload was dropped, 0 if none was kept);
 *    otherwise the kept upload is dropped by the next pass that finds it older than a second, and its device counts as busy until then. */
int fabgpu_csp_block_pass_abandon(fabgpu_csp* csp);
/* 0: hit, *status = 0 valid / 1 arithmetic reject / 2 high-S / 3 r out of range (the reference rejects: ask bccsp/sw for its error
 * text); 1: miss -> bccsp/sw.  Never an infrastructure error: a miss is always a correct answer. */
int fabgpu_csp_memo_lookup(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen, const uint8_t* digest,
                           size_t dlen, uint8_t* status);
/* The same for an idemix pseudonym signature (key = Nym.x, Nym.y; digest = SHA-256(message)): entries are bound to the issuer key they
 * were verified under - issuer_hash32 = idemix.IssuerPublicKey.Hash of the key the CALLER verifies under (bccsp.IdemixNymSignerOpts.IssuerPK) -
 * so a verdict reached under one channel's issuer can never answer for another's (two channels may give their idemix MSPs the same id).
 * ECDSA entries and pseudonym entries live in separate key domains. */
int fabgpu_csp_memo_lookup_nym(fabgpu_csp* csp, const uint8_t* issuer_hash32, const uint8_t* nym_x32, const uint8_t* nym_y32, const uint8_t* sig, size_t siglen,
                               const uint8_t* digest, size_t dlen, uint8_t* status);
/* *entries = memo entries still held under block_seq (0: never seeded, evicted, or pushed out by newer blocks).  The pass may run when
 * a block ARRIVES (extensions/gossip/state AddPayload: the marshalled bytes are in hand, gossip/state/state.go:592,785-787) and the
 * validator wrapper asks this before it would marshal and submit the same block again at Validate. */
int fabgpu_csp_memo_has_block(fabgpu_csp* csp, uint64_t block_seq, uint64_t* entries);
int fabgpu_csp_memo_evict_block(fabgpu_csp* csp, uint64_t block_seq, uint64_t* evicted);
int fabgpu_csp_memo_stats(fabgpu_csp* csp, uint64_t* entries, uint64_t* hits, uint64_t* misses, uint64_t* evicted);
int fabgpu_csp_memo_set_capacity(fabgpu_csp* csp, uint64_t max_entries);
/* Bounds of the pass's identity cache (identities come out of unvalidated blocks): at most max_identities cached (LRU, like
 * msp/cache), a device comb table only for an identity named register_after_hits times, at most max_registered_keys tables. */
int fabgpu_csp_identity_cache_limits(fabgpu_csp* csp, uint64_t max_identities, uint64_t max_registered_keys, uint32_t register_after_hits);
int fabgpu_csp_identity_cache_size(fabgpu_csp* csp, uint64_t* identities);
/* Both forms of the pass walk a staged block ON THE DEVICE (envelope walk, signature gates - every DER encoding, decided as
 * bccsp/utils/ecdsa.go:43-92 decides it -, identity lookup, certificates of identities nobody has met decoded by a wavefront each,
 * submission arrays, digest comparisons and flags as kernels; block_walk_dev.h).  The host walk answers what is left: blocks that were
 * not staged, blocks on a provider with idemix MSPs, a certificate longer than the device decoder's buffer (3 KiB of DER), a block
 * without any signature for the device to decide. Same answers either way.  The provider option pass_device_walk < 0 keeps every block on the host
 * walk.  This reports how many passes went which way and why the last block was declined by the device walk. */
int fabgpu_csp_pass_routes(fabgpu_csp* csp, uint64_t* device_walks, uint64_t* host_walks, char* last_decline, size_t cap);
/* Device-route statistics since the provider was made: out4[0] verify launches repeated because the prediction "every signer of this
 * launch class has a comb table" (taken from the previous block, so that nothing waits for the gates) did not hold, [1] tuples whose
 * identity was not in the device's table (certificate decoded on the device), [2] identities that entered the cache that way,
 * [3] signatures outside the common DER shape (decided by the general parser, on the device). */
int fabgpu_csp_pass_stats(fabgpu_csp* csp, uint64_t* out4);
/* what the host route makes of a SerializedIdentity: 0 = PEM x509 certificate with an on-curve P-256 key (qxy64 = X || Y), 1 = anything else
 * (pure host; callers that hold identities as bytes - tools/go_call_replay.c - take the key bccsp.KeyImport would have remembered from it) */
int fabgpu_identity_to_p256(const uint8_t* ident, size_t len, uint8_t* qxy64);

/* (An x509 chain-link batch - crypto/x509 CheckSignatureFrom for n certificates - was an entry point of rounds 2-4 WITHOUT a consumer:
 * the only caller on this path is crypto/x509's own Verify (msp/mspimpl.go:721-739), which offers no hook for a pre-computed verdict, and
 * using one as advice would have meant skipping or forking chain building, constraints and expiry.  Removed in round 5) */
/* pure host helpers of the pass (no device): block structure, and the P-256 key of an x509 certificate */
int fabgpu_block_parse(const uint8_t* block, size_t len, uint32_t* n_tx, uint32_t* n_tuples, uint32_t* n_prefixes, uint8_t* tx_type, uint32_t cap_tx,
                       char* channel_id, size_t channel_cap);
int fabgpu_x509_p256_pubkey(const uint8_t* cert, size_t len, int is_pem, uint8_t* qx32, uint8_t* qy32);
/* the hash checks of a block: kind 0 = TxID (expect = 64 hex characters), 1 = proposal hash (expect = 32 bytes); message = the
 * concatenation of the three (start, end) spans; all offsets into the block buffer.  FABGPU_ETOOBIG with *n_checks set if cap is small. */
int fabgpu_block_hash_checks(const uint8_t* block, size_t len, uint32_t cap, uint32_t* n_checks, uint32_t* tx, uint8_t* kind, uint32_t* spans6,
                             uint32_t* expect2);

/* every (identity, message, signature) tuple the pass derives from a block (pure host; tests compare it with an independent decoder
 * and verify the tuples with the CPU oracle).  spans8[8i..8i+7] = (start, length) of identity, prefix, suffix, sig; the signed message is
 * prefix || suffix.  Offsets address the virtual arena  block || zero padding up to *tail_base || tail : the orderers' signatures over the
 * block (kind 2, tx 0xFFFFFFFF; internal/peer/gossip/mcs.go:166-193) sign  Metadata.value || signature_header ||
 * protoutil.BlockHeaderBytes(header)  - bytes the walker has to build, returned in `tail` (may be NULL). kind: 0 creator, 1 endorsement. */
int fabgpu_block_tuples(const uint8_t* block, size_t len, uint32_t cap, uint32_t* n_tuples, uint32_t* tx, uint8_t* kind, uint32_t* spans8,
                        uint8_t* tail, uint32_t tail_cap, uint32_t* tail_len, uint32_t* tail_base);

/* ---- idemix pseudonym signatures (creator signatures of idemix MSPs): host mirror of bccsp/idemix/handlers ----
 * fabgpu_csp_idemix_issuer_import: IssuerPublicKeyImporter.KeyImport (bccsp/idemix/handlers/issuer.go:115-136) for the
 * accelerated part - takes the marshalled idemix.IssuerPublicKey, registers HSk / HRand / Hash on the device.
 * *issuer_id = -1 with err == "": a key the device does not take (the proof inside the key is checked by bccsp/idemix).
 * fabgpu_csp_idemix_nym_verify_batch: NymVerifier.Verify (bccsp/idemix/handlers/nymsigner.go:62-95) over n tuples under one
 * issuer: nym public keys as NymPublicKeyImporter.KeyImport receives them (x || y), marshalled idemix.NymSignature bytes,
 * messages.  valid[i] 0/1; flags[i] bit0: tuple left to bccsp/idemix; errs[i] = Go error text ("" == nil). */
/* An idemix MSP of the channel (msp/idemixmsp.go:99-173): after this call fabgpu_csp_block_preverify also verifies the pseudonym
 * signatures of creators serialized under `mspid` (tuple_status 0 valid / 1 invalid / 6 left to bccsp/idemix).  Registering one MSP id
 * with a SECOND, different issuer key (another channel's MSP of the same name) makes the pass leave that MSP id's creators to
 * bccsp/idemix (status 6): it sees MSP ids, not channels.
 * fabgpu_csp_idemix_msp_register2 names the channel whose config carries the MSP: a channel's LATEST key for an MSP id replaces its earlier
 * one (issuer-key rotation), and the MSP id is ambiguous only while two channels' current keys for it differ.  (..._register = channel "".)
 * A key whose field 10 (Hash) is not HashModOrder of the rest of the key - which is what the reference recomputes and uses,
 * idemix/issuerkey.go:171-182 - is not accelerated (*issuer_id = -1). */
int fabgpu_csp_idemix_msp_register(fabgpu_csp* csp, const char* mspid, const uint8_t* ipk_raw, size_t len, int64_t* issuer_id);
int fabgpu_csp_idemix_msp_register2(fabgpu_csp* csp, const char* channel, const char* mspid, const uint8_t* ipk_raw, size_t len, int64_t* issuer_id);
int fabgpu_csp_idemix_issuer_import(fabgpu_csp* csp, const uint8_t* ipk_raw, size_t len, int64_t* issuer_id, char* err, size_t errcap);
/* 1: the marshalled idemix.IssuerPublicKey is in the encoding golang/protobuf itself produces (known fields only, ascending, single-byte
 * tags, minimal length varints, no empty singular bytes fields; idemix.proto:18-48) - only such a key is accelerated, because only for
 * it the library's issuer hash (the given bytes minus field 10) equals SetHash's (idemix/issuerkey.go:171-182, over the RE-MARSHALLED
 * key).  0: any other encoding (the import entry points then answer *issuer_id = -1 and bccsp/idemix serves the key).  No device needed. */
int fabgpu_idemix_issuer_key_is_canonical(const uint8_t* ipk_raw, size_t len);
int fabgpu_csp_idemix_nym_verify_batch(fabgpu_csp* csp, int64_t issuer_id, size_t n, const uint8_t* nym_arena, const uint32_t* nym_off,
                                       const uint8_t* sig_arena, const uint32_t* sig_off, const uint8_t* msg_arena, const uint32_t* msg_off,
                                       uint8_t* valid, uint8_t* flags, char* errs, size_t errstride);

#ifdef __cplusplus
}
#endif
#endif
