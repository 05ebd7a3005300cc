/*
 * fabgpu.h - C ABI of the MI355X block-validation signature verifier.
 *
 * This is the drop-in boundary for ONE path of trustbloc/fabric-mod (Hyperledger Fabric 2.2):
 *     msp/identities.go:169-196   identity.Verify  = bccsp.Hash(msg) ; bccsp.Verify(pk, sig, digest)
 *     bccsp/bccsp.go:90-134       the bccsp.BCCSP interface (Hash / Verify / KeyImport)
 *     bccsp/sw/impl.go:177-194    CSP.Hash         -> bccsp/sw/hash.go:29-33 (SHA-256)
 *     bccsp/sw/impl.go:247-270    CSP.Verify       -> bccsp/sw/ecdsa.go:41-57 verifyECDSA
 *     bccsp/utils/ecdsa.go:43-92  UnmarshalECDSASignature / IsLowS
 * A Go provider (fabric-mod_amd/go/bccsp/gpu, shown in INTEGRATION.md) embeds bccsp/sw exactly like
 * bccsp/pkcs11/pkcs11.go:35-52 does and binds the entry points below through cgo.  Nothing like this
 * ABI exists in the reference (it is 100 % Go); SURVEY.md section 8(b) fixes its shape.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ types, no callbacks, no exceptions cross this boundary;
 *   - every function returns FABGPU_OK (0) or a negative FABGPU_E* infrastructure error.  A non-zero
 *     return NEVER means "signature invalid": the Go side must then fall back to bccsp/sw, because
 *     verdicts are consensus-critical (core/committer/txvalidator/v20/validator.go:261);
 *   - signature verdicts are data: one bit per tuple in `verdict_bits` (bit i%64 of word i/64, 1 = the
 *     reference would return (true, nil)) and optionally one byte per tuple in `status`;
 *   - host-pointer entry points copy inputs into staging the library owns before returning (cgo
 *     pointer rules: nothing is retained); `_dev` entry points take HIP device pointers and a HIP
 *     stream and are asynchronous on that stream;
 *   - all big integers are 32-byte big-endian, struct-of-arrays: field[i] at base + 32*i.
 *   - a context is bound to one GPU; one context per GPU, calls on one context are serialised
 *     internally (bccsp.Verify is called from up to validatorPoolSize goroutines,
 *     core/committer/txvalidator/v20/validator.go:198-208).
 */
#ifndef FABGPU_H
#define FABGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FABGPU_ABI_VERSION 6

/* ---- return codes (infrastructure only) ---- */
#define FABGPU_OK 0
#define FABGPU_EINVAL (-1)     /* NULL / inconsistent arguments */
#define FABGPU_ENODEV (-2)     /* no usable gfx950 device / HIP runtime failure at init */
#define FABGPU_ENOMEM (-3)     /* host or device allocation failed */
#define FABGPU_ELAUNCH (-4)    /* kernel launch / execution / copy failed */
#define FABGPU_ETOOBIG (-5)    /* batch or arena larger than the ABI's 32-bit offsets allow */

/* ---- per-tuple status codes (data) ----
 * What bccsp/sw would have returned for the tuple, see SURVEY.md Appendix A:
 *   0  (true,  nil)
 *   1  (false, nil)  arithmetic reject, including the point at infinity
 *   2  (false, err)  "Invalid S. Must be smaller than half the order" (bccsp/sw/ecdsa.go:52-54)
 *   3  r or s is zero (error at bccsp/utils/ecdsa.go:59-64) or r >= n ((false, nil) inside ecdsa.Verify)
 *   4  public key is not a point of P-256: the reference can only reach this through
 *      ECDSAGoPublicKeyImportOpts (bccsp/sw/keyimport.go:103-112); the caller must use bccsp/sw. */
#define FABGPU_ST_VALID 0
#define FABGPU_ST_BAD_MATH 1
#define FABGPU_ST_HIGH_S 2
#define FABGPU_ST_RANGE 3
#define FABGPU_ST_OFF_CURVE 4

typedef struct fabgpu_ctx fabgpu_ctx;

/* fabgpu_cfg.flags */
#define FABGPU_FLAG_ONE_LANE_ONLY 1u /* never use the two-lanes-per-signature kernel (parity tests run both variants) */
#define FABGPU_FLAG_NO_QUAD 4u       /* idemix: never use the four-lanes-per-signature kernel (parity tests run all three variants) */
#define FABGPU_FLAG_TIME_KERNELS 2u  /* bracket every launch with timing events (tools only: read back through the test-hook library's fabgpu_last_kernel_ms) */
#define FABGPU_FLAG_PAIR_TABLE_LDS 8u    /* two-lanes-per-signature verify kernel: per-signature table in LDS (8 entries, 65 signed 4-bit windows) */
#define FABGPU_FLAG_PAIR_TABLE_GLOBAL 16u /* ... in the global workspace (16 entries, 52 signed 5-bit windows).  Neither: the default */
#define FABGPU_FLAG_NO_WIDE 32u          /* registered keys: never use the eight-lanes-per-signature, two-phase kernels that serve launches of up to
                                            8 192 signatures (parity tests run both forms) */

#define FABGPU_FLAG_NYM_FUSED_HASH 64u   /* idemix, four-lanes-per-signature form: hash the challenge on one lane of four inside the same kernel (round 4's
                                            form) instead of a second launch with eight lanes on a message (parity tests run both forms) */
#define FABGPU_FLAG_NYM_NO_SIDE_STREAM 128u /* idemix, four-lanes-per-signature form: compute the fixed-base terms inside the commitment kernel instead of
                                            a launch of their own on a second stream beside it (parity tests run both forms) */

#define FABGPU_FLAG_KEY_TABLES_16BIT 256u   /* registered keys (fabgpu_p256_key_register): each key also gets a 16-bit comb table - 80 MiB, the
                                            generator's format, built on the device BEHIND the registration (nobody waits for it), up to
                                            64 keys = 5 GiB of the device's 288 GB.  A wavefront all of whose keys have one computes
                                            u2*Q in 16 mixed additions instead of 32; any other wavefront uses the 8-bit combs as before.
                                            For a peer whose channels have a few dozen signers (msp/cache/cache.go:14-18) */

typedef struct fabgpu_cfg {
    int32_t device;      /* HIP device ordinal; -1 = the current device */
    uint32_t max_batch;  /* staging pre-allocation hint in tuples (0 = grow on demand) */
    uint32_t max_arena;  /* staging pre-allocation hint in message bytes (0 = grow on demand) */
    uint32_t flags;      /* FABGPU_FLAG_* bits, normally 0; unknown bits are rejected */
} fabgpu_cfg;

/* Lifecycle.  Replaces sw.NewWithParams (bccsp/sw/new.go:39-98) for the accelerated verbs. */
int fabgpu_init(const fabgpu_cfg* cfg, fabgpu_ctx** out);
void fabgpu_shutdown(fabgpu_ctx* ctx);
int fabgpu_device_count(fabgpu_ctx* ctx); /* ctx may be NULL; <0 on error */
const char* fabgpu_strerror(int code);
int fabgpu_abi_version(void);

/* Batched ecdsa.Verify with the bccsp/sw gates (bccsp/sw/ecdsa.go:41-57 after DER decoding).
 * e[i] is hashToInt(digest) left-padded to 32 bytes.  verdict_bits: ceil(n/64) words, required.
 * status: n bytes or NULL. */
int fabgpu_p256_verify_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* e,
                             const uint8_t* r, const uint8_t* s, uint64_t* verdict_bits, uint8_t* status);

/* Batched CSP.Hash(msg, SHA256Opts) (bccsp/sw/hash.go:29-33).  Message i = arena[off[i], off[i+1]);
 * off has n+1 entries, off[0] may be non-zero, messages may overlap / be empty. digests: n x 32 bytes. */
int fabgpu_sha256_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, uint8_t* digests);

/* identity.Verify over a flattened batch (msp/identities.go:169-196): SHA-256 fused ahead of the verify;
 * the digest never leaves the chip. */
int fabgpu_sha256_p256_verify_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off,
                                    const uint8_t* qx, const uint8_t* qy, const uint8_t* r, const uint8_t* s,
                                    uint64_t* verdict_bits, uint8_t* status);

/* Device-resident variants: all pointers are HIP device pointers (16-byte aligned), `stream` is a
 * hipStream_t (NULL = the null stream).  Asynchronous; the caller synchronises the stream.
 * arena_bytes = readable size of the arena allocation. */
int fabgpu_p256_verify_batch_dev(fabgpu_ctx* ctx, size_t n, const void* qx, const void* qy, const void* e, const void* r,
                                 const void* s, void* verdict_bits, void* status, void* stream);
int fabgpu_sha256_batch_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off,
                            void* digests, void* stream);
int fabgpu_sha256_p256_verify_batch_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off,
                                        const void* qx, const void* qy, const void* r, const void* s,
                                        void* verdict_bits, void* status, void* stream);
/* ---- registered public keys (bccsp.KeyImport, bccsp/bccsp.go:112; pattern bccsp/pkcs11/pkcs11.go:148-179) ----
 * A Fabric block is signed by few distinct identities (endorsers, orderers; msp/cache/cache.go:14-18 caches 100).
 * Registering a public key builds a 640 KiB comb table for it on the device once; batches that name registered keys by
 * id verify with 64 mixed additions and no doublings (about 3.5x fewer instructions than a fresh key).  Verdicts are
 * bit-identical to fabgpu_p256_verify_batch on the same (key, e, r, s).
 * fabgpu_p256_key_register: idempotent per (qx, qy); FABGPU_EINVAL for a point that is not on P-256 (the KeyImport gate:
 * such keys stay with bccsp/sw), FABGPU_ENOMEM when FABGPU_MAX_KEYS tables exist.  A key id out of range in a batch
 * yields status 4 for that tuple. */
#define FABGPU_MAX_KEYS 4096
int fabgpu_p256_key_register(fabgpu_ctx* ctx, const uint8_t* qx32, const uint8_t* qy32, uint32_t* key_id);
/* The same key on n contexts - a provider that drives every GPU of the node (fabgpu_csp_new2) keeps each registered key's table on
 * each of them: built once on the host, uploaded n times.  key_ids[g] = the key's id on ctxs[g]. */
int fabgpu_p256_key_register_many(fabgpu_ctx* const* ctxs, int n, const uint8_t* qx32, const uint8_t* qy32, uint32_t* key_ids);
int fabgpu_p256_key_lookup(fabgpu_ctx* ctx, const uint8_t* qx32, const uint8_t* qy32, uint32_t* key_id); /* 0 found, 1 not registered */
int fabgpu_p256_key_count(fabgpu_ctx* ctx);
int fabgpu_p256_verify_batch_keyed(fabgpu_ctx* ctx, size_t n, const uint32_t* key_id, const uint8_t* e, const uint8_t* r,
                                   const uint8_t* s, uint64_t* verdict_bits, uint8_t* status);
int fabgpu_p256_verify_batch_keyed_dev(fabgpu_ctx* ctx, size_t n, const void* key_id, const void* e, const void* r, const void* s,
                                       void* verdict_bits, void* status, void* stream);
/* identity.Verify (msp/identities.go:169-196) for registered keys: SHA-256 fused ahead of the keyed verify. */
int fabgpu_sha256_p256_verify_batch_keyed(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, const uint32_t* key_id,
                                          const uint8_t* r, const uint8_t* s, uint64_t* verdict_bits, uint8_t* status);
int fabgpu_sha256_p256_verify_batch_keyed_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off,
                                              const void* key_id, const void* r, const void* s, void* verdict_bits, void* status,
                                              void* stream);

/* ---- identity.Verify over a DESCRIBED batch: shared message prefixes, fresh or registered keys ----
 * The endorsements of one transaction all sign  prp || endorser_i  (core/common/validation/statebased/
 * validator_keylevel.go:246-258): the proposal-response payload is a shared PREFIX.  A batch may list m prefixes
 * (prefix p = arena[pre_off[p], pre_off[p+1])) and name one per message (pre_idx[i], 0xFFFFFFFF = none); message i is then
 * prefix || arena[off[i], off[i+1]).  The whole 64-byte blocks of every prefix are hashed once (SHA-256 mid-state), the
 * messages continue from there.  Keys: either (qx, qy) per message or key_id per message (registered keys), not both.
 * Verdicts are those of fabgpu_sha256_p256_verify_batch on the concatenated messages.
 * Host variant: all pointers are host memory.  _dev: all pointers are device memory, arena_bytes = readable size of the
 * arena allocation, mid_scratch = n_prefixes x 32 bytes of device scratch, asynchronous on `stream`. */
typedef struct fabgpu_identity_batch {
    size_t n;
    const void* arena;
    size_t arena_bytes;          /* _dev only */
    const uint32_t* off;         /* n + 1 */
    uint32_t n_prefixes;         /* 0 = no prefixes */
    const uint32_t* pre_off;     /* n_prefixes + 1 */
    const uint32_t* pre_idx;     /* n */
    const void* qx;              /* n x 32, or NULL with key_id */
    const void* qy;
    const uint32_t* key_id;      /* n, or NULL with qx / qy */
    const void* r;
    const void* s;
    void* verdict_bits;          /* ceil(n/64) x u64 */
    void* status;                /* n bytes or NULL */
    uint32_t flags;              /* FABGPU_IDB_* */
    /* Optional: more SHA-256 digests over the SAME arena in the same submission - the TxID and proposal-hash checks of
     * ValidateTransaction (protoutil/proputils.go:357-375, protoutil/txutils.go:431-447).  Message j is the
     * concatenation of up to three arena spans gather_spans[6j .. 6j+5] = (start, end) x 3 (an unused piece has start == end);
     * a gather kernel stitches them on the device, the batched SHA-256 kernel hashes them.  n_gather = 0: none.
     * _dev additionally needs gather_off (n_gather + 1 running offsets of the stitched messages) and gather_scratch
     * (gather_off[n_gather] + 64 bytes of device scratch); the host variant computes and owns both. */
    uint32_t n_gather;
    const uint32_t* gather_spans;    /* n_gather x 6 */
    void* gather_digests;            /* n_gather x 32 bytes, out */
    const uint32_t* gather_off;      /* _dev only */
    void* gather_scratch;            /* _dev only */
    size_t gather_scratch_bytes;     /* _dev only: size of the gather_scratch allocation */
    uint64_t stage_token;            /* host variant with FABGPU_IDB_ARENA_STAGED: the token fabgpu_arena_stage returned */
    /* ABI v3, host variant: a TAIL - bytes that are not in the caller's arena but belong to the batch.  The orderers' signatures over a
     * block sign  Metadata.value || signature_header || ASN.1(block header)  (internal/peer/gossip/mcs.go:166-193,
     * protoutil/blockutils.go:38-58): the host builds those few hundred bytes, and they ride in the same submission as the block's
     * other signatures.  Offsets >= tail_base address tail[offset - tail_base]; tail_base must be >= the end of every arena span used
     * and a multiple of 64; a span may not straddle tail_base.  tail == NULL or tail_len == 0: none.  (_dev callers simply place such
     * bytes in their device arena.) */
    const void* tail;
    uint32_t tail_base;
    uint32_t tail_len;
    /* ABI v3: optional out, n x 32 bytes - the SHA-256 of every message exactly as the fused kernel computed and verified it.
     * bccsp.Verify(k, signature, digest) is the question the Go validators ask later (msp/identities.go:188), so a verdict memo
     * must be keyed on this digest, not on a second hash of bytes some other parser extracted.  NULL = digests stay on the chip. */
    void* digests;
    /* ABI v4, host variant with FABGPU_IDB_SPANS: idemix pseudonym signatures over messages of the SAME arena ride in the submission - the
     * idemix creators of a block (msp/idemixmsp.go:584-599 -> idemix/nymsignature.go:74-109).  They run on a stream of their own next to
     * the ECDSA kernels: no second upload of their messages, no second blocking call.  Message i = arena[nym_off[2i], nym_off[2i+1]);
     * statuses and verdicts as fabgpu_idemix_nym_verify_batch.  n_nym = 0: none.  (_dev callers use fabgpu_idemix_nym_verify_batch_dev.) */
    uint32_t n_nym;
    const uint32_t* nym_off;         /* n_nym x 2 */
    const uint32_t* nym_issuer;      /* n_nym issuer ids (fabgpu_idemix_issuer_register) */
    const void* nym_fields;          /* 6 x n_nym x 32 bytes, column after column: nym_x | nym_y | proof_c | proof_s_sk | proof_s_r_nym | nonce */
    void* nym_verdict_bits;          /* ceil(n_nym / 64) x u64, out */
    void* nym_status;                /* n_nym bytes, out (FABGPU_NYM_*), or NULL */
} fabgpu_identity_batch;
#define FABGPU_IDB_SPANS 1u /* off holds n (start, end) pairs and pre_off n_prefixes pairs: messages are arbitrary sub-slices of the arena
                             (a marshalled block), not consecutive */
#define FABGPU_IDB_ARENA_STAGED 2u /* host variant: the arena is already on the device (fabgpu_arena_stage); `arena` is ignored, all
                                    offsets are offsets into the staged bytes */
/* Uploads `len` bytes to a context-owned device buffer and returns when they are there; *token names the upload.  Meant to run on
 * a helper thread WHILE the caller still prepares the batch that refers to these bytes (the block pre-verify pass parses a
 * 50 MB block while it travels).  A later fabgpu_identity_verify_batch with FABGPU_IDB_ARENA_STAGED and this token uses the
 * staged bytes; if another upload replaced them meanwhile the call returns FABGPU_EINVAL and the caller resubmits unstaged.  A context
 * keeps the three most recent uploads (channels validating at once), the least recent one that no batch is reading gives way. */
int fabgpu_arena_stage(fabgpu_ctx* ctx, const void* arena, size_t len, uint64_t* token);
int fabgpu_identity_verify_batch(fabgpu_ctx* ctx, const fabgpu_identity_batch* batch);
int fabgpu_identity_verify_batch_dev(fabgpu_ctx* ctx, const fabgpu_identity_batch* batch, void* mid_scratch, void* stream);

/* ---- idemix pseudonym signatures on FP256BN (BASELINE config 5) ----
 * NymSignature.Ver (idemix/nymsignature.go:74-109), reached per CREATOR signature from msp/idemixmsp.go:584-599 through
 * bccsp/idemix/handlers/nymsigner.go:62-95:  t = HSk*s_sk + HRand*s_rnym - Nym*c  (three G1 scalar multiplications, no
 * pairing), then two SHA-256:  valid <=> c == H(H("sign" || t || Nym || ipk.Hash || msg) mod r || nonce) mod r.
 *
 * An ISSUER (bccsp.IdemixIssuerPublicKeyImportOpts -> bccsp/idemix/handlers/issuer.go) is registered once: comb tables for
 * its two bases HSk and HRand (2 x 640 KiB on the device) and its ipk.Hash.  fabgpu_idemix_issuer_register is idempotent per
 * (HSk, HRand, hash); FABGPU_EINVAL when a base is not a point of G1 with coordinates < p, FABGPU_ENOMEM beyond
 * FABGPU_MAX_ISSUERS.
 *
 * Batch layout: SoA, 32-byte big-endian fields (nym_x, nym_y: the coordinates of the pseudonym as NewPublicNymFromBytes
 * splits them, bccsp/idemix/bridge/user.go:72-86; proof_c, proof_s_sk, proof_s_r_nym, nonce: the four fields of the
 * NymSignature message, idemix/idemix.proto), message i = arena[off[i], off[i+1]), issuer_id[i] (NULL = issuer 0 for all).
 * status: FABGPU_NYM_VALID, FABGPU_NYM_BAD_PROOF ("pseudonym signature invalid: zero-knowledge proof is invalid"), or
 * FABGPU_NYM_NEEDS_SW for inputs the device does not decide - Nym not on the curve or with a coordinate >= p (amcl turns
 * such input into the point at infinity), an s-value >= r, a commitment t at infinity, an unknown issuer id: ask bccsp/sw.
 * The verdict bit is set only for FABGPU_NYM_VALID. */
#define FABGPU_MAX_ISSUERS 64
#define FABGPU_NYM_VALID 0
#define FABGPU_NYM_BAD_PROOF 1
#define FABGPU_NYM_NEEDS_SW 6
int fabgpu_idemix_issuer_register(fabgpu_ctx* ctx, const uint8_t* hsk_x32, const uint8_t* hsk_y32, const uint8_t* hrand_x32,
                                  const uint8_t* hrand_y32, const uint8_t* ipk_hash32, uint32_t* issuer_id);
int fabgpu_idemix_issuer_count(fabgpu_ctx* ctx);
int fabgpu_idemix_nym_verify_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, const uint32_t* issuer_id,
                                   const uint8_t* nym_x, const uint8_t* nym_y, const uint8_t* proof_c, const uint8_t* proof_s_sk,
                                   const uint8_t* proof_s_r_nym, const uint8_t* nonce, uint64_t* verdict_bits, uint8_t* status);
int fabgpu_idemix_nym_verify_batch_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off,
                                       const void* issuer_id, const void* nym_x, const void* nym_y, const void* proof_c,
                                       const void* proof_s_sk, const void* proof_s_r_nym, const void* nonce, void* verdict_bits,
                                       void* status, void* stream);
/* 1 if (x, y) is a point of FP256BN's G1 (y^2 = x^3 + 3) with x, y < p, else 0  (pure CPU) */
int fabgpu_bn256_g1_on_curve(const uint8_t* x32, const uint8_t* y32);
/* ---- one batch, G devices (BASELINE.json configs[2]) ----
 * ONE process drives every GPU of the node: the batch is cut into G contiguous shards whose starts are multiples of 64 tuples
 * (verify-only: equal counts; hash mode: equal message BYTES), every shard is uploaded from pinned staging and verified on its device
 * on that device's own stream, then ONE ncclAllGather (RCCL over xGMI) of the per-shard verdict words leaves the merged bitmap resident
 * on EVERY device (fabgpu_multi_merged_bitmap_dev: for an on-device policy step) and a single D2H from device 0 returns it.  No other
 * collective: signatures are independent.  Verdicts and statuses are those of fabgpu_p256_verify_batch / fabgpu_sha256_p256_verify_batch
 * on the whole batch.  devices == NULL: ordinals 0 .. n_devices-1.  RCCL is bound at run time (dlopen); FABGPU_MULTI_HOST_MERGE replaces
 * the collective by G small D2H copies (and is what allows the same ordinal to appear twice: several shards on one device).
 * Note: a block that fits one GPU is not made faster by more GPUs - its latency is one wavefront's instruction
 * stream; this entry point is for batches beyond one GPU's saturation point, N blocks in flight are N contexts. */
typedef struct fabgpu_multi fabgpu_multi;
#define FABGPU_MULTI_HOST_MERGE 1u
/* bits 8-15 of `flags`: the deadline, in seconds, of the one-word all-gather self-check fabgpu_multi_init runs (0: 10 s).  Init blocks
 * for at most that long on a node whose collective never completes, then merges on the host. */
#define FABGPU_MULTI_SELFCHECK_SECONDS(s) (((uint32_t)(s) & 0xFFu) << 8)
int fabgpu_multi_init(const int32_t* devices, int n_devices, uint32_t flags, fabgpu_multi** out);
void fabgpu_multi_shutdown(fabgpu_multi* m);
int fabgpu_multi_device_count(fabgpu_multi* m);
int fabgpu_multi_p256_verify_batch(fabgpu_multi* m, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* e, const uint8_t* r,
                                   const uint8_t* s, uint64_t* verdict_bits, uint8_t* status);
int fabgpu_multi_sha256_p256_verify_batch(fabgpu_multi* m, size_t n, const uint8_t* arena, const uint32_t* off, const uint8_t* qx,
                                          const uint8_t* qy, const uint8_t* r, const uint8_t* s, uint64_t* verdict_bits, uint8_t* status);
/* the shard boundaries [lo[g], hi[g]) a batch of n tuples gets on n_devices devices (pure host): off == NULL by count, else by bytes */
int fabgpu_multi_plan(size_t n, const uint32_t* off, uint32_t n_devices, uint64_t* lo, uint64_t* hi, uint64_t* words_per_rank);
const void* fabgpu_multi_merged_bitmap_dev(fabgpu_multi* m, int g);
/* How the shard bitmaps are merged.  fabgpu_multi_init never fails for want of a collective: one device, a repeated ordinal, no
 * librccl, ncclCommInitAll failing, or the one-word ncclAllGather self-check it runs across its devices failing (wrong word, error,
 * or no completion within the deadline of FABGPU_MULTI_SELFCHECK_SECONDS, 10 s by default) all end in the host merge (G small D2H
 * copies), and so does an ncclAllGather error on a later batch - the communicators are then abandoned (never used or
 * destroyed again), the shards move to fresh streams, and fabgpu_multi_shutdown leaves the device buffers to process exit instead of
 * waiting for a collective that may never finish.  Returns the number of RCCL ranks (G) when the merge is the
 * collective, 0 when the host merges; `why` (optional, NUL-terminated) says which. */
int fabgpu_multi_collective(fabgpu_multi* m, char* why, size_t why_cap);

/* ---- host-side gates the Go provider calls before marshalling a tuple (pure CPU, no device) ---- */

/* utils.UnmarshalECDSASignature (bccsp/utils/ecdsa.go:43-67) with Go encoding/asn1 strictness.
 * Returns 0 ok; 1 asn1 failure; 2 R <= 0; 3 S <= 0.  r32/s32: low 256 bits big-endian;
 * *flags bit0: R >= 2^256, bit1: S >= 2^256 (such values can only fail the range / low-S gates). */
int fabgpu_ecdsa_unmarshal_signature(const uint8_t* sig, size_t len, uint8_t* r32, uint8_t* s32, int* flags);
/* utils.IsLowS (bccsp/utils/ecdsa.go:84-92): 1 if s <= n>>1 else 0. */
int fabgpu_ecdsa_is_low_s(const uint8_t* s32);
/* Key-import gate (what x509.ParseCertificate guarantees for bccsp/sw/keyimport.go:114-134 keys):
 * 1 if (x,y) is a point of P-256 with x,y < p, else 0. */
int fabgpu_p256_pubkey_on_curve(const uint8_t* qx32, const uint8_t* qy32);
/* hashToInt (Go crypto/ecdsa): leftmost 32 bytes of the digest, left-padded with zeros into e32. */
void fabgpu_hash_to_int(const uint8_t* digest, size_t len, uint8_t* e32);

#ifdef __cplusplus
}
#endif
#endif /* FABGPU_H */
