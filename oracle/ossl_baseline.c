/*
 * CPU BASELINE PROXY (test/bench infrastructure, NOT product code).
 *
 * The reference's own CPU path is Go (bccsp/sw -> Go 1.14 crypto/ecdsa.Verify, P-256 amd64 assembly) and
 * no Go toolchain exists in this image, so oracle/_ref cannot be built (DESIGN.md "Oracle").  BASELINE.md
 * section 3 names the stand-in: OpenSSL 3 libcrypto (nistz256 assembly, the same algorithm family as Go's
 * p256_asm) driven exactly like msp/identities.go:169-196 drives bccsp/sw:
 *     digest = SHA-256(msg); DER-free tuple -> low-S gate (bccsp/sw/ecdsa.go:47-54) -> ECDSA_do_verify.
 * One worker per host core mirrors peer.validatorPoolSize = runtime.NumCPU() (core/peer/config.go:255-257).
 * Labelled everywhere as "proxy for bccsp/sw - Go toolchain absent".
 */
#include <openssl/bn.h>
#include <openssl/ec.h>
#include <openssl/ecdsa.h>
#include <openssl/obj_mac.h>
#include <openssl/sha.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static const uint8_t HALF_N[32] = {0x7f, 0xff, 0xff, 0xff, 0x80, 0x00, 0x00, 0x00, 0x7f, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
                                   0xde, 0x73, 0x7d, 0x56, 0xd3, 0x8b, 0xcf, 0x42, 0x79, 0xdc, 0xe5, 0x61, 0x7e, 0x31, 0x92, 0xa8};

/* Per-thread verifier state: everything OpenSSL would otherwise allocate per tuple (EC_KEY, EC_POINT, BIGNUMs, BN_CTX, ECDSA_SIG) is
 * made once per worker and reused - round 1 built an EC_KEY per tuple with EC_KEY_set_public_key_affine_coordinates (which runs
 * EC_KEY_check_key: a full extra scalar multiplication by n) inside a 14 ms parallel region, and 256 threads delivered 8 cores' worth. */
typedef struct {
    EC_GROUP *grp;
    EC_KEY *key;
    EC_POINT *pt;
    BIGNUM *x, *y, *r, *s;
    BN_CTX *bn;
    ECDSA_SIG *sig;
} worker;

static void worker_init(worker *w) {
    w->grp = EC_GROUP_new_by_curve_name(NID_X9_62_prime256v1);
    w->key = EC_KEY_new();
    EC_KEY_set_group(w->key, w->grp);
    w->pt = EC_POINT_new(w->grp);
    w->x = BN_new(); w->y = BN_new();
    w->bn = BN_CTX_new();
    w->sig = ECDSA_SIG_new();
    w->r = BN_new(); w->s = BN_new();
    ECDSA_SIG_set0(w->sig, w->r, w->s);          /* the signature object owns r and s; they are overwritten in place per tuple */
}
static void worker_free(worker *w) {
    ECDSA_SIG_free(w->sig);
    BN_CTX_free(w->bn);
    BN_free(w->x); BN_free(w->y);
    EC_POINT_free(w->pt);
    EC_KEY_free(w->key);
    EC_GROUP_free(w->grp);
}

/* status as include/fabgpu.h: 0 valid, 1 bad math, 2 high-S, 3 range, 4 off-curve.
 * bccsp/sw order of the gates: r, s > 0 (utils/ecdsa.go:59-64), low-S (sw/ecdsa.go:47-54), then ecdsa.Verify (r < n inside). */
static int one(worker *w, const uint8_t *qx, const uint8_t *qy, const uint8_t *e, const uint8_t *r, const uint8_t *s) {
    static const uint8_t zero[32] = {0};
    if (!memcmp(r, zero, 32) || !memcmp(s, zero, 32)) return 3;
    if (memcmp(s, HALF_N, 32) > 0) return 2;
    BN_bin2bn(qx, 32, w->x);
    BN_bin2bn(qy, 32, w->y);
    /* on-curve check only (what x509.ParseCertificate guarantees for the reference's keys) - no EC_KEY_check_key */
    if (EC_POINT_set_affine_coordinates(w->grp, w->pt, w->x, w->y, w->bn) != 1) return 4;
    if (EC_KEY_set_public_key(w->key, w->pt) != 1) return 4;
    BN_bin2bn(r, 32, w->r);
    BN_bin2bn(s, 32, w->s);
    if (BN_cmp(w->r, EC_GROUP_get0_order(w->grp)) >= 0) return 3;
    return ECDSA_do_verify(e, 32, w->sig, w->key) == 1 ? 0 : 1;
}

void ossl_p256_verify_batch(size_t n, const uint8_t *qx, const uint8_t *qy, const uint8_t *e, const uint8_t *r,
                            const uint8_t *s, uint8_t *status) {
#pragma omp parallel
    {
        worker w;
        worker_init(&w);
#pragma omp for schedule(dynamic, 64)
        for (long i = 0; i < (long)n; i++)
            status[i] = (uint8_t)one(&w, qx + 32 * i, qy + 32 * i, e + 32 * i, r + 32 * i, s + 32 * i);
        worker_free(&w);
    }
}

/* The timed leg of bench.py's cpu_baseline: `reps` passes over the same n tuples on exactly `threads` workers inside ONE parallel
 * region (thread start-up and per-worker allocation stay outside the clock), statically chunked like validatorPoolSize goroutines
 * draining a block.  Returns the wall seconds of the slowest worker between two barriers; status is written on the last pass. */
#include <omp.h>
double ossl_p256_verify_timed(size_t n, const uint8_t *qx, const uint8_t *qy, const uint8_t *e, const uint8_t *r, const uint8_t *s,
                              uint8_t *status, int threads, int reps) {
    /* the clock: from the EARLIEST worker leaving the start barrier to the LATEST worker finishing - with more threads than the container
     * owns CPUs a single "master" thread may be descheduled for milliseconds right after the barrier and start its clock late (a 256-thread
     * run under a 16-CPU quota once "measured" 6 M verifies/s that way) */
    double t_first = 1e300, t_last = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        worker w;
        worker_init(&w);
        /* warm: one tuple per worker touches every lazily initialised table of libcrypto */
        if (n) (void)one(&w, qx, qy, e, r, s);
#pragma omp barrier
        double t0 = omp_get_wtime();
        for (int rep = 0; rep < reps; rep++) {
#pragma omp for schedule(dynamic, 16) nowait
            for (long i = 0; i < (long)n; i++)
                status[i] = (uint8_t)one(&w, qx + 32 * i, qy + 32 * i, e + 32 * i, r + 32 * i, s + 32 * i);
        }
        double t1 = omp_get_wtime();
#pragma omp critical
        {
            if (t0 < t_first) t_first = t0;
            if (t1 > t_last) t_last = t1;
        }
        worker_free(&w);
    }
    return t_last - t_first;
}

void ossl_sha256_p256_verify_batch(size_t n, const uint8_t *arena, const uint32_t *off, const uint8_t *qx, const uint8_t *qy,
                                   const uint8_t *r, const uint8_t *s, uint8_t *status) {
#pragma omp parallel
    {
        worker w;
        worker_init(&w);
#pragma omp for schedule(dynamic, 64)
        for (long i = 0; i < (long)n; i++) {
            uint8_t d[32];
            SHA256(arena + off[i], off[i + 1] - off[i], d);
            status[i] = (uint8_t)one(&w, qx + 32 * i, qy + 32 * i, d, r + 32 * i, s + 32 * i);
        }
        worker_free(&w);
    }
}

/* The CPU side of BASELINE's SECOND metric (validated tx/s per block): identity.Verify (msp/identities.go:169-196) for every tuple of
 * a marshalled block, as validatorPoolSize goroutines would drain it - per tuple  digest = SHA-256(prefix || suffix)  (an endorsement
 * signs prp || endorser, a creator the envelope payload), the DER signature unmarshalled (bccsp/utils/ecdsa.go:43-67), the low-S gate,
 * ECDSA_do_verify.  spans6[6 i ..] = (prefix off, len, suffix off, len, signature off, len) into `arena`; qxy = X || Y per tuple.
 * `reps` passes on exactly `threads` workers inside one parallel region; returns the wall seconds between the earliest start and the
 * latest finish.  status: 0 valid, 1 reject, 2 high-S, 3 range, 4 off-curve, 5 signature does not unmarshal. */
double ossl_identity_verify_spans_timed(size_t n, const uint8_t *arena, const uint32_t *spans6, const uint8_t *qxy, uint8_t *status,
                                        int threads, int reps) {
    double t_first = 1e300, t_last = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        worker w;
        worker_init(&w);
#pragma omp barrier
        double t0 = omp_get_wtime();
        for (int rep = 0; rep < reps; rep++) {
#pragma omp for schedule(dynamic, 16) nowait
            for (long i = 0; i < (long)n; i++) {
                const uint32_t *sp = spans6 + 6 * i;
                uint8_t d[32], r32[32], s32[32];
                SHA256_CTX c;
                SHA256_Init(&c);
                if (sp[1]) SHA256_Update(&c, arena + sp[0], sp[1]);
                if (sp[3]) SHA256_Update(&c, arena + sp[2], sp[3]);
                SHA256_Final(d, &c);
                const unsigned char *p = arena + sp[4];
                ECDSA_SIG *sg = d2i_ECDSA_SIG(NULL, &p, (long)sp[5]);
                int st = 5;
                if (sg) {
                    const BIGNUM *r = NULL, *s = NULL;
                    ECDSA_SIG_get0(sg, &r, &s);
                    if (BN_num_bytes(r) <= 32 && BN_num_bytes(s) <= 32 && !BN_is_negative(r) && !BN_is_negative(s)) {
                        BN_bn2binpad(r, r32, 32);
                        BN_bn2binpad(s, s32, 32);
                        st = one(&w, qxy + 64 * i, qxy + 64 * i + 32, d, r32, s32);
                    } else {
                        st = BN_num_bytes(s) > 32 ? 2 : 3;
                    }
                    ECDSA_SIG_free(sg);
                }
                status[i] = (uint8_t)st;
            }
        }
        double t1 = omp_get_wtime();
#pragma omp critical
        {
            if (t0 < t_first) t_first = t0;
            if (t1 > t_last) t_last = t1;
        }
        worker_free(&w);
    }
    return t_last - t_first;
}
