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

/* status as include/fabgpu.h: 0 valid, 1 bad math, 2 high-S, 3 range, 4 off-curve */
static int one(const EC_GROUP *grp, const uint8_t *qx, const uint8_t *qy, const uint8_t *e, const uint8_t *r, const uint8_t *s) {
    static const uint8_t zero[32] = {0};
    if (!memcmp(r, zero, 32) || !memcmp(s, zero, 32)) return 3;
    if (memcmp(s, HALF_N, 32) > 0) return 2;
    int st = 1;
    EC_KEY *key = EC_KEY_new();
    EC_KEY_set_group(key, grp);
    BIGNUM *x = BN_bin2bn(qx, 32, NULL), *y = BN_bin2bn(qy, 32, NULL);
    if (EC_KEY_set_public_key_affine_coordinates(key, x, y) != 1) st = 4;
    else {
        ECDSA_SIG *sig = ECDSA_SIG_new();
        ECDSA_SIG_set0(sig, BN_bin2bn(r, 32, NULL), BN_bin2bn(s, 32, NULL));
        const BIGNUM *order = EC_GROUP_get0_order(grp);
        const BIGNUM *br = ECDSA_SIG_get0_r(sig);
        if (BN_cmp(br, order) >= 0) st = 3;
        else st = ECDSA_do_verify(e, 32, sig, key) == 1 ? 0 : 1;
        ECDSA_SIG_free(sig);
    }
    BN_free(x); BN_free(y); EC_KEY_free(key);
    return st;
}

void ossl_p256_verify_batch(size_t n, const uint8_t *qx, const uint8_t *qy, const uint8_t *e, const uint8_t *r,
                            const uint8_t *s, uint8_t *status) {
#pragma omp parallel
    {
        EC_GROUP *grp = EC_GROUP_new_by_curve_name(NID_X9_62_prime256v1);
#pragma omp for schedule(dynamic, 64)
        for (long i = 0; i < (long)n; i++)
            status[i] = (uint8_t)one(grp, qx + 32 * i, qy + 32 * i, e + 32 * i, r + 32 * i, s + 32 * i);
        EC_GROUP_free(grp);
    }
}

void ossl_sha256_p256_verify_batch(size_t n, const uint8_t *arena, const uint32_t *off, const uint8_t *qx, const uint8_t *qy,
                                   const uint8_t *r, const uint8_t *s, uint8_t *status) {
#pragma omp parallel
    {
        EC_GROUP *grp = EC_GROUP_new_by_curve_name(NID_X9_62_prime256v1);
#pragma omp for schedule(dynamic, 64)
        for (long i = 0; i < (long)n; i++) {
            uint8_t d[32];
            SHA256(arena + off[i], off[i + 1] - off[i], d);
            status[i] = (uint8_t)one(grp, qx + 32 * i, qy + 32 * i, d, r + 32 * i, s + 32 * i);
        }
        EC_GROUP_free(grp);
    }
}
