/*
 * CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's block-validation signature path, fast enough to check
 * BASELINE-sized batches (30 000 - 300 000 tuples) in seconds:
 *
 *   msp/identities.go:169-196          identity.Verify = Hash(msg); Verify(pk, sig, digest)
 *   bccsp/sw/hash.go:29-33             SHA-256                               -> oracle_sha256*
 *   bccsp/sw/impl.go:247-270           CSP.Verify argument checks            -> oracle_bccsp_verify
 *   bccsp/sw/ecdsa.go:41-57            verifyECDSA (DER, low-S, ecdsa.Verify)-> oracle_bccsp_verify
 *   bccsp/utils/ecdsa.go:43-67         UnmarshalECDSASignature               -> oracle_der_unmarshal
 *   bccsp/utils/ecdsa.go:84-92,27-33   IsLowS / curveHalfOrders              -> low-S gate below
 *
 * The arithmetic is in a dependency that is not under /root/reference: Go 1.14.4 standard library
 * (reference Makefile:79) crypto/ecdsa.Verify + crypto/elliptic P-256 + crypto/sha256 +
 * encoding/asn1.  Its published algorithms (FIPS 186-4 6.4.2, FIPS 180-4, X.690 DER as enforced
 * by Go's asn1.go) are restated here exactly as SURVEY.md Appendix A lists them.
 *
 * This file is pinned by tests/test_oracle_golden.py against the reference's certificate
 * fixtures, its literal DER vectors, the pure-Python restatement and OpenSSL.  It deliberately
 * shares no code with fabric-mod_amd/csrc: generic 4x64-bit Montgomery arithmetic, textbook
 * Jacobian formulas with explicit exceptional-case handling, bit-at-a-time double-and-add.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } bn; /* little-endian limbs */

/* ---- constants -------------------------------------------------------------------------- */
static const bn P256_P = {{0xFFFFFFFFFFFFFFFFull, 0x00000000FFFFFFFFull, 0x0000000000000000ull, 0xFFFFFFFF00000001ull}};
static const bn P256_N = {{0xF3B9CAC2FC632551ull, 0xBCE6FAADA7179E84ull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFF00000000ull}};
static const bn P256_B = {{0x3BCE3C3E27D2604Bull, 0x651D06B0CC53B0F6ull, 0xB3EBBD55769886BCull, 0x5AC635D8AA3A93E7ull}};
static const bn P256_GX = {{0xF4A13945D898C296ull, 0x77037D812DEB33A0ull, 0xF8BCE6E563A440F2ull, 0x6B17D1F2E12C4247ull}};
static const bn P256_GY = {{0xCBB6406837BF51F5ull, 0x2BCE33576B315ECEull, 0x8EE7EB4A7C0F9E16ull, 0x4FE342E2FE1A7F9Bull}};
/* n >> 1, bccsp/utils/ecdsa.go:27-33 */
static const bn P256_HALF_N = {{0x79DCE5617E3192A8ull, 0xDE737D56D38BCF42ull, 0x7FFFFFFFFFFFFFFFull, 0x7FFFFFFF80000000ull}};

enum { ST_VALID = 0, ST_BAD_MATH = 1, ST_HIGH_S = 2, ST_RANGE = 3, ST_OFF_CURVE = 4 };

/* ---- 256-bit helpers -------------------------------------------------------------------- */
static int bn_cmp(const bn *a, const bn *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->v[i] < b->v[i]) return -1;
        if (a->v[i] > b->v[i]) return 1;
    }
    return 0;
}
static int bn_is_zero(const bn *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static uint64_t bn_add(bn *r, const bn *a, const bn *b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; r->v[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t bn_sub(bn *r, const bn *a, const bn *b) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - borrow;
        r->v[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static void bn_from_be(bn *r, const uint8_t *be, size_t len) { /* len <= 32 */
    memset(r, 0, sizeof *r);
    for (size_t i = 0; i < len; i++) {
        size_t bit = (len - 1 - i) * 8;
        r->v[bit / 64] |= (uint64_t)be[i] << (bit % 64);
    }
}
static void bn_to_be(uint8_t *be, const bn *a) {
    for (int i = 0; i < 32; i++) be[i] = (uint8_t)(a->v[(31 - i) / 8] >> (((31 - i) % 8) * 8));
}
static int bn_bit(const bn *a, int i) { return (a->v[i / 64] >> (i % 64)) & 1; }

/* ---- generic modular arithmetic for an odd 256-bit modulus (Montgomery, R = 2^256) ---------- */
typedef struct { bn m; uint64_t m0inv; bn rr; bn one; } modctx;

static void mod_add(bn *r, const bn *a, const bn *b, const modctx *M) {
    bn t; uint64_t c = bn_add(&t, a, b);
    bn u; uint64_t br = bn_sub(&u, &t, &M->m);
    *r = (c || !br) ? u : t;
}
static void mod_sub(bn *r, const bn *a, const bn *b, const modctx *M) {
    bn t; if (bn_sub(&t, a, b)) bn_add(&t, &t, &M->m);
    *r = t;
}
/* r = a*b/R mod m (CIOS) */
static void mont_mul(bn *r, const bn *a, const bn *b, const modctx *M) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * M->m0inv;
        c = (u128)q * M->m.v[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)q * M->m.v[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    bn x = {{t[0], t[1], t[2], t[3]}}, y;
    uint64_t br = bn_sub(&y, &x, &M->m);
    *r = (t[4] || !br) ? y : x;
}
static void modctx_init(modctx *M, const bn *m) {
    M->m = *m;
    uint64_t inv = 1; /* Newton: inv = m^-1 mod 2^64 */
    for (int i = 0; i < 6; i++) inv *= 2 - m->v[0] * inv;
    M->m0inv = (uint64_t)0 - inv;
    /* one = R mod m, rr = R^2 mod m by repeated doubling */
    bn x = {{1, 0, 0, 0}};
    for (int i = 0; i < 256; i++) mod_add(&x, &x, &x, M);
    M->one = x;
    for (int i = 0; i < 256; i++) mod_add(&x, &x, &x, M);
    M->rr = x;
}
static void to_mont(bn *r, const bn *a, const modctx *M) { mont_mul(r, a, &M->rr, M); }
static void from_mont(bn *r, const bn *a, const modctx *M) { bn one = {{1, 0, 0, 0}}; mont_mul(r, a, &one, M); }
/* a^(m-2) in the Montgomery domain (m prime) */
static void mont_inv(bn *r, const bn *a, const modctx *M) {
    bn e = M->m; bn two = {{2, 0, 0, 0}}; bn_sub(&e, &e, &two);
    bn acc = M->one;
    for (int i = 255; i >= 0; i--) {
        mont_mul(&acc, &acc, &acc, M);
        if (bn_bit(&e, i)) mont_mul(&acc, &acc, a, M);
    }
    *r = acc;
}
/* reduce an arbitrary 256-bit value below m (m > 2^255 for both p and n) */
static void mod_reduce_once(bn *r, const bn *a, const modctx *M) {
    bn t; *r = bn_sub(&t, a, &M->m) ? *a : t;
}

static modctx CP, CN;
static bn MB, MGX, MGY, MTHREE; /* Montgomery forms mod p */
static int g_init_done;
static void oracle_init_once(void) {
    if (g_init_done) return;
    modctx_init(&CP, &P256_P);
    modctx_init(&CN, &P256_N);
    to_mont(&MB, &P256_B, &CP); to_mont(&MGX, &P256_GX, &CP); to_mont(&MGY, &P256_GY, &CP);
    bn three = {{3, 0, 0, 0}}; to_mont(&MTHREE, &three, &CP);
    g_init_done = 1;
}
__attribute__((constructor)) static void oracle_ctor(void) { oracle_init_once(); }

/* ---- Jacobian points over F_p, Montgomery domain; inf <=> Z == 0 ------------------------------ */
typedef struct { bn X, Y, Z; } jpt;

static void jpt_set_inf(jpt *r) { memset(r, 0, sizeof *r); r->X = CP.one; r->Y = CP.one; }
static int jpt_is_inf(const jpt *a) { return bn_is_zero(&a->Z); }

/* dbl-2001-b (a = -3) */
static void jpt_double(jpt *r, const jpt *a) {
    if (jpt_is_inf(a) || bn_is_zero(&a->Y)) { jpt_set_inf(r); return; }
    bn delta, gamma, beta, alpha, t1, t2, X3, Y3, Z3;
    mont_mul(&delta, &a->Z, &a->Z, &CP);
    mont_mul(&gamma, &a->Y, &a->Y, &CP);
    mont_mul(&beta, &a->X, &gamma, &CP);
    mod_sub(&t1, &a->X, &delta, &CP); mod_add(&t2, &a->X, &delta, &CP);
    mont_mul(&alpha, &t1, &t2, &CP); mod_add(&t1, &alpha, &alpha, &CP); mod_add(&alpha, &t1, &alpha, &CP);
    mont_mul(&X3, &alpha, &alpha, &CP);
    mod_add(&t1, &beta, &beta, &CP); mod_add(&t1, &t1, &t1, &CP);      /* 4 beta */
    mod_add(&t2, &t1, &t1, &CP);                                        /* 8 beta */
    mod_sub(&X3, &X3, &t2, &CP);
    mod_add(&Z3, &a->Y, &a->Z, &CP); mont_mul(&Z3, &Z3, &Z3, &CP);
    mod_sub(&Z3, &Z3, &gamma, &CP); mod_sub(&Z3, &Z3, &delta, &CP);
    mod_sub(&t1, &t1, &X3, &CP); mont_mul(&Y3, &alpha, &t1, &CP);
    mont_mul(&t2, &gamma, &gamma, &CP); mod_add(&t2, &t2, &t2, &CP); mod_add(&t2, &t2, &t2, &CP); mod_add(&t2, &t2, &t2, &CP);
    mod_sub(&Y3, &Y3, &t2, &CP);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
/* add-2007-bl style general addition with every exceptional case handled */
static void jpt_add(jpt *r, const jpt *a, const jpt *b) {
    if (jpt_is_inf(a)) { *r = *b; return; }
    if (jpt_is_inf(b)) { *r = *a; return; }
    bn z1z1, z2z2, u1, u2, s1, s2, h, rr, t;
    mont_mul(&z1z1, &a->Z, &a->Z, &CP); mont_mul(&z2z2, &b->Z, &b->Z, &CP);
    mont_mul(&u1, &a->X, &z2z2, &CP);   mont_mul(&u2, &b->X, &z1z1, &CP);
    mont_mul(&t, &b->Z, &z2z2, &CP);    mont_mul(&s1, &a->Y, &t, &CP);
    mont_mul(&t, &a->Z, &z1z1, &CP);    mont_mul(&s2, &b->Y, &t, &CP);
    mod_sub(&h, &u2, &u1, &CP); mod_sub(&rr, &s2, &s1, &CP);
    if (bn_is_zero(&h)) {
        if (bn_is_zero(&rr)) { jpt_double(r, a); return; }
        jpt_set_inf(r); return;
    }
    bn hh, hhh, v, X3, Y3, Z3;
    mont_mul(&hh, &h, &h, &CP); mont_mul(&hhh, &hh, &h, &CP); mont_mul(&v, &u1, &hh, &CP);
    mont_mul(&X3, &rr, &rr, &CP); mod_sub(&X3, &X3, &hhh, &CP); mod_sub(&X3, &X3, &v, &CP); mod_sub(&X3, &X3, &v, &CP);
    mod_sub(&t, &v, &X3, &CP); mont_mul(&Y3, &rr, &t, &CP); mont_mul(&t, &s1, &hhh, &CP); mod_sub(&Y3, &Y3, &t, &CP);
    mont_mul(&Z3, &a->Z, &b->Z, &CP); mont_mul(&Z3, &Z3, &h, &CP);
    r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void jpt_mul(jpt *r, const bn *k, const jpt *p) {
    jpt acc; jpt_set_inf(&acc);
    for (int i = 255; i >= 0; i--) {
        jpt_double(&acc, &acc);
        if (bn_bit(k, i)) jpt_add(&acc, &acc, p);
    }
    *r = acc;
}
/* returns 0 for infinity; x,y in plain (non-Montgomery) form */
static int jpt_to_affine(bn *x, bn *y, const jpt *a) {
    if (jpt_is_inf(a)) return 0;
    bn zi, zi2, zi3, t;
    mont_inv(&zi, &a->Z, &CP); mont_mul(&zi2, &zi, &zi, &CP); mont_mul(&zi3, &zi2, &zi, &CP);
    mont_mul(&t, &a->X, &zi2, &CP); from_mont(x, &t, &CP);
    mont_mul(&t, &a->Y, &zi3, &CP); from_mont(y, &t, &CP);
    return 1;
}
static int on_curve_plain(const bn *x, const bn *y) {
    if (bn_cmp(x, &P256_P) >= 0 || bn_cmp(y, &P256_P) >= 0) return 0;
    bn mx, my, l, r, t;
    to_mont(&mx, x, &CP); to_mont(&my, y, &CP);
    mont_mul(&l, &my, &my, &CP);
    mont_mul(&r, &mx, &mx, &CP); mont_mul(&r, &r, &mx, &CP);
    mont_mul(&t, &mx, &MTHREE, &CP); mod_sub(&r, &r, &t, &CP); mod_add(&r, &r, &MB, &CP);
    return bn_cmp(&l, &r) == 0;
}

/* ---- ecdsa.Verify (Go 1.14 crypto/ecdsa) on plain big-endian ---- */
/* r_len/s_len > 32 encode "value >= 2^256" (only reachable through DER). digest: hashToInt. */
static int verify_tuple(const uint8_t *qx_be, const uint8_t *qy_be, const uint8_t *digest, size_t dlen,
                        const bn *r, int r_oversize, const bn *s, int s_oversize) {
    /* sign checks happened at DER level / caller: zero is "not > 0" */
    if ((!r_oversize && bn_is_zero(r)) || (!s_oversize && bn_is_zero(s))) return ST_RANGE;
    /* bccsp/sw/ecdsa.go:47-54 low-S gate comes before ecdsa.Verify */
    if (s_oversize || bn_cmp(s, &P256_HALF_N) > 0) return ST_HIGH_S;
    if (r_oversize || bn_cmp(r, &P256_N) >= 0) return ST_RANGE;
    bn qx, qy; bn_from_be(&qx, qx_be, 32); bn_from_be(&qy, qy_be, 32);
    if (!on_curve_plain(&qx, &qy)) return ST_OFF_CURVE;
    bn e; if (dlen > 32) dlen = 32; bn_from_be(&e, digest, dlen);
    /* w = s^-1 mod n; u1 = e w; u2 = r w */
    bn ms, mw, me, mr, u1, u2, ered;
    to_mont(&ms, s, &CN); mont_inv(&mw, &ms, &CN);
    mod_reduce_once(&ered, &e, &CN);
    to_mont(&me, &ered, &CN); to_mont(&mr, r, &CN);
    mont_mul(&u1, &me, &mw, &CN); from_mont(&u1, &u1, &CN);
    mont_mul(&u2, &mr, &mw, &CN); from_mont(&u2, &u2, &CN);
    jpt G, Q, A, B, S;
    G.X = MGX; G.Y = MGY; G.Z = CP.one;
    to_mont(&Q.X, &qx, &CP); to_mont(&Q.Y, &qy, &CP); Q.Z = CP.one;
    jpt_mul(&A, &u1, &G); jpt_mul(&B, &u2, &Q); jpt_add(&S, &A, &B);
    bn x, y;
    if (!jpt_to_affine(&x, &y, &S)) return ST_BAD_MATH;       /* x == 0 && y == 0 */
    bn v; mod_reduce_once(&v, &x, &CN);                         /* x mod n, x < p < 2n */
    return bn_cmp(&v, r) == 0 ? ST_VALID : ST_BAD_MATH;
}

/* Flattened-tuple boundary: n x 32-byte big-endian SoA fields (include/fabgpu.h layout). */
void oracle_p256_verify_batch(size_t n, const uint8_t *qx, const uint8_t *qy, const uint8_t *e,
                              const uint8_t *r, const uint8_t *s, uint8_t *status) {
    oracle_init_once();
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < (long)n; i++) {
        bn br, bs; bn_from_be(&br, r + 32 * i, 32); bn_from_be(&bs, s + 32 * i, 32);
        status[i] = (uint8_t)verify_tuple(qx + 32 * i, qy + 32 * i, e + 32 * i, 32, &br, 0, &bs, 0);
    }
}
/* one tuple with an arbitrary-length digest (hashToInt cases) */
int oracle_p256_verify_one(const uint8_t *qx, const uint8_t *qy, const uint8_t *digest, size_t dlen,
                           const uint8_t *r, const uint8_t *s) {
    oracle_init_once();
    bn br, bs; bn_from_be(&br, r, 32); bn_from_be(&bs, s, 32);
    return verify_tuple(qx, qy, digest, dlen, &br, 0, &bs, 0);
}

/* ---- encoding/asn1 restatement (Go 1.14 asn1.go) -------------------------------------------- */
/* returns 0 ok, <0 error; *tag_ok set iff identifier octet == want */
static int der_tl(const uint8_t *b, size_t len, size_t *off, uint8_t want, size_t *out_len) {
    if (*off >= len) return -1;
    uint8_t id = b[(*off)++];
    if ((id & 0x1F) == 0x1F) return -2;        /* long-form tag: either malformed or a tag mismatch */
    if (*off >= len) return -3;                /* truncated tag or length */
    uint8_t l0 = b[(*off)++];
    size_t L;
    if (!(l0 & 0x80)) L = l0;
    else {
        int nb = l0 & 0x7F;
        if (nb == 0) return -4;                /* indefinite length */
        L = 0;
        for (int i = 0; i < nb; i++) {
            if (*off >= len) return -3;
            if (L >= ((size_t)1 << 23)) return -5; /* length too large */
            L = (L << 8) | b[(*off)++];
            if (L == 0) return -6;             /* superfluous leading zeros in length */
        }
        if (L < 0x80) return -7;               /* non-minimal length */
    }
    if (id != want) return -8;                 /* tags don't match (class, compound bit, number) */
    if (L > len - *off) return -9;             /* data truncated */
    *out_len = L;
    return 0;
}
/* one *big.Int field. sign: -1,0,+1. mag: low 256 bits; oversize: magnitude >= 2^256 */
static int der_bigint(const uint8_t *b, size_t len, size_t *off, int *sign, bn *mag, int *oversize) {
    if (*off == len) return -10;               /* sequence truncated */
    size_t L; int rc = der_tl(b, len, off, 0x02, &L);
    if (rc) return rc;
    const uint8_t *p = b + *off; *off += L;
    if (L == 0) return -11;                    /* empty integer */
    if (L > 1 && ((p[0] == 0x00 && !(p[1] & 0x80)) || (p[0] == 0xFF && (p[1] & 0x80)))) return -12;
    if (p[0] & 0x80) { *sign = -1; memset(mag, 0, sizeof *mag); *oversize = 0; return 0; }
    while (L > 0 && p[0] == 0) { p++; L--; }   /* at most one sign octet */
    if (L == 0) { *sign = 0; memset(mag, 0, sizeof *mag); *oversize = 0; return 0; }
    *sign = 1;
    if (L > 32) { *oversize = 1; bn_from_be(mag, p + (L - 32), 32); }
    else { *oversize = 0; bn_from_be(mag, p, L); }
    return 0;
}
/* bccsp/utils/ecdsa.go:43-67. Returns 0 ok; 1 asn1 failure; 2 R<=0; 3 S<=0.
 * r32/s32 big-endian (low 256 bits), flags bit0: r >= 2^256, bit1: s >= 2^256 */
int oracle_der_unmarshal(const uint8_t *sig, size_t len, uint8_t *r32, uint8_t *s32, int *flags) {
    size_t off = 0, L;
    if (len == 0) return 1;
    if (der_tl(sig, len, &off, 0x30, &L)) return 1;
    const uint8_t *in = sig + off; size_t ioff = 0;
    int rs, ss, ro, so; bn r, s;
    if (der_bigint(in, L, &ioff, &rs, &r, &ro)) return 1;
    if (der_bigint(in, L, &ioff, &ss, &s, &so)) return 1;
    if (rs != 1) return 2;
    if (ss != 1) return 3;
    bn_to_be(r32, &r); bn_to_be(s32, &s);
    *flags = (ro ? 1 : 0) | (so ? 2 : 0);
    return 0;
}

/* CSP.Verify for an ECDSA public key (bccsp/sw/impl.go:247-270 -> ecdsa.go:41-57).
 * Return: 0 (true,nil); 1 (false,nil); 2 (false,err) high-S; 3 range reject (false,nil);
 *         4 off-curve key; 10 empty signature; 11 empty digest; 12 DER failure; 13 R<=0; 14 S<=0 */
int oracle_bccsp_verify(const uint8_t *qx, const uint8_t *qy, const uint8_t *sig, size_t siglen,
                        const uint8_t *digest, size_t dlen) {
    oracle_init_once();
    if (siglen == 0) return 10;
    if (dlen == 0) return 11;
    uint8_t r32[32], s32[32]; int flags = 0;
    int rc = oracle_der_unmarshal(sig, siglen, r32, s32, &flags);
    if (rc) return 11 + rc;
    bn r, s; bn_from_be(&r, r32, 32); bn_from_be(&s, s32, 32);
    return verify_tuple(qx, qy, digest, dlen, &r, flags & 1, &s, (flags >> 1) & 1);
}

/* ---- SHA-256 (FIPS 180-4; Go crypto/sha256 via bccsp/sw/hash.go:29-33) ------------------------ */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t *blk) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)blk[4 * i] << 24 | (uint32_t)blk[4 * i + 1] << 16 | (uint32_t)blk[4 * i + 2] << 8 | blk[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void oracle_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= len; i += 64) sha256_block(h, msg + i);
    uint8_t tail[128]; size_t rem = len - i;
    memset(tail, 0, sizeof tail);
    if (rem) memcpy(tail, msg + i, rem);
    tail[rem] = 0x80;
    size_t tl = (rem + 9 <= 64) ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(h, tail);
    if (tl == 128) sha256_block(h, tail + 64);
    for (int k = 0; k < 8; k++) { out[4 * k] = h[k] >> 24; out[4 * k + 1] = h[k] >> 16; out[4 * k + 2] = h[k] >> 8; out[4 * k + 3] = h[k]; }
}
/* ragged batch: message i = arena[off[i] .. off[i+1]) (include/fabgpu.h layout) */
void oracle_sha256_batch(size_t n, const uint8_t *arena, const uint32_t *off, uint8_t *digests) {
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)n; i++) oracle_sha256(arena + off[i], off[i + 1] - off[i], digests + 32 * i);
}
/* identity.Verify over the flattened batch: hash then verify (msp/identities.go:169-196) */
void oracle_sha256_p256_verify_batch(size_t n, const uint8_t *arena, const uint32_t *off, const uint8_t *qx,
                                     const uint8_t *qy, const uint8_t *r, const uint8_t *s, uint8_t *status) {
    oracle_init_once();
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < (long)n; i++) {
        uint8_t d[32]; oracle_sha256(arena + off[i], off[i + 1] - off[i], d);
        bn br, bs; bn_from_be(&br, r + 32 * i, 32); bn_from_be(&bs, s + 32 * i, 32);
        status[i] = (uint8_t)verify_tuple(qx + 32 * i, qy + 32 * i, d, 32, &br, 0, &bs, 0);
    }
}

/* ---- test-vector construction helpers (sign side of bccsp/sw/ecdsa.go:27-39) -------------------- */
/* Q = d G */
void oracle_p256_pubkey(const uint8_t *d32, uint8_t *qx32, uint8_t *qy32) {
    oracle_init_once();
    bn d; bn_from_be(&d, d32, 32);
    jpt G, R; G.X = MGX; G.Y = MGY; G.Z = CP.one;
    jpt_mul(&R, &d, &G);
    bn x, y; memset(&x, 0, sizeof x); memset(&y, 0, sizeof y);
    jpt_to_affine(&x, &y, &R);
    bn_to_be(qx32, &x); bn_to_be(qy32, &y);
}
/* (r,s) = sign(d, e, k) with utils.ToLowS applied when low_s != 0. returns 0 ok, 1 if r or s is zero */
int oracle_p256_sign(const uint8_t *d32, const uint8_t *e32, const uint8_t *k32, int low_s, uint8_t *r32, uint8_t *s32) {
    oracle_init_once();
    bn d, e, k; bn_from_be(&d, d32, 32); bn_from_be(&e, e32, 32); bn_from_be(&k, k32, 32);
    jpt G, R; G.X = MGX; G.Y = MGY; G.Z = CP.one;
    jpt_mul(&R, &k, &G);
    bn x, y; if (!jpt_to_affine(&x, &y, &R)) return 1;
    bn r; mod_reduce_once(&r, &x, &CN);
    if (bn_is_zero(&r)) return 1;
    bn mk, mki, mr, md, me, t, s, ered;
    to_mont(&mk, &k, &CN); mont_inv(&mki, &mk, &CN);
    to_mont(&mr, &r, &CN); to_mont(&md, &d, &CN);
    mod_reduce_once(&ered, &e, &CN); to_mont(&me, &ered, &CN);
    mont_mul(&t, &mr, &md, &CN); mod_add(&t, &t, &me, &CN); mont_mul(&t, &t, &mki, &CN); from_mont(&s, &t, &CN);
    if (bn_is_zero(&s)) return 1;
    if (low_s && bn_cmp(&s, &P256_HALF_N) > 0) bn_sub(&s, &P256_N, &s);
    bn_to_be(r32, &r); bn_to_be(s32, &s);
    return 0;
}
/* keygen+sign n tuples from seeds: d[i], k[i] given (32-byte BE, already in [1,n-1]) */
void oracle_p256_make_batch(size_t n, const uint8_t *d, const uint8_t *k, const uint8_t *e,
                            uint8_t *qx, uint8_t *qy, uint8_t *r, uint8_t *s) {
    oracle_init_once();
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < (long)n; i++) {
        oracle_p256_pubkey(d + 32 * i, qx + 32 * i, qy + 32 * i);
        oracle_p256_sign(d + 32 * i, e + 32 * i, k + 32 * i, 1, r + 32 * i, s + 32 * i);
    }
}
