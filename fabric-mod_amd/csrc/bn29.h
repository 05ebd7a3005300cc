// FP256BN base-field arithmetic mod p (the 256-bit BN prime of idemix's pairing curve) in the fe29 representation:
// 9 signed limbs of 29 bits, lazy additions, Montgomery radix R = 2^261 - see fe29.h for why this shape suits the CDNA4
// integer VALU.  Unlike the P-256 prime this modulus has no structure, so the reduction is the generic one: per column a
// quotient digit q = column * (-p^-1) mod 2^29 and nine MACs q * p_j, with p held as BALANCED digits (|p_j| <= 2^28) so
// that a column stays inside the signed 64-bit accumulator:
//     9 * L(a) L(b) * 2^56  +  9 * 2^29 * 2^28  <  2^63    <=>    L(a) * L(b) <= 12          (P-256: 14)
// (L = limb magnitude in units of 2^28, as in fe29.h; FE29_CHECK builds assert the accumulator on every product.)
//
// Replaces, on the device, github.com/hyperledger/fabric-amcl amcl/FP256BN's FP arithmetic (third-party, not in the
// reference tree; go.mod:44) as reached from idemix/nymsignature.go:86-87.
#pragma once
#include "bn29_consts.h"
#include "fe29.h"
#if defined(__HIP_DEVICE_COMPILE__)
#include "bn29_gcn.h"
#endif

namespace fab {

// a distinct type (not a typedef of fe): handing a BN element to a P-256 routine must not compile
struct fbn {
    int32_t v[9];
};

// plain C multiply-accumulate: for this field hipcc's own v_mad_i64_i32 selection is kept (an asm statement per MAC, as fe29.h
// uses for its host-spec bodies, costs an s_nop of hazard padding each - 160 per product)
FAB_HD void bn_mac(fe29_acc_t& acc, int32_t a, int32_t b) { acc += (fe29_acc_t)((int64_t)a * b); }

#define BN29_COLUMN_TAIL(k)                                                                              \
    do {                                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 9; i_++) {                                               \
            int j_ = (k) - i_;                                                                           \
            if (i_ < (k) && j_ >= 0 && j_ < 9) bn_mac(acc, q[i_], PB[j_]);                         \
        }                                                                                                \
        if ((k) <= 8) {                                                                                  \
            q[(k)] = (int32_t)(((uint32_t)acc * BN29_N0) & (uint32_t)FE_M29);                            \
            bn_mac(acc, q[(k)], PB[0]);                                                            \
            FE29_ASSERT_ACC(acc);                                                                        \
            acc >>= 29;                                                                                  \
        } else {                                                                                         \
            FE29_ASSERT_ACC(acc);                                                                        \
            r.v[(k) - 9] = fe_sext29((uint32_t)acc);                                                     \
            acc = (acc + (1 << 28)) >> 29;                                                               \
            if ((k) == 16) {                                                                             \
                FE29_ASSERT_LIMB((int64_t)acc);                                                          \
                r.v[8] = (int32_t)acc;                                                                   \
            }                                                                                            \
        }                                                                                                \
    } while (0)

// r = a * b / 2^261 mod p, some representative in (T/R, T/R + p), T = a*b; digits 0..7 balanced, digit 8 small
// (on the device mul and sqr are real functions, by value in VGPRs: inlined, the ~100 products of a verification are 450 KB of
// code against a 64 KB instruction cache - the lesson of profiles/r01_bench_v0_inlined.txt)
FAB_FN fbn fbn_mul_fn(fbn a, fbn b) {
    const int32_t PB[9] = BN29_P_BAL;
    int32_t q[9];
    fbn r;
    fe29_acc_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j >= 0 && j < 9) bn_mac(acc, a.v[i], b.v[j]);
        }
        BN29_COLUMN_TAIL(k);
    }
    return r;
}
FAB_FN fbn fbn_sqr_fn(fbn a) {
    const int32_t PB[9] = BN29_P_BAL;
    int32_t q[9];
    int32_t a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.v[i] * 2;
    fbn r;
    fe29_acc_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j > i && j < 9) bn_mac(acc, a2[i], a.v[j]);
            if (j == i) bn_mac(acc, a.v[i], a.v[i]);
        }
        BN29_COLUMN_TAIL(k);
    }
    return r;
}
// On the device the product is ONE generated asm statement (bn29_gcn.h, gcn_dsl.py with a GenericField: 214 / 186 instructions,
// all 8-byte encodings, no hazard padding), inlined like P-256's: every loop body of the nym kernel (a doubling 11 KB, an
// addition 31 KB, a mixed addition 19 KB) still fits the 64 KB instruction cache.  The C bodies above are the host build and
// the specification; BN29_CALL_FIELD_FNS keeps them as real functions on the device too (the round-1 baseline: 246 / 215
// instructions plus ~30 of call overhead per product).
FAB_HD void fe_mul(fbn& r, const fbn& a, const fbn& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BN29_CALL_FIELD_FNS)
    fbn t;
    BN29_GCN_MUL(t, a, b);
    r = t;
#else
    r = fbn_mul_fn(a, b);
#endif
}
FAB_HD void fe_sqr(fbn& r, const fbn& a) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BN29_CALL_FIELD_FNS)
    fbn t, dbl;
    BN29_GCN_SQR(t, dbl, a);
    r = t;
#else
    r = fbn_sqr_fn(a);
#endif
}

FAB_HD void fe_set_one(fbn& r) {
    const fbn ONE = {BN29_R1};
    r = ONE;
}
FAB_HD void fe_to_mont(fbn& r, const u256& a) {
    const fbn RR = {BN29_RR};
    fbn t;
    fe_from_u256(t, a);       // unsigned digits: L = 2, times RR (L = 1)
    fe_mul(r, t, RR);
}
// canonical residue a / 2^261 mod p as unsigned digits of the representative in [0, p]  (|a| < 16 p; see fe29.h)
FAB_HD void fe_canon_div_r(uint32_t c[9], const fbn& a) {
    const fbn ONE = {{1, 0, 0, 0, 0, 0, 0, 0, 0}};
    fbn t;
    fe_mul(t, a, ONE);
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t x = t.v[i] + carry;
        c[i] = (uint32_t)x & (uint32_t)FE_M29;
        carry = x >> 29;
    }
    c[8] = (uint32_t)(t.v[8] + carry);
}
FAB_HD bool fe_is_zero(const fbn& a) {
    const uint32_t PU[9] = BN29_P_UNS;
    uint32_t c[9];
    fe_canon_div_r(c, a);
    uint32_t z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        z |= c[i];
        e |= c[i] ^ PU[i];
    }
    return (z == 0) | (e == 0);
}
// Montgomery form -> plain integer in [0, p)
FAB_HD void fe_from_mont(u256& r, const fbn& a) {
    const uint32_t PU[9] = BN29_P_UNS;
    uint32_t c[9];
    fe_canon_div_r(c, a);
    uint32_t e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) e |= c[i] ^ PU[i];
    if (e == 0) {
#pragma unroll
        for (int i = 0; i < 9; i++) c[i] = 0;
    }
    uint64_t bits = 0;
    int have = 0, w = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        bits |= (uint64_t)c[i] << have;
        have += (i < 8) ? 29 : 24;
        if (have >= 32 && w < 8) {
            r.w[w++] = (uint32_t)bits;
            bits >>= 32;
            have -= 32;
        }
    }
    if (w < 8) r.w[w] = (uint32_t)bits;
}

}  // namespace fab
