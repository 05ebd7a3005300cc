// Modular inversion by Bernstein-Yang "safegcd" division steps (https://gcd.cr.yp.to/papers.html#safegcd), in the
// fixed-iteration form with 30 division steps per batch on signed 30-bit limbs.  Lane-uniform by construction: 20
// batches x 30 steps = 600 >= 590 steps, the proven bound for any odd modulus below 2^256 with the "delta = 1/2"
// start used here, so every lane runs the same instruction stream whatever its input.
//
// Cost: ~14 k issue slots per inversion against ~195 k for the Fermat ladder a^(n-2) it replaces in the verify kernel
// (329 generic Montgomery products mod n): w = s^-1 mod n is step 7 of SURVEY.md Appendix A
// (crypto/ecdsa.Verify reached from bccsp/sw/ecdsa.go:56).
//
// Value = sum v[i] * 2^(30 i), v[0..7] in [0, 2^30), v[8] signed.
#pragma once
#include <stdint.h>

#include "fp256.h"

namespace fab {

struct s30 {
    int32_t v[9];
};
struct modinv_info {
    s30 modulus;          // the odd modulus M
    uint32_t inv30;       // M^-1 mod 2^30
};
struct trans2x2 {
    int32_t u, v, q, r;
};

constexpr int32_t MI_M30 = (1 << 30) - 1;

#define MODINV_N_INFO {{{1013130577, 250030859, 830072911, 968797033, 1073741756, 1073741823, 4095, 1073725440, 65535}}, 301941681u}
#define MODINV_P_INFO {{{1073741823, 1073741823, 1073741823, 63, 0, 0, 4096, 1073725440, 65535}}, 1073741823u}

FAB_HD void s30_from_u256(s30& r, const u256& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int bit = 30 * i;
        int w = bit >> 5, sh = bit & 31;
        uint32_t lo = a.w[w] >> sh;
        if (sh > 2 && w + 1 < 8) lo |= a.w[w + 1] << (32 - sh);
        r.v[i] = (int32_t)(lo & (uint32_t)MI_M30);
    }
}
// value must be in [0, 2^256)
FAB_HD void s30_to_u256(u256& r, const s30& a) {
    uint64_t bits = 0;
    int have = 0, w = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        bits |= (uint64_t)(uint32_t)a.v[i] << have;
        have += (i < 8) ? 30 : 16;
        if (have >= 32 && w < 8) {
            r.w[w++] = (uint32_t)bits;
            bits >>= 32;
            have -= 32;
        }
    }
}

// 30 division steps on the low words of f and g.  zeta = -(delta + 1/2).  Returns the new zeta and the transition matrix
// t (scaled by 2^30):  2^30 * [f', g'] = t * [f, g].
FAB_HD int32_t modinv_divsteps30(int32_t zeta, uint32_t f0, uint32_t g0, trans2x2& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    uint32_t f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        uint32_t neg = (uint32_t)(zeta >> 31);        // all-ones when delta > 0
        uint32_t odd = 0u - (g & 1u);                 // all-ones when g is odd
        // (x, y, z) = +-(f, u, v), the sign chosen so that g + x is the division step's numerator
        uint32_t x = (f ^ neg) - neg, y = (u ^ neg) - neg, z = (v ^ neg) - neg;
        g += x & odd;
        q += y & odd;
        r += z & odd;
        uint32_t swap = neg & odd;                    // delta > 0 and g odd: f takes the old g
        zeta = (int32_t)(((uint32_t)zeta ^ swap) - 1u);
        f += g & swap;
        u += q & swap;
        v += r & swap;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}

// The same 30 division steps with ONE COLUMN of the transition matrix: the recurrences of (u, q) and of (v, r) are the same code on
// different initial values - (1, 0) and (0, 1) - so the two lanes of a signature's pair compute a column each (p256_pair29.h pair_modinv)
// and exchange them: six instructions less per division step than both lanes computing all four entries.  a, b: in = the column's
// initial values, out = its entries (u, q) or (v, r).
FAB_HD int32_t modinv_divsteps30_column(int32_t zeta, uint32_t f0, uint32_t g0, int32_t& a, int32_t& b) {
    uint32_t u = (uint32_t)a, q = (uint32_t)b;
    uint32_t f = f0, g = g0;
#pragma unroll
    for (int i = 0; i < 30; i++) {
        uint32_t neg = (uint32_t)(zeta >> 31);
        uint32_t odd = 0u - (g & 1u);
        uint32_t x = (f ^ neg) - neg, y = (u ^ neg) - neg;
        g += x & odd;
        q += y & odd;
        uint32_t swap = neg & odd;
        zeta = (int32_t)(((uint32_t)zeta ^ swap) - 1u);
        f += g & swap;
        u += q & swap;
        g >>= 1;
        u <<= 1;
    }
    a = (int32_t)u;
    b = (int32_t)q;
    return zeta;
}

// (f, g) <- t * (f, g) / 2^30  (exact)
FAB_HD void modinv_update_fg(s30& f, s30& g, const trans2x2& t) {
    int64_t cf = (int64_t)t.u * f.v[0] + (int64_t)t.v * g.v[0];
    int64_t cg = (int64_t)t.q * f.v[0] + (int64_t)t.r * g.v[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += (int64_t)t.u * f.v[i] + (int64_t)t.v * g.v[i];
        cg += (int64_t)t.q * f.v[i] + (int64_t)t.r * g.v[i];
        f.v[i - 1] = (int32_t)((uint32_t)cf & (uint32_t)MI_M30);
        g.v[i - 1] = (int32_t)((uint32_t)cg & (uint32_t)MI_M30);
        cf >>= 30;
        cg >>= 30;
    }
    f.v[8] = (int32_t)cf;
    g.v[8] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^30 mod M, keeping both in (-2M, M)
FAB_HD void modinv_update_de(s30& d, s30& e, const trans2x2& t, const modinv_info& mi) {
    int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    // start from the multiple of M that keeps the result in range ...
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d.v[0] + (int64_t)t.v * e.v[0];
    int64_t ce = (int64_t)t.q * d.v[0] + (int64_t)t.r * e.v[0];
    // ... and correct it so that the low 30 bits of (cd + M * md) vanish
    md -= (int32_t)((mi.inv30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)MI_M30);
    me -= (int32_t)((mi.inv30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)MI_M30);
    cd += (int64_t)mi.modulus.v[0] * md;
    ce += (int64_t)mi.modulus.v[0] * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd += (int64_t)t.u * d.v[i] + (int64_t)t.v * e.v[i] + (int64_t)mi.modulus.v[i] * md;
        ce += (int64_t)t.q * d.v[i] + (int64_t)t.r * e.v[i] + (int64_t)mi.modulus.v[i] * me;
        d.v[i - 1] = (int32_t)((uint32_t)cd & (uint32_t)MI_M30);
        e.v[i - 1] = (int32_t)((uint32_t)ce & (uint32_t)MI_M30);
        cd >>= 30;
        ce >>= 30;
    }
    d.v[8] = (int32_t)cd;
    e.v[8] = (int32_t)ce;
}

// r in (-2M, M), limbs 0..7 in [0, 2^30): negate if neg_mask, then bring into [0, M).
FAB_HD void modinv_normalize(s30& r, int32_t neg_mask, const modinv_info& mi) {
    // 1. add M if negative  -> (-M, M)
    // 2. conditional negation -> (-M, M)
    // 3. add M if negative  -> [0, M)
    // each step followed by a carry propagation to limbs in [0, 2^30) with a signed top limb
    int32_t add = r.v[8] >> 31;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t x = r.v[i] + (mi.modulus.v[i] & add);
        x = (x ^ neg_mask) - neg_mask;
        x += c;
        if (i < 8) {
            r.v[i] = x & MI_M30;
            c = x >> 30;
        } else {
            r.v[i] = x;
        }
    }
    add = r.v[8] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t x = r.v[i] + (mi.modulus.v[i] & add) + c;
        if (i < 8) {
            r.v[i] = x & MI_M30;
            c = x >> 30;
        } else {
            r.v[i] = x;
        }
    }
}

// r = x^-1 mod M for 0 < x < M, M odd (x = 0 gives 0).  Plain integers in and out.
FAB_HD void modinv(u256& out, const u256& x, const modinv_info& mi) {
    s30 d, e, f, g;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d.v[i] = 0;
        e.v[i] = 0;
    }
    e.v[0] = 1;
    f = mi.modulus;
    s30_from_u256(g, x);
    int32_t zeta = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int it = 0; it < 20; it++) {
        // 600 division steps is the PROVED bound for 256 bits; uniformly random inputs need 502-531 (17 or 18 batches), and once g == 0 further
        // batches only move d by multiples of M.  The inputs are public (r, s of a signature), so nothing is given away by stopping when
        // every lane of the wavefront is done - a crafted input simply runs the full count.
        if (it >= 17) {
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) nz |= (uint32_t)g.v[i];
#if defined(__HIP_DEVICE_COMPILE__)
            if (__builtin_amdgcn_ballot_w64(nz != 0) == 0) break;
#else
            if (nz == 0) break;
#endif
        }
        trans2x2 t;
        zeta = modinv_divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        modinv_update_de(d, e, t, mi);
        modinv_update_fg(f, g, t);
    }
    // g == 0, f == +-1: the inverse is d * sign(f)
    modinv_normalize(d, f.v[8] >> 31, mi);
    s30_to_u256(out, d);
}

}  // namespace fab
