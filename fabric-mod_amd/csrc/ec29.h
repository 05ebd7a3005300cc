// Short-Weierstrass point arithmetic shared by the curves of this library, over any field element type F in the fe29
// representation (fe29.h: secp256r1's prime, type fe; bn29.h: FP256BN's prime, type fbn).  What is curve-specific - the
// doubling (it depends on the coefficient a) and the verification cores - lives in p256_verify29.h and bn_nym29.h; F's
// fe_mul / fe_sqr / fe_is_zero / fe_set_one and the curve's pt_dbl29 are found by overload on F.
//
// Limb-magnitude bookkeeping (fe29.h): every product is annotated  [L(a) x L(b)]  in units of 2^28; nothing here exceeds 4,
// inside the bound of either field (14 for fe, 12 for fbn).
#pragma once
#include "fe29.h"

namespace fab {

template <class F>
struct jac_t {
    F X, Y, Z;  // invariants between operations: L(X) = 1, L(Y) <= 3, L(Z) <= 2
};

// Comb tables: WINDOWS = 256 / BITS windows over a scalar k, T[w][d] = d * 2^(BITS w) * B for d = 1 .. 2^BITS - 1 as affine
// Montgomery fe29 points (80-byte entries x[9] y[9] pad[2], 16-byte aligned; entry 0 of each window is unused).  k * B is then
// WINDOWS mixed additions and no doubling.  Instances:
//   * the P-256 generator, BITS = 16: 16 windows, 80 MiB, built once per fabgpu_init, resident in the 256 MiB Infinity Cache;
//   * a registered P-256 public key, BITS = 8: 32 windows, 640 KiB per key (fabgpu_p256_key_register), L2-resident;
//   * the two bases HSk, HRand of a registered idemix issuer, BITS = 8 (fabgpu_idemix_issuer_register).
// Each lane gathers one entry (five 16-byte loads) per window, issued one window ahead so that the latency hides behind the
// previous mixed addition.  (History: a 4-bit generator comb staged in LDS needed 64 additions and, at 72 KiB -> 80 KiB
// allocated, pinned occupancy; the 8-bit comb from L2 needed 32; the additions, not the gathers, are what the kernel pays for.)
constexpr int COMB_ENTRY_WORDS = 20;
struct alignas(16) comb_quad {
    int32_t x, y, z, w;
};
template <int BITS>
struct CombTab {
    static constexpr int WINDOWS = 256 / BITS;
    static constexpr size_t TABLE_WORDS = (size_t)WINDOWS * (1u << BITS) * COMB_ENTRY_WORDS;
    static_assert(256 % BITS == 0 && BITS <= 16, "window width must divide 256");
    const int32_t* w;
    FAB_HD static size_t index(int window, uint32_t digit) { return ((size_t)window * (1u << BITS) + digit) * COMB_ENTRY_WORDS; }
    FAB_HD static uint32_t digit(const u256& k, int i) {          // bits [BITS i, BITS i + BITS) of k (never straddles a word)
        int bit = BITS * i;
        return (k.w[bit >> 5] >> (bit & 31)) & ((1u << BITS) - 1u);
    }
    template <class F>
    FAB_HD void load(int window, uint32_t digit, F& x, F& y) const {
        const comb_quad* e = reinterpret_cast<const comb_quad*>(w + index(window, digit));   // five global_load_dwordx4
        comb_quad a = e[0], b = e[1], c = e[2], d = e[3], f = e[4];
        x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
        x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
        x.v[8] = c.x; y.v[0] = c.y; y.v[1] = c.z; y.v[2] = c.w;
        y.v[3] = d.x; y.v[4] = d.y; y.v[5] = d.z; y.v[6] = d.w;
        y.v[7] = f.x; y.v[8] = f.y;
    }
};
typedef CombTab<16> GTab16;   // the P-256 generator
typedef CombTab<8> KeyTab8;   // a registered base point

template <class F>
FAB_HD void sel_jac29(jac_t<F>& r, bool c, const jac_t<F>& a, const jac_t<F>& b) {
    fe_sel(r.X, c, a.X, b.X);
    fe_sel(r.Y, c, a.Y, b.Y);
    fe_sel(r.Z, c, a.Z, b.Z);
}

// General Jacobian + Jacobian (12M + 4S).  Valid when neither input is infinity and P != +-Q; h and rr are returned so
// that the one caller that can meet the exceptional cases (the final addition) can test them.
// in: L(X1) <= 2, L(Y1) <= 3, L(Z1) <= 2;  L(X2) = 1, L(Y2) <= 3, L(Z2) <= 2.   out: L(X) = 1, L(Y) = 2, L(Z) = 1.
template <class F>
FAB_HD void pt_add29(jac_t<F>& r, const jac_t<F>& a, const jac_t<F>& b, F& h, F& rr) {
    F z1z1, z2z2, u1, u2, s1, s2, hh, hhh, v, t, x3;
    fe_sqr(z1z1, a.Z);             // [2x2]
    fe_sqr(z2z2, b.Z);             // [2x2]
    fe_mul(u1, a.X, z2z2);         // [2x1]
    fe_mul(u2, b.X, z1z1);         // [1x1]
    fe_mul(t, b.Z, z2z2);          // [2x1]
    fe_mul(s1, a.Y, t);            // [3x1]
    fe_mul(t, a.Z, z1z1);          // [2x1]
    fe_mul(s2, b.Y, t);            // [3x1]
    fe_sub(h, u2, u1);             // L2
    fe_sub(rr, s2, s1);            // L2
    fe_sqr(hh, h);                 // [2x2]
    fe_mul(hhh, hh, h);            // [1x2]
    fe_mul(v, u1, hh);             // [1x1]
    fe_sqr(t, rr);                 // [2x2]
    fe_sub(t, t, hhh);
    fe_sub(t, t, v);
    fe_sub(t, t, v);               // r^2 - h^3 - 2 v   L4
    fe_weak_norm(x3, t);           // L1
    fe_sub(t, v, x3);              // L2
    fe_mul(t, rr, t);              // [2x2]
    fe_mul(s1, s1, hhh);           // [1x1]
    fe_sub(r.Y, t, s1);            // L2
    fe_mul(t, a.Z, b.Z);           // [2x2]
    fe_mul(r.Z, t, h);             // [1x2]
    r.X = x3;
}

// Jacobian + affine (8M + 3S), same contract.   in: L(X1) = 1, L(Y1) <= 3, L(Z1) <= 2; bx, by normalised.
template <class F>
FAB_HD void pt_add_mixed29(jac_t<F>& r, const jac_t<F>& a, const F& bx, const F& by, F& h, F& rr) {
    F z1z1, u2, s2, hh, hhh, v, t, x3;
    fe_sqr(z1z1, a.Z);             // [2x2]
    fe_mul(u2, bx, z1z1);          // [1x1]
    fe_mul(t, a.Z, z1z1);          // [2x1]
    fe_mul(s2, by, t);             // [1x1]
    fe_sub(h, u2, a.X);            // L2
    fe_sub(t, s2, a.Y);            // L4
    fe_weak_norm(rr, t);           // L1
    fe_sqr(hh, h);                 // [2x2]
    fe_mul(hhh, hh, h);            // [1x2]
    fe_mul(v, a.X, hh);            // [1x1]
    fe_sqr(t, rr);                 // [1x1]
    fe_sub(t, t, hhh);
    fe_sub(t, t, v);
    fe_sub(t, t, v);               // L4
    fe_weak_norm(x3, t);           // L1
    fe_sub(t, v, x3);              // L2
    fe_mul(t, rr, t);              // [1x2]
    fe_mul(s2, a.Y, hhh);          // [3x1]
    fe_sub(r.Y, t, s2);            // L2
    fe_mul(r.Z, a.Z, h);           // [2x2]
    r.X = x3;
}

constexpr int Q5_WINDOWS = 52;   // signed 5-bit windows over a 256-bit scalar (52 * 5 = 260 >= 257 bits)

// Per-lane table j*Q, j = 1..16, kept in a plain array: host builds and tests.
template <class F>
struct LocalQTab {
    jac_t<F> t[16];
    FAB_HD void store(int j, const jac_t<F>& p) { t[j - 1] = p; }
    FAB_HD void load(uint32_t d, jac_t<F>& p) const { p = t[d - 1]; }
};

// S = k * B over a comb table of B (Tab::WINDOWS mixed additions; the next window's entry is gathered while this one is added).
// No addition can meet P == +-Q when k < the (prime) group order: the partial sum is (k mod 2^(BITS i)) B while the addend is
// d 2^(BITS i) B, and neither their difference nor their sum is a multiple of the order.  seed: any valid point.
template <class Tab, class F>
FAB_HD void comb_mult29(jac_t<F>& S, bool& s_inf, const u256& k, const Tab& tab, const jac_t<F>& seed) {
    F ONE;
    fe_set_one(ONE);
    S = seed;
    s_inf = true;
    uint32_t nd = Tab::digit(k, 0);
    F nx, ny;
    tab.load(0, nd ? nd : 1u, nx, ny);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 0; i < Tab::WINDOWS; i++) {
        uint32_t d = nd;
        jac_t<F> ent, sum;
        F h, rr;
        ent.X = nx;
        ent.Y = ny;
        ent.Z = ONE;
        int inext = i + 1 < Tab::WINDOWS ? i + 1 : i;
        nd = Tab::digit(k, inext);
        tab.load(inext, nd ? nd : 1u, nx, ny);
        pt_add_mixed29(sum, S, ent.X, ent.Y, h, rr);
        bool take_ent = s_inf & (d != 0);
        bool take_sum = (!s_inf) & (d != 0);
        sel_jac29(S, take_sum, sum, S);
        sel_jac29(S, take_ent, ent, S);
        s_inf = s_inf & (d == 0);
    }
}

// R = S + T with the exceptional cases of the group law: doubling when S == T, infinity when S == -T.
template <class F>
FAB_HD void final_add29(jac_t<F>& Rr, bool& r_inf, const jac_t<F>& S, bool s_inf, const jac_t<F>& T, bool t_inf) {
    jac_t<F> Rp, Rd;
    F h, rr;
    pt_add29(Rp, S, T, h, rr);
    bool hz = fe_is_zero(h), rz = fe_is_zero(rr);
    pt_dbl29(Rd, T);
    r_inf = t_inf & s_inf;
    bool use_T = s_inf & !t_inf;
    bool use_S = t_inf & !s_inf;
    bool both = !s_inf & !t_inf;
    bool use_dbl = both & hz & rz;                    // S == T
    r_inf = r_inf | (both & hz & !rz);                // S == -T  -> point at infinity
    Rr = Rp;
    sel_jac29(Rr, use_dbl, Rd, Rr);
    sel_jac29(Rr, use_T, T, Rr);
    sel_jac29(Rr, use_S, S, Rr);
}

// T = k * Q for an arbitrary (on-curve, affine, Montgomery-form) point Q of prime order and k < 2^(5 WINDOWS - 1), below that order:
// a per-lane table j*Q, j = 1..16 (8 doublings + 7 mixed additions; QTab provides store(j, point) / load(j, point)), then
// WINDOWS signed 5-bit (Booth) windows, digit_i = -16 k[5i+4] + 8 k[5i+3] + .. + k[5i] + k[5i-1] in [-16, 16]: (WINDOWS - 1) x 5
// doublings and at most WINDOWS additions of +-|digit| Q.  (No addition can meet P == +-Q: DESIGN.md.)  t_inf: k == 0.
template <class F, class QTab>
FAB_HD void build_lane_table29(const jac_t<F>& Q, QTab& qtab) {
    qtab.store(1, Q);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int j = 2; j <= 16; j += 2) {
        jac_t<F> d, a, half;
        F h, rr;
        qtab.load((uint32_t)(j >> 1), half);
        pt_dbl29(d, half);
        qtab.store(j, d);
        if (j < 16) {
            pt_add_mixed29(a, d, Q.X, Q.Y, h, rr);
            qtab.store(j + 1, a);
        }
    }
}
// signed 5-bit Booth digit i of a little-endian word array (NW words of scalar + one zero word): bits 5i-1 .. 5i+4, bit -1 = 0
template <int NW>
FAB_HD int32_t booth5_digit(const uint32_t (&kw)[NW], int i) {
    uint32_t six;
    if (i == 0) {
        six = (kw[0] << 1) & 63u;
    } else {
        int p = 5 * i - 1;
        uint64_t two = ((uint64_t)kw[(p >> 5) + 1] << 32) | kw[p >> 5];
        six = (uint32_t)(two >> (p & 31)) & 63u;
    }
    return (int32_t)((six >> 1) & 15u) + (int32_t)(six & 1u) - (int32_t)((six >> 5) << 4);
}
template <int WINDOWS, class F, class QTab>
FAB_HD void booth_mult29(jac_t<F>& T, bool& t_inf, const u256& k, const jac_t<F>& Q, QTab& qtab) {
    static_assert(WINDOWS >= 1 && WINDOWS <= Q5_WINDOWS, "5-bit windows over at most 260 bits");
    build_lane_table29(Q, qtab);

    uint32_t kw[9];
#pragma unroll
    for (int i = 0; i < 8; i++) kw[i] = k.w[i];
    kw[8] = 0;
    T = Q;
    t_inf = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = WINDOWS - 1; i >= 0; i--) {
        int32_t digit = booth5_digit(kw, i);
        bool neg = digit < 0;
        uint32_t mag = (uint32_t)(neg ? -digit : digit);
        jac_t<F> ent;
        qtab.load(mag ? mag : 1u, ent);                // issued ahead of the doublings: the gather latency hides behind them
        if (i != WINDOWS - 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int k5 = 0; k5 < 5; k5++) {
                jac_t<F> dd;
                pt_dbl29(dd, T);
                T = dd;
            }
        }
#pragma unroll
        for (int l = 0; l < 9; l++) ent.Y.v[l] = neg ? -ent.Y.v[l] : ent.Y.v[l];
        jac_t<F> sum;
        F h, rr;
        pt_add29(sum, T, ent, h, rr);
        bool take_ent = t_inf & (mag != 0);
        bool take_sum = (!t_inf) & (mag != 0);
        sel_jac29(T, take_sum, sum, T);
        sel_jac29(T, take_ent, ent, T);
        t_inf = t_inf & (mag == 0);
    }
}
// the full-width form: k below the group order, 52 windows
template <class F, class QTab>
FAB_HD void var_base_mult29(jac_t<F>& T, bool& t_inf, const u256& k, const jac_t<F>& Q, QTab& qtab) {
    booth_mult29<Q5_WINDOWS>(T, t_inf, k, Q, qtab);
}

// T = +-m1 * Q +- m2 * phi(Q) for an efficiently computable endomorphism phi(x, y) = (beta x, y) (GLV): both magnitudes below
// 2^134, so 27 signed 5-bit windows interleaved over ONE accumulator - 26 x 5 doublings and at most 54 additions, half the
// doublings of var_base_mult29 on the full scalar.  Entries of phi(Q)'s table are the entries of Q's with X scaled by beta
// (Jacobian: beta x = beta X / Z^2), one product per addition instead of a second table.
// Unlike the single-scalar loop, the interleaved one has no proof that an addition never meets P == +-Q (the accumulator is
// (a + b lambda) Q for partial scalars a, b): every addition that is taken tests h == 0, and `exc` reports the event - the
// caller must then discard T.  (For honest inputs the event needs a + b lambda = +-d (mod r) with small d: it does not occur.)
constexpr int GLV_WINDOWS = 27;
template <class F, class QTab>
FAB_HD void glv_mult29(jac_t<F>& T, bool& t_inf, bool& exc, const u256& m1, bool n1, const u256& m2, bool n2, const jac_t<F>& Q, const F& beta,
                       QTab& qtab) {
    build_lane_table29(Q, qtab);
    uint32_t k1[6], k2[6];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        k1[i] = m1.w[i];
        k2[i] = m2.w[i];
    }
    k1[5] = 0;
    k2[5] = 0;
    T = Q;
    t_inf = true;
    exc = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = GLV_WINDOWS - 1; i >= 0; i--) {
        int32_t d1 = booth5_digit(k1, i), d2 = booth5_digit(k2, i);
        uint32_t g1 = (uint32_t)(d1 < 0 ? -d1 : d1), g2 = (uint32_t)(d2 < 0 ? -d2 : d2);
        jac_t<F> e1, e2;
        qtab.load(g1 ? g1 : 1u, e1);                   // both gathers issued ahead of the doublings
        qtab.load(g2 ? g2 : 1u, e2);
        if (i != GLV_WINDOWS - 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int k5 = 0; k5 < 5; k5++) {
                jac_t<F> dd;
                pt_dbl29(dd, T);
                T = dd;
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
        for (int half = 0; half < 2; half++) {
            jac_t<F> ent = half ? e2 : e1;
            uint32_t mag = half ? g2 : g1;
            bool neg = half ? ((d2 < 0) != n2) : ((d1 < 0) != n1);
            if (half) {
                F bx;
                fe_mul(bx, ent.X, beta);                // [1x1]
                ent.X = bx;
            }
#pragma unroll
            for (int l = 0; l < 9; l++) ent.Y.v[l] = neg ? -ent.Y.v[l] : ent.Y.v[l];
            jac_t<F> sum;
            F h, rr;
            pt_add29(sum, T, ent, h, rr);
            bool take_ent = t_inf & (mag != 0);
            bool take_sum = (!t_inf) & (mag != 0);
            exc = exc | (take_sum & fe_is_zero(h));
            sel_jac29(T, take_sum, sum, T);
            sel_jac29(T, take_ent, ent, T);
            t_inf = t_inf & (mag == 0);
        }
    }
}

}  // namespace fab
