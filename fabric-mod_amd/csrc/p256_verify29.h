// secp256r1 point arithmetic on fe29 field elements + the ECDSA verification core of the verify kernels.
//
// What it replaces: Go 1.14 crypto/ecdsa.Verify + crypto/elliptic CombinedMult, reached from
// bccsp/sw/ecdsa.go:56.  Structure (DESIGN.md "Kernels"):
//   * one signature per lane; lane-uniform control flow (flags + selects);
//   * w = s^-1 mod n by safegcd (modinv30.h); u1 = e w, u2 = r w by Montgomery products mod n (fp256.h);
//   * u2*Q : 52 signed 5-bit (Booth) windows over a 16-entry per-lane Jacobian table in a global workspace;
//   * u1*G : 16-window 16-bit comb over a precomputed affine table (80 MiB, Infinity-Cache resident), mixed additions only;
//   * the two partial sums stay in SEPARATE accumulators, so for an on-curve Q no addition inside either loop can
//     meet P == +-Q (proof in DESIGN.md); only the final addition handles doubling / infinity explicitly;
//   * no field inversion: x(R) mod n == r is tested as X == r Z^2 or X == (r + n) Z^2.
//
// Limb-magnitude bookkeeping (fe29.h): every product below is annotated  [L(a) x L(b)]  in units of 2^28; the bound is 14.
#pragma once
#include "fe29.h"
#include "modinv30.h"
#include "p256_point.h"  // status codes, nibble(), range_status() shared with the host-side u256 code

namespace fab {

struct jac29 {
    fe X, Y, Z;  // invariants between operations: L(X) = 1, L(Y) <= 3, L(Z) = 1
};

// Comb tables: WINDOWS = 256 / BITS windows over a scalar k, T[w][d] = d * 2^(BITS w) * B for d = 1 .. 2^BITS - 1 as affine
// Montgomery fe29 points (80-byte entries x[9] y[9] pad[2], 16-byte aligned; entry 0 of each window is unused).  k * B is then
// WINDOWS mixed additions and no doubling.  Two instances:
//   * the generator, BITS = 16: 16 windows, 80 MiB, built once per fabgpu_init, resident in the 256 MiB Infinity Cache;
//   * a registered public key, BITS = 8: 32 windows, 640 KiB per key (fabgpu_p256_key_register), L2-resident.
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
    FAB_HD void load(int window, uint32_t digit, fe& x, fe& y) const {
        const comb_quad* e = reinterpret_cast<const comb_quad*>(w + index(window, digit));   // five global_load_dwordx4
        comb_quad a = e[0], b = e[1], c = e[2], d = e[3], f = e[4];
        x.v[0] = a.x; x.v[1] = a.y; x.v[2] = a.z; x.v[3] = a.w;
        x.v[4] = b.x; x.v[5] = b.y; x.v[6] = b.z; x.v[7] = b.w;
        x.v[8] = c.x; y.v[0] = c.y; y.v[1] = c.z; y.v[2] = c.w;
        y.v[3] = d.x; y.v[4] = d.y; y.v[5] = d.z; y.v[6] = d.w;
        y.v[7] = f.x; y.v[8] = f.y;
    }
};
typedef CombTab<16> GTab16;   // the generator
typedef CombTab<8> KeyTab8;   // a registered public key

FAB_HD void sel_jac29(jac29& r, bool c, const jac29& a, const jac29& b) {
    fe_sel(r.X, c, a.X, b.X);
    fe_sel(r.Y, c, a.Y, b.Y);
    fe_sel(r.Z, c, a.Z, b.Z);
}

// y^2 == x^3 - 3x + b, Montgomery form, x and y normalised
FAB_HD bool on_curve29(const fe& x, const fe& y) {
    const fe B = {FE29_B_MONT};
    fe l, x2, x3, t, d;
    fe_sqr(l, y);          // [1x1]
    fe_sqr(x2, x);         // [1x1]
    fe_mul(x3, x2, x);     // [1x1]
    fe_add(t, x, x);
    fe_add(t, t, x);       // 3x        L3
    fe_sub(d, l, x3);      //           L2
    fe_add(d, d, t);       //           L5
    fe_sub(d, d, B);       //           L6, |value| < 12 p
    return fe_is_zero(d);  // [6x1]
}

// Doubling, a = -3 (dbl-2001-b with Z3 = 2YZ so that no operand of a square exceeds the limb bound): 4M + 4S.
// in: L(X) <= 2, L(Y) <= 3, L(Z) <= 3.   out: L(X) = 1, L(Y) = 3, L(Z) = 1.
FAB_HD void pt_dbl29(jac29& r, const jac29& a) {
    fe delta, gamma, t1, t2, m, alpha, beta4, x3, y2, g2, gg, yy;
    fe_sqr(delta, a.Z);            // [3x3]
    fe_sqr(gamma, a.Y);            // [3x3]
    fe_sub(t1, a.X, delta);        // L3
    fe_add(t2, a.X, delta);        // L3
    fe_mul(m, t1, t2);             // [3x3]
    fe_add(alpha, m, m);
    fe_add(alpha, alpha, m);       // 3 (X - delta)(X + delta)   L3
    fe_add(t1, a.X, a.X);
    fe_add(t1, t1, t1);            // 4X   L8
    fe_mul(beta4, t1, gamma);      // [8x1]  4 X Y^2
    fe_sqr(t2, alpha);             // [3x3]
    fe_sub(t2, t2, beta4);
    fe_sub(t2, t2, beta4);         // alpha^2 - 8 beta   L3
    fe_weak_norm(x3, t2);          // L1
    fe_add(y2, a.Y, a.Y);          // L6
    fe_mul(r.Z, y2, a.Z);          // [6x1]  (Lz = 1 whenever Ly = 3)
    fe_add(g2, gamma, gamma);      // L2
    fe_sqr(gg, g2);                // [2x2]  4 gamma^2
    fe_sub(t1, beta4, x3);         // L2
    fe_mul(yy, alpha, t1);         // [3x2]
    fe_sub(yy, yy, gg);
    fe_sub(r.Y, yy, gg);           // alpha (4 beta - X3) - 8 gamma^2   L3
    r.X = x3;
}

// General Jacobian + Jacobian (12M + 4S).  Valid when neither input is infinity and P != +-Q; h and rr are returned so
// that the one caller that can meet the exceptional cases (the final addition) can test them.
// in: L(X1) <= 2, L(Y1) <= 3, L(Z1) = 1;  L(X2) = 1, L(Y2) <= 3, L(Z2) = 1.   out: L(X) = 1, L(Y) = 2, L(Z) = 1.
FAB_HD void pt_add29(jac29& r, const jac29& a, const jac29& b, fe& h, fe& rr) {
    fe z1z1, z2z2, u1, u2, s1, s2, hh, hhh, v, t, x3;
    fe_sqr(z1z1, a.Z);             // [1x1]
    fe_sqr(z2z2, b.Z);             // [1x1]
    fe_mul(u1, a.X, z2z2);         // [2x1]
    fe_mul(u2, b.X, z1z1);         // [1x1]
    fe_mul(t, b.Z, z2z2);          // [1x1]
    fe_mul(s1, a.Y, t);            // [3x1]
    fe_mul(t, a.Z, z1z1);          // [1x1]
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
    fe_mul(t, a.Z, b.Z);           // [1x1]
    fe_mul(r.Z, t, h);             // [1x2]
    r.X = x3;
}

// Jacobian + affine (8M + 3S), same contract.   in: L(X1) = 1, L(Y1) <= 3, L(Z1) = 1; bx, by normalised.
FAB_HD void pt_add_mixed29(jac29& r, const jac29& a, const fe& bx, const fe& by, fe& h, fe& rr) {
    fe z1z1, u2, s2, hh, hhh, v, t, x3;
    fe_sqr(z1z1, a.Z);             // [1x1]
    fe_mul(u2, bx, z1z1);          // [1x1]
    fe_mul(t, a.Z, z1z1);          // [1x1]
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
    fe_mul(r.Z, a.Z, h);           // [1x2]
    r.X = x3;
}

constexpr int Q5_WINDOWS = 52;   // signed 5-bit windows over u2 (52 * 5 = 260 >= 257 bits)

// Per-lane table j*Q, j = 1..16, kept in a plain array: host builds and tests.
struct LocalQTab29 {
    jac29 t[16];
    FAB_HD void store(int j, const jac29& p) { t[j - 1] = p; }
    FAB_HD void load(uint32_t d, jac29& p) const { p = t[d - 1]; }
};

// S = k * B over a comb table of B (Tab::WINDOWS mixed additions; the next window's entry is gathered while this one is added).
// No addition can meet P == +-Q: the partial sum is < 2^(BITS i) B while the addend is d 2^(BITS i) B.  seed: any valid point.
template <class Tab>
FAB_HD void comb_mult29(jac29& S, bool& s_inf, const u256& k, const Tab& tab, const jac29& seed) {
    const fe ONE = {FE29_R1};
    S = seed;
    s_inf = true;
    uint32_t nd = Tab::digit(k, 0);
    fe nx, ny;
    tab.load(0, nd ? nd : 1u, nx, ny);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 0; i < Tab::WINDOWS; i++) {
        uint32_t d = nd;
        jac29 ent, sum;
        fe h, rr;
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

// R = S + T with the exceptional cases of the group law (Appendix A step 8): doubling when S == T, infinity when S == -T.
FAB_HD void final_add29(jac29& Rr, bool& r_inf, const jac29& S, bool s_inf, const jac29& T, bool t_inf) {
    jac29 Rp, Rd;
    fe h, rr;
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

// x(R) mod n == r without inverting Z:  X == r Z^2  or  (r < p - n and X == (r + n) Z^2)
FAB_HD bool x_equals_r29(const jac29& Rr, bool r_inf, const u256& r) {
    const u256 N = FAB_P256_N;
    const u256 PMN = FAB_P256_P_MINUS_N;
    fe zz, rm, rhs, d;
    u256 r2;
    fe_sqr(zz, Rr.Z);                                  // [1x1]
    fe_to_mont(rm, r);
    fe_mul(rhs, rm, zz);
    fe_sub(d, Rr.X, rhs);
    bool ok = fe_is_zero(d);
    add256(r2, r, N);                                  // only meaningful when r < p - n (no wrap)
    fe_to_mont(rm, r2);
    fe_mul(rhs, rm, zz);
    fe_sub(d, Rr.X, rhs);
    ok = ok | (lt256(r, PMN) & fe_is_zero(d));
    return ok & !r_inf;
}

// w = s^-1, u1 = e w, u2 = r w  (mod n)
FAB_HD void ecdsa_scalars29(u256& u1, u256& u2, const u256& e, const u256& r, const u256& s) {
    const u256 N = FAB_P256_N;
    u256 w, ered, t;
    {
        const modinv_info NI = MODINV_N_INFO;
        modinv(w, s, NI);                       // s >= n only on lanes already rejected by the low-S gate
    }
    uint32_t br = sub256(t, e, N);              // e < 2^256 < 2n: one conditional subtraction
    sel256(ered, br == 0, t, e);
    fn_to_mont(t, ered);
    fn_mul(u1, t, w);                           // (e R)(w) / R
    fn_to_mont(t, r);
    fn_mul(u2, t, w);
}

// R = u1*G + u2*Q for an on-curve affine Q (Montgomery form) and u1, u2 < n, u2 != 0: the CombinedMult of the reference's
// crypto/elliptic.  r_inf reports the point at infinity (then Rr is meaningless).
template <class GTab, class QTab>
FAB_HD void p256_combined_mult29(jac29& Rr, bool& r_inf, const u256& u1, const u256& u2, const jac29& Q, const GTab& gtab, QTab& qtab) {

    // --- per-lane table j*Q, j = 1..16 (8 doublings + 7 mixed additions) ---
    qtab.store(1, Q);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int j = 2; j <= 16; j += 2) {
        jac29 d, a, half;
        fe h, rr;
        qtab.load((uint32_t)(j >> 1), half);
        pt_dbl29(d, half);
        qtab.store(j, d);
        if (j < 16) {
            pt_add_mixed29(a, d, Q.X, Q.Y, h, rr);
            qtab.store(j + 1, a);
        }
    }

    // --- T = u2 * Q : 52 signed 5-bit (Booth) windows, digit_i = -16 k[5i+4] + 8 k[5i+3] + .. + k[5i] + k[5i-1] in [-16, 16];
    //     51 x 5 doublings and at most 52 additions of +-|digit| Q.  (No addition can meet P == +-Q: DESIGN.md.) ---
    uint32_t kw[9];
#pragma unroll
    for (int i = 0; i < 8; i++) kw[i] = u2.w[i];
    kw[8] = 0;
    jac29 T = Q;
    bool t_inf = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = Q5_WINDOWS - 1; i >= 0; i--) {
        uint32_t six;                                  // bits 5i-1 .. 5i+4 of u2 (bit -1 = 0)
        if (i == 0) {
            six = (kw[0] << 1) & 63u;
        } else {
            int p = 5 * i - 1;
            uint64_t two = ((uint64_t)kw[(p >> 5) + 1] << 32) | kw[p >> 5];
            six = (uint32_t)(two >> (p & 31)) & 63u;
        }
        int32_t digit = (int32_t)((six >> 1) & 15u) + (int32_t)(six & 1u) - (int32_t)((six >> 5) << 4);
        bool neg = digit < 0;
        uint32_t mag = (uint32_t)(neg ? -digit : digit);
        jac29 ent;
        qtab.load(mag ? mag : 1u, ent);                // issued ahead of the doublings: the gather latency hides behind them
        if (i != Q5_WINDOWS - 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int k = 0; k < 5; k++) {
                jac29 dd;
                pt_dbl29(dd, T);
                T = dd;
            }
        }
#pragma unroll
        for (int l = 0; l < 9; l++) ent.Y.v[l] = neg ? -ent.Y.v[l] : ent.Y.v[l];
        jac29 sum;
        fe h, rr;
        pt_add29(sum, T, ent, h, rr);
        bool take_ent = t_inf & (mag != 0);
        bool take_sum = (!t_inf) & (mag != 0);
        sel_jac29(T, take_sum, sum, T);
        sel_jac29(T, take_ent, ent, T);
        t_inf = t_inf & (mag == 0);
    }

    // --- S = u1 * G (8-bit comb), then R = S + T ---
    jac29 S;
    bool s_inf;
    comb_mult29(S, s_inf, u1, gtab, Q);
    final_add29(Rr, r_inf, S, s_inf, T, t_inf);
}

// R = u1*G + u2*Q with BOTH points on precomputed comb tables (a registered public key): 16 + 32 mixed additions, no
// doublings, no per-lane table.  seed: any valid point (used as filler while an accumulator is still at infinity).
template <class GTab, class KTab>
FAB_HD void p256_combined_mult_keyed29(jac29& Rr, bool& r_inf, const u256& u1, const u256& u2, const GTab& gtab, const KTab& ktab) {
    const fe ONE = {FE29_R1};
    jac29 seed, S, T;
    bool s_inf, t_inf;
    gtab.load(0, 1u, seed.X, seed.Y);
    seed.Z = ONE;
    comb_mult29(T, t_inf, u2, ktab, seed);
    comb_mult29(S, s_inf, u1, gtab, seed);
    final_add29(Rr, r_inf, S, s_inf, T, t_inf);
}

// The verification core.  GTab provides  void load(int window, uint32_t digit /*1..255*/, fe& x, fe& y);
// QTab provides store(j, point) / load(j, point) for the 16 per-lane entries j = 1..16.  Inputs are plain integers; e is hashToInt(digest).
template <class GTab, class QTab>
FAB_HD uint32_t p256_verify_core29(const u256& qx, const u256& qy, const u256& e, const u256& r, const u256& s,
                                   const GTab& gtab, QTab& qtab) {
    const u256 P = FAB_P256_P;
    const fe ONE = {FE29_R1};
    uint32_t early = range_status(r, s);

    // --- public key to Montgomery form + curve membership (reference: enforced at key import) ---
    bool q_in_field = lt256(qx, P) & lt256(qy, P);
    jac29 Q;
    fe_to_mont(Q.X, qx);
    fe_to_mont(Q.Y, qy);
    Q.Z = ONE;
    bool q_ok = q_in_field & on_curve29(Q.X, Q.Y);
    if (early == ST_VALID && !q_ok) early = ST_OFF_CURVE;

    u256 u1, u2;
    ecdsa_scalars29(u1, u2, e, r, s);

    jac29 Rr;
    bool r_inf;
    p256_combined_mult29(Rr, r_inf, u1, u2, Q, gtab, qtab);
    bool ok = x_equals_r29(Rr, r_inf, r);
    uint32_t st = ok ? ST_VALID : ST_BAD_MATH;
    return early != ST_VALID ? early : st;
}

// The same verification for a REGISTERED key (curve membership was checked at registration): ktab is the key's comb table.
template <class GTab, class KTab>
FAB_HD uint32_t p256_verify_keyed_core29(const u256& e, const u256& r, const u256& s, const GTab& gtab, const KTab& ktab) {
    uint32_t early = range_status(r, s);
    u256 u1, u2;
    ecdsa_scalars29(u1, u2, e, r, s);
    jac29 Rr;
    bool r_inf;
    p256_combined_mult_keyed29(Rr, r_inf, u1, u2, gtab, ktab);
    bool ok = x_equals_r29(Rr, r_inf, r);
    uint32_t st = ok ? ST_VALID : ST_BAD_MATH;
    return early != ST_VALID ? early : st;
}

}  // namespace fab
