// secp256r1 point arithmetic on fe29 field elements + the ECDSA verification core of the verify kernels.
//
// What it replaces: Go 1.14 crypto/ecdsa.Verify + crypto/elliptic CombinedMult, reached from
// bccsp/sw/ecdsa.go:56.  Structure (DESIGN.md "Kernels"):
//   * one signature per lane; lane-uniform control flow (flags + selects);
//   * w = s^-1 mod n by safegcd (modinv30.h); u1 = e w, u2 = r w by Montgomery products mod n (fp256.h);
//   * u2*Q : fixed 4-bit windows over a 15-entry per-lane Jacobian table in private memory;
//   * u1*G : 64-window comb over a precomputed affine table staged in LDS, mixed additions only;
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

constexpr int G29_WINDOWS = 64;   // 4-bit comb windows over u1
constexpr int G29_ENTRIES = 15;   // digits 1..15
// LDS / global layout of the comb table: [window][coord(2)][limb(9)][digit-1 (16 slots, 15 used)] i32.  For a fixed
// (window, coord, limb) the 16 digits are 16 consecutive dwords: lanes with different digits hit different banks,
// equal digits broadcast (ds_read_b32 services 32 lanes per cycle over 32 banks).
constexpr int G29_TABLE_WORDS = G29_WINDOWS * 2 * 9 * 16;
FAB_HD int g29_index(int window, int coord, int limb, int digit_minus_1) {
    return ((window * 2 + coord) * 9 + limb) * 16 + digit_minus_1;
}
struct FlatGTab29 {
    const int32_t* w;
    FAB_HD void load(int window, uint32_t digit, fe& x, fe& y) const {
        const int32_t* base = w + g29_index(window, 0, 0, (int)digit - 1);
#pragma unroll
        for (int l = 0; l < 9; l++) {
            x.v[l] = base[l * 16];
            y.v[l] = base[(9 + l) * 16];
        }
    }
};

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

// Per-lane table j*Q (j = 0..15, entry 0 is filler) kept in a plain array: host builds and tests.
struct LocalQTab29 {
    jac29 t[16];
    FAB_HD void store(int j, const jac29& p) { t[j] = p; }
    FAB_HD void load(uint32_t d, jac29& p) const { p = t[d]; }
};

// The verification core.  GTab provides  void load(int window, uint32_t digit /*1..15*/, fe& x, fe& y);
// QTab provides store(j, point) / load(digit, point) over 16 per-lane entries.  Inputs are plain integers; e is hashToInt(digest).
template <class GTab, class QTab>
FAB_HD uint32_t p256_verify_core29(const u256& qx, const u256& qy, const u256& e, const u256& r, const u256& s,
                                   const GTab& gtab, QTab& qtab) {
    const u256 P = FAB_P256_P;
    const u256 N = FAB_P256_N;
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

    // --- scalars: w = s^-1, u1 = e w, u2 = r w  (mod n) ---
    u256 w, u1, u2, ered, t;
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

    // --- per-lane table j*Q, j = 1..15 ---
    qtab.store(0, Q);
    qtab.store(1, Q);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int j = 2; j < 16; j += 2) {
        jac29 d, a;
        fe h, rr;
        jac29 half;
        qtab.load((uint32_t)(j >> 1), half);
        pt_dbl29(d, half);
        qtab.store(j, d);
        pt_add_mixed29(a, d, Q.X, Q.Y, h, rr);
        qtab.store(j + 1, a);
    }

    // --- T = u2 * Q ---
    jac29 T = Q;
    bool t_inf = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 63; i >= 0; i--) {
        uint32_t d = nibble(u2, i);
        jac29 ent;
        qtab.load(d, ent);                             // issued ahead of the doublings: the gather latency hides behind them
        if (i != 63) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
            for (int k = 0; k < 4; k++) {
                jac29 dd;
                pt_dbl29(dd, T);
                T = dd;
            }
        }
        jac29 sum;
        fe h, rr;
        pt_add29(sum, T, ent, h, rr);
        bool take_ent = t_inf & (d != 0);
        bool take_sum = (!t_inf) & (d != 0);
        sel_jac29(T, take_sum, sum, T);
        sel_jac29(T, take_ent, ent, T);
        t_inf = t_inf & (d == 0);
    }

    // --- S = u1 * G (comb) ---
    jac29 S = Q;
    bool s_inf = true;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int i = 0; i < G29_WINDOWS; i++) {
        uint32_t d = nibble(u1, i);
        jac29 ent, sum;
        fe h, rr;
        gtab.load(i, d ? d : 1u, ent.X, ent.Y);
        ent.Z = ONE;
        pt_add_mixed29(sum, S, ent.X, ent.Y, h, rr);
        bool take_ent = s_inf & (d != 0);
        bool take_sum = (!s_inf) & (d != 0);
        sel_jac29(S, take_sum, sum, S);
        sel_jac29(S, take_ent, ent, S);
        s_inf = s_inf & (d == 0);
    }

    // --- R = S + T with the exceptional cases of the group law (Appendix A step 8) ---
    jac29 Rp, Rd;
    fe h, rr;
    pt_add29(Rp, S, T, h, rr);
    bool hz = fe_is_zero(h), rz = fe_is_zero(rr);
    pt_dbl29(Rd, T);
    bool r_inf = t_inf & s_inf;                       // cannot happen for u2 != 0; kept for completeness
    bool use_T = s_inf & !t_inf;
    bool use_S = t_inf & !s_inf;
    bool both = !s_inf & !t_inf;
    bool use_dbl = both & hz & rz;                    // S == T
    r_inf = r_inf | (both & hz & !rz);                // S == -T  -> point at infinity
    jac29 Rr = Rp;
    sel_jac29(Rr, use_dbl, Rd, Rr);
    sel_jac29(Rr, use_T, T, Rr);
    sel_jac29(Rr, use_S, S, Rr);
    // a Jacobian Z == 0 also encodes infinity (doubling a point of order 2 cannot happen on a prime-order curve; kept cheap)

    // --- x(R) mod n == r  without inverting Z ---
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
    ok = ok & !r_inf;

    uint32_t st = ok ? ST_VALID : ST_BAD_MATH;
    return early != ST_VALID ? early : st;
}

}  // namespace fab
