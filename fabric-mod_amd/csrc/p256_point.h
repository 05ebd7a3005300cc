// secp256r1 point arithmetic + the ECDSA verification core, shared by the HIP kernels and host tools.
//
// What it replaces: Go 1.14 crypto/ecdsa.Verify + crypto/elliptic CombinedMult, reached from
// bccsp/sw/ecdsa.go:56.  Design notes (DESIGN.md "Kernel"):
//   * one signature per lane; all control flow is lane-uniform (flags + selects), no divergence;
//   * u2*Q : fixed 4-bit windows, 16-entry per-lane Jacobian table kept in private (scratch) memory
//            whose layout is lane-interleaved, i.e. coalesced;
//   * u1*G : 64-window comb over a precomputed affine table (15 entries per window) that the kernel
//            stages in LDS; adds only, no doublings;
//   * the two partial sums are kept in SEPARATE accumulators so that, for an on-curve Q, no
//            exceptional case (P == +-Q) can occur inside either loop (proof in DESIGN.md); the single
//            final addition handles doubling / infinity explicitly;
//   * no field inversion: x(R) mod n == r is tested as X == r*Z^2 or X == (r+n)*Z^2 (r + n < p).
#pragma once
#include "fp256.h"

namespace fab {

// status codes of include/fabgpu.h
enum : uint32_t { ST_VALID = 0, ST_BAD_MATH = 1, ST_HIGH_S = 2, ST_RANGE = 3, ST_OFF_CURVE = 4 };

struct jac {
    u256 X, Y, Z;
};
struct aff {
    u256 x, y;
};

constexpr int G_WINDOWS = 64;          // 4-bit comb windows over u1
constexpr int G_ENTRIES = 15;          // digits 1..15
// LDS / global layout of the comb table: [window][coord(2)][limb(8)][digit-1 (16 slots, 15 used)] u32.
// For a fixed (window, coord, limb) the 16 digits are 16 consecutive dwords -> lanes with different
// digits hit different LDS banks, equal digits broadcast.
constexpr int G_TABLE_WORDS = G_WINDOWS * 2 * 8 * 16;
FAB_HD int g_index(int window, int coord, int limb, int digit_minus_1) { return ((window * 2 + coord) * 8 + limb) * 16 + digit_minus_1; }

// y^2 == x^3 - 3x + b, x,y already < p and in Montgomery form
FAB_HD bool on_curve_mont(const u256& x, const u256& y) {
    const u256 B = FAB_P256_B_MONT;
    u256 l, r, t;
    fp_sqr(l, y);
    fp_sqr(r, x);
    fp_mul(r, r, x);
    fp_add(t, x, x);
    fp_add(t, t, x);
    fp_sub(r, r, t);
    fp_add(r, r, B);
    return eq256(l, r);
}

// dbl-2001-b, a = -3: 3M + 5S
FAB_HD void pt_dbl(jac& r, const jac& a) {
    u256 delta, gamma, beta, alpha, t1, t2;
    fp_sqr(delta, a.Z);
    fp_sqr(gamma, a.Y);
    fp_mul(beta, a.X, gamma);
    fp_sub(t1, a.X, delta);
    fp_add(t2, a.X, delta);
    fp_mul(alpha, t1, t2);
    fp_add(t1, alpha, alpha);
    fp_add(alpha, t1, alpha);              // 3 (X-delta)(X+delta)
    fp_add(t1, a.Y, a.Z);
    fp_sqr(t1, t1);
    fp_sub(t1, t1, gamma);
    fp_sub(r.Z, t1, delta);                // Z3 = (Y+Z)^2 - gamma - delta
    fp_add(t1, beta, beta);
    fp_add(t1, t1, t1);                    // 4 beta
    fp_sqr(t2, alpha);
    fp_sub(t2, t2, t1);
    fp_sub(r.X, t2, t1);                   // X3 = alpha^2 - 8 beta
    fp_sub(t1, t1, r.X);
    fp_mul(t1, alpha, t1);                 // alpha (4 beta - X3)
    fp_sqr(t2, gamma);
    fp_add(t2, t2, t2);
    fp_add(t2, t2, t2);
    fp_add(t2, t2, t2);                    // 8 gamma^2
    fp_sub(r.Y, t1, t2);
}

// General Jacobian + Jacobian (12M + 4S).  Valid when neither input is infinity and P != +-Q;
// h_zero / r_zero report the exceptional cases to the caller.
FAB_HD void pt_add(jac& r, const jac& a, const jac& b, bool& h_zero, bool& r_zero) {
    u256 z1z1, z2z2, u1, u2, s1, s2, h, rr, hh, hhh, v, t;
    fp_sqr(z1z1, a.Z);
    fp_sqr(z2z2, b.Z);
    fp_mul(u1, a.X, z2z2);
    fp_mul(u2, b.X, z1z1);
    fp_mul(t, b.Z, z2z2);
    fp_mul(s1, a.Y, t);
    fp_mul(t, a.Z, z1z1);
    fp_mul(s2, b.Y, t);
    fp_sub(h, u2, u1);
    fp_sub(rr, s2, s1);
    h_zero = is_zero(h);
    r_zero = is_zero(rr);
    fp_sqr(hh, h);
    fp_mul(hhh, hh, h);
    fp_mul(v, u1, hh);
    fp_sqr(t, rr);
    fp_sub(t, t, hhh);
    fp_sub(t, t, v);
    fp_sub(r.X, t, v);                     // X3 = r^2 - h^3 - 2 v
    fp_sub(t, v, r.X);
    fp_mul(t, rr, t);
    fp_mul(s1, s1, hhh);
    fp_sub(r.Y, t, s1);                    // Y3 = r (v - X3) - s1 h^3
    fp_mul(t, a.Z, b.Z);
    fp_mul(r.Z, t, h);                     // Z3 = Z1 Z2 h
}

// Jacobian + affine (8M + 3S), same contract as pt_add.
FAB_HD void pt_add_mixed(jac& r, const jac& a, const u256& bx, const u256& by, bool& h_zero, bool& r_zero) {
    u256 z1z1, u2, s2, h, rr, hh, hhh, v, t;
    fp_sqr(z1z1, a.Z);
    fp_mul(u2, bx, z1z1);
    fp_mul(t, a.Z, z1z1);
    fp_mul(s2, by, t);
    fp_sub(h, u2, a.X);
    fp_sub(rr, s2, a.Y);
    h_zero = is_zero(h);
    r_zero = is_zero(rr);
    fp_sqr(hh, h);
    fp_mul(hhh, hh, h);
    fp_mul(v, a.X, hh);
    fp_sqr(t, rr);
    fp_sub(t, t, hhh);
    fp_sub(t, t, v);
    fp_sub(r.X, t, v);
    fp_sub(t, v, r.X);
    fp_mul(t, rr, t);
    fp_mul(s2, a.Y, hhh);
    fp_sub(r.Y, t, s2);
    fp_mul(r.Z, a.Z, h);
}

FAB_HD void sel_jac(jac& r, bool c, const jac& a, const jac& b) {
    sel256(r.X, c, a.X, b.X);
    sel256(r.Y, c, a.Y, b.Y);
    sel256(r.Z, c, a.Z, b.Z);
}

FAB_HD uint32_t nibble(const u256& k, int i) { return (k.w[i >> 3] >> ((i & 7) * 4)) & 15u; }

// Early (pre-arithmetic) status in the reference's order of checks:
//   r,s > 0 (bccsp/utils/ecdsa.go:59-64) -> low-S (bccsp/sw/ecdsa.go:47-54) -> r < n (ecdsa.Verify)
FAB_HD uint32_t range_status(const u256& r, const u256& s) {
    const u256 N = FAB_P256_N;
    const u256 HALF = FAB_P256_HALF_N;
    uint32_t st = ST_VALID;
    if (!lt256(r, N)) st = ST_RANGE;
    if (lt256(HALF, s)) st = ST_HIGH_S;
    if (is_zero(r) | is_zero(s)) st = ST_RANGE;
    return st;
}

// The verification core.  GTab provides  void load(int window, uint32_t digit /*1..15*/, u256& x, u256& y).
// qtab: 16 jac entries of per-lane storage (entry 0 unused filler).
// Inputs are plain integers (limbs of the big-endian C-ABI fields); e is hashToInt(digest).
template <class GTab>
FAB_HD uint32_t p256_verify_core(const u256& qx, const u256& qy, const u256& e, const u256& r, const u256& s,
                                 const GTab& gtab, jac* qtab) {
    const u256 P = FAB_P256_P;
    const u256 N = FAB_P256_N;
    const u256 ONE = FAB_P256_R1;
    uint32_t early = range_status(r, s);

    // --- public key to Montgomery form + curve membership (reference: enforced at key import) ---
    bool q_in_field = lt256(qx, P) & lt256(qy, P);
    jac Q;
    fp_to_mont(Q.X, qx);
    fp_to_mont(Q.Y, qy);
    Q.Z = ONE;
    bool q_ok = q_in_field && on_curve_mont(Q.X, Q.Y);
    if (early == ST_VALID && !q_ok) early = ST_OFF_CURVE;

    // --- scalars: w = s^-1, u1 = e w, u2 = r w  (mod n) ---
    u256 sm, wm, em, rm, u1, u2, ered, t;
    fn_to_mont(sm, s);
    fn_inv(wm, sm);
    uint32_t br = sub256(t, e, N);          // e < 2^256 < 2n: one conditional subtraction
    sel256(ered, br == 0, t, e);
    fn_to_mont(em, ered);
    fn_to_mont(rm, r);
    fn_mul(u1, em, wm);
    fn_from_mont(u1, u1);
    fn_mul(u2, rm, wm);
    fn_from_mont(u2, u2);

    // --- per-lane table j*Q, j = 1..15 ---
    qtab[0] = Q;
    qtab[1] = Q;
    for (int j = 2; j < 16; j += 2) {
        jac d, a;
        bool hz, rz;
        jac half = qtab[j >> 1];
        pt_dbl(d, half);
        qtab[j] = d;
        pt_add_mixed(a, d, Q.X, Q.Y, hz, rz);
        qtab[j + 1] = a;
    }

    // --- T = u2 * Q ---
    jac T = Q;
    bool t_inf = true;
    for (int i = 63; i >= 0; i--) {
        if (i != 63) {
            pt_dbl(T, T);
            pt_dbl(T, T);
            pt_dbl(T, T);
            pt_dbl(T, T);
        }
        uint32_t d = nibble(u2, i);
        jac ent = qtab[d];
        jac sum;
        bool hz, rz;
        pt_add(sum, T, ent, hz, rz);
        bool take_ent = t_inf & (d != 0);
        bool take_sum = (!t_inf) & (d != 0);
        sel_jac(T, take_sum, sum, T);
        sel_jac(T, take_ent, ent, T);
        t_inf = t_inf & (d == 0);
    }

    // --- S = u1 * G (comb) ---
    jac S = Q;
    bool s_inf = true;
    for (int i = 0; i < G_WINDOWS; i++) {
        uint32_t d = nibble(u1, i);
        u256 gx, gy;
        gtab.load(i, d ? d : 1u, gx, gy);
        jac sum, ent;
        bool hz, rz;
        pt_add_mixed(sum, S, gx, gy, hz, rz);
        ent.X = gx;
        ent.Y = gy;
        ent.Z = ONE;
        bool take_ent = s_inf & (d != 0);
        bool take_sum = (!s_inf) & (d != 0);
        sel_jac(S, take_sum, sum, S);
        sel_jac(S, take_ent, ent, S);
        s_inf = s_inf & (d == 0);
    }

    // --- R = S + T with the exceptional cases of the group law (Appendix A step 8) ---
    jac Rp, Rd;
    bool hz, rz;
    pt_add(Rp, S, T, hz, rz);
    pt_dbl(Rd, T);
    bool r_inf = t_inf & s_inf;                       // cannot happen for u2 != 0; kept for completeness
    bool use_T = s_inf & !t_inf;
    bool use_S = t_inf & !s_inf;
    bool both = !s_inf & !t_inf;
    bool use_dbl = both & hz & rz;                    // S == T
    r_inf = r_inf | (both & hz & !rz);                // S == -T  -> point at infinity
    jac Rr = Rp;
    sel_jac(Rr, use_dbl, Rd, Rr);
    sel_jac(Rr, use_T, T, Rr);
    sel_jac(Rr, use_S, S, Rr);

    // --- x(R) mod n == r  without inverting Z ---
    const u256 PMN = FAB_P256_P_MINUS_N;
    u256 zz, rmp, rhs, r2;
    fp_sqr(zz, Rr.Z);
    fp_to_mont(rmp, r);
    fp_mul(rhs, rmp, zz);
    bool ok = eq256(rhs, Rr.X);
    add256(r2, r, N);                                  // only meaningful when r < p - n (no wrap)
    fp_to_mont(rmp, r2);
    fp_mul(rhs, rmp, zz);
    ok = ok | (lt256(r, PMN) & eq256(rhs, Rr.X));
    ok = ok & !r_inf;

    uint32_t st = ok ? ST_VALID : ST_BAD_MATH;
    return early != ST_VALID ? early : st;
}

}  // namespace fab
