// secp256r1 point arithmetic on fe29 field elements + the ECDSA verification core of the verify kernels.
//
// What it replaces: Go 1.14 crypto/ecdsa.Verify + crypto/elliptic CombinedMult, reached from
// bccsp/sw/ecdsa.go:56.  Structure (DESIGN.md "Kernels"):
//   * one signature per lane; lane-uniform control flow (flags + selects);
//   * w = s^-1 mod n by safegcd (modinv30.h); u1 = e w, u2 = r w by Montgomery products mod n (fp256.h);
//   * u2*Q : 52 signed 5-bit (Booth) windows over a 16-entry per-lane Jacobian table in a global workspace (ec29.h);
//   * u1*G : 16-window 16-bit comb over a precomputed affine table (80 MiB, Infinity-Cache resident), mixed additions only;
//   * the two partial sums stay in SEPARATE accumulators, so for an on-curve Q no addition inside either loop can
//     meet P == +-Q (proof in DESIGN.md); only the final addition handles doubling / infinity explicitly;
//   * no field inversion: x(R) mod n == r is tested as X == r Z^2 or X == (r + n) Z^2.
//
// Limb-magnitude bookkeeping (fe29.h): every product below is annotated  [L(a) x L(b)]  in units of 2^28; the bound is 14.
#pragma once
#include "ec29.h"
#include "modinv30.h"
#include "p256_point.h"  // status codes, nibble(), range_status() shared with the host-side u256 code

namespace fab {

typedef jac_t<fe> jac29;
typedef LocalQTab<fe> LocalQTab29;

}  // namespace fab
#if defined(__HIP_DEVICE_COMPILE__)
#include "one29_gcn.h"   // the three point operations for the device, generated (gen_pair_gcn.py one): see pt_dbl29 below
#endif
namespace fab {

#if defined(__HIP_DEVICE_COMPILE__)
// On the device pt_add29 / pt_add_mixed29 of ec29.h resolve to these for P-256 (a non-template overload wins over the template): the same
// formulas with round 6's shortcuts - subtractions made inside the products, no carry passes, unsigned digits where the interval proof of
// gen_pair_gcn.one_contracts_closed allows them (2 374 / 1 642 instructions instead of 2 678 / 1 900).  The templates stay: host build,
// FP256BN, and the specification.  Contract of a state between operations: gen_pair_gcn.STATE_ONE (digits of X, Y, Z in [-2^28, 2^29)).
FAB_D void pt_add29(jac29& r, const jac29& a, const jac29& b, fe& h, fe& rr) {
    fe x3, y3, z3, hh, rrr;
    one29_add(x3, y3, z3, hh, rrr, a.X, a.Y, a.Z, b.X, b.Y, b.Z);
    r.X = x3;
    r.Y = y3;
    r.Z = z3;
    h = hh;
    rr = rrr;
}
FAB_D void pt_add_mixed29(jac29& r, const jac29& a, const fe& bx, const fe& by, fe& h, fe& rr) {
    fe x3, y3, z3, hh, rrr;
    one29_madd(x3, y3, z3, hh, rrr, a.X, a.Y, a.Z, bx, by);
    r.X = x3;
    r.Y = y3;
    r.Z = z3;
    h = hh;
    rr = rrr;
}
#endif

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
#if defined(__HIP_DEVICE_COMPILE__)
    {   // the generated form (one29_gcn.h): 1 201 instructions instead of 1 339; out: digits of X, Y, Z in [-2^28, 2^29)
        fe x3, y3, z3;
        one29_dbl(x3, y3, z3, a.X, a.Y, a.Z);
        r.X = x3;
        r.Y = y3;
        r.Z = z3;
        return;
    }
#endif
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
// crypto/elliptic.  r_inf reports the point at infinity (then Rr is meaningless).  The two partial sums stay in SEPARATE
// accumulators (ec29.h: var_base_mult29 over a per-lane table of Q, comb_mult29 over the generator table).
template <class GTab, class QTab>
FAB_HD void p256_combined_mult29(jac29& Rr, bool& r_inf, const u256& u1, const u256& u2, const jac29& Q, const GTab& gtab, QTab& qtab) {
    jac29 T, S;
    bool t_inf, s_inf;
    var_base_mult29(T, t_inf, u2, Q, qtab);
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
