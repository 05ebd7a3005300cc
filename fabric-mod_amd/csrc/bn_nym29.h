// Idemix pseudonym-signature verification on FP256BN's G1 (y^2 = x^3 + 3 over the 256-bit BN prime), fe29 representation.
//
// What it replaces: idemix/nymsignature.go:74-109 (NymSignature.Ver), reached from msp/idemixmsp.go:584-599 through
// bccsp/idemix/handlers/nymsigner.go:62-95 and bccsp/idemix/bridge/nymsignaturescheme.go:66-89; the arithmetic under it is
// github.com/hyperledger/fabric-amcl amcl/FP256BN (ECP.Mul2, ECP.Mul, ECP.Sub, ECP.ToBytes; third-party, go.mod:44).
//
//     t  = HSk * s_sk + HRand * s_rnym - Nym * c                       (three G1 scalar multiplications, NO pairing)
//     c' = H("sign" || t || Nym || ipk.Hash || msg) mod r ;  valid  <=>  c == H(c' || nonce) mod r
//
// Structure on the device (one signature per lane, lane-uniform control flow):
//   * HSk and HRand are fixed per issuer: each gets an 8-bit comb table at fabgpu_idemix_issuer_register (32 mixed additions
//     per scalar, no doublings) - the same CombTab<8> layout as a registered P-256 key;
//   * Nym is fresh per signature: c * Nym = k1 * Nym + k2 * phi(Nym) with the GLV endomorphism phi(x, y) = (beta x, y) and
//     |k1|, |k2| < 2^129: 27 signed 5-bit windows over one per-lane table, 130 doublings instead of 255 (ec29.h: glv_mult29);
//   * the three partial results stay in separate accumulators and are merged by two final additions that handle doubling and
//     the point at infinity explicitly;
//   * one inversion mod p (safegcd, modinv30.h) gives the affine t that the hash needs.
// Inputs outside the domain on which the CPU oracle is pinned to the reference's fixtures (oracle/idemix_oracle.py: Nym off the
// curve or with coordinates >= p, s-values >= r, t at infinity) are NOT decided here: the status says "ask bccsp/sw".
//
// Limb bookkeeping: products annotated [L(a) x L(b)] in units of 2^28; the bound for this field is 12 (bn29.h).
#pragma once
#include "bn29.h"
#include "ec29.h"
#include "modinv30.h"

namespace fab {

typedef jac_t<fbn> jacbn;

// status codes of the nym-signature verbs (include/fabgpu.h FABGPU_NYM_*)
enum : uint32_t {
    NYM_VALID = 0,
    NYM_BAD_PROOF = 1,     // "pseudonym signature invalid: zero-knowledge proof is invalid" (idemix/nymsignature.go:105)
    NYM_NEEDS_SW = 6,      // outside the pinned domain: the caller must ask bccsp/sw
};

// y^2 == x^3 + 3, Montgomery form, x and y normalised
FAB_HD bool bn_on_curve29(const fbn& x, const fbn& y) {
    const fbn B = {BN29_B_MONT};
    fbn l, x2, x3, d;
    fe_sqr(l, y);          // [1x1]
    fe_sqr(x2, x);         // [1x1]
    fe_mul(x3, x2, x);     // [1x1]
    fe_sub(d, l, x3);      // L2
    fe_sub(d, d, B);       // L3
    return fe_is_zero(d);
}

// Doubling, a = 0 (dbl-2009-l with D = 4 X Y^2 as one product): 3M + 4S.
// in: L(X) <= 2, L(Y) <= 3, L(Z) <= 2.   out: L(X) = 1, L(Y) = 3, L(Z) = 2.
FAB_HD void pt_dbl29(jacbn& r, const jacbn& a) {
    fbn A, Bq, b2, c4, x4, D, E, F, t, x3, yy, yz;
    fe_sqr(A, a.X);                // [2x2]
    fe_sqr(Bq, a.Y);               // [3x3]
    fe_add(b2, Bq, Bq);            // L2
    fe_sqr(c4, b2);                // [2x2]  4 Y^4
    fe_add(x4, a.X, a.X);
    fe_add(x4, x4, x4);            // 4X   L(X) = 1 on every path that reaches a doubling -> L4 (L8 is the int32 edge)
    fe_mul(D, x4, Bq);             // [4x1]  4 X Y^2
    fe_add(E, A, A);
    fe_add(E, E, A);               // 3 X^2   L3
    fe_sqr(F, E);                  // [3x3]
    fe_sub(t, F, D);
    fe_sub(t, t, D);               // E^2 - 2D   L3
    fe_weak_norm(x3, t);           // L1
    fe_sub(t, D, x3);              // L2
    fe_mul(yy, E, t);              // [3x2]
    fe_sub(yy, yy, c4);
    fe_sub(r.Y, yy, c4);           // E (D - X3) - 8 Y^4   L3
    fe_mul(yz, a.Y, a.Z);          // [3x2]
    fe_add(r.Z, yz, yz);           // 2 Y Z   L2
    r.X = x3;
}

FAB_HD void bn_neg29(jacbn& p) {
#pragma unroll
    for (int l = 0; l < 9; l++) p.Y.v[l] = -p.Y.v[l];
}

// x < m ?  (both plain 256-bit integers)
FAB_HD bool bn_lt(const u256& x, const u256& m) { return lt256(x, m); }

// GLV decomposition of a scalar for phi(x, y) = (beta x, y) = lambda (x, y), lambda = 36u^4 - 1:  k = k1 + k2 lambda (mod r) with
// |k1|, |k2| < 2^129.  With the reduced lattice basis v1 = (a1, b1), v2 = (a2, b2) of {(a, b): a + b lambda = 0 mod r}:
//     c1 ~ b2 k / r,  c2 ~ -b1 k / r;   k1 = k - c1 a1 - c2 a2,   k2 = -c1 b1 - c2 b2
// ANY integers c1, c2 give a correct decomposition (only the size of k1, k2 depends on how well they approximate the
// quotients), so the quotients are taken as floor(k G / 2^384) with precomputed G = floor(2^384 b / r): no division on the
// device, and the bound 2^129 is checked exhaustively-at-random in tests/test_idemix_oracle.py against gen_bn_consts.py.
// Out: magnitudes and signs.  (Arithmetic is mod 2^256, two's complement; a1, a2, -b1, b2 are positive.)
FAB_HD void bn_glv_decompose(u256& m1, bool& n1, u256& m2, bool& n2, const u256& k) {
    const u256 G1 = FAB_BN_GLV_G1, G2 = FAB_BN_GLV_G2, A1 = FAB_BN_GLV_A1, A2 = FAB_BN_GLV_A2, NB1 = FAB_BN_GLV_NB1, B2 = FAB_BN_GLV_B2;
    uint32_t t[16];
    u256 c1 = zero256(), c2 = zero256(), hi = zero256(), p, k1, k2;
    mul512(t, k, G1);
#pragma unroll
    for (int i = 0; i < 4; i++) c1.w[i] = t[12 + i];              // (k G1) >> 384
    mul512(t, k, G2);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        c2.w[i] = t[12 + i];
        hi.w[i] = k.w[4 + i];                                     // k >> 128  (G2 stands for 2^256 + G2)
    }
    add256(c2, c2, hi);
    // k1 = k - c1 a1 - c2 a2
    mul512(t, c1, A1);
#pragma unroll
    for (int i = 0; i < 8; i++) p.w[i] = t[i];
    sub256(k1, k, p);
    mul512(t, c2, A2);
#pragma unroll
    for (int i = 0; i < 8; i++) p.w[i] = t[i];
    sub256(k1, k1, p);
    // k2 = c1 (-b1) - c2 b2
    mul512(t, c1, NB1);
#pragma unroll
    for (int i = 0; i < 8; i++) k2.w[i] = t[i];
    mul512(t, c2, B2);
#pragma unroll
    for (int i = 0; i < 8; i++) p.w[i] = t[i];
    sub256(k2, k2, p);
    const u256 Z = zero256();
    u256 neg;
    n1 = (k1.w[7] >> 31) != 0;
    sub256(neg, Z, k1);
    sel256(m1, n1, neg, k1);
    n2 = (k2.w[7] >> 31) != 0;
    sub256(neg, Z, k2);
    sel256(m2, n2, neg, k2);
}

// ---- pieces shared by the one-lane and the two-lane form of the commitment -------------------------------------------------
// input gates + the pseudonym in Montgomery form.  early: NYM_BAD_PROOF when c >= r (an unreduced ProofC can never equal a value
// reduced mod r: idemix/nymsignature.go:104 compares BIGs); dom: inside the pinned domain.
FAB_HD void bn_nym_gates29(uint32_t& early, bool& dom, jacbn& N, const u256& nx, const u256& ny, const u256& c, const u256& s_sk,
                           const u256& s_rnym) {
    const u256 P = FAB_BN_P, R = FAB_BN_R;
    early = bn_lt(c, R) ? (uint32_t)NYM_VALID : (uint32_t)NYM_BAD_PROOF;
    dom = bn_lt(nx, P) & bn_lt(ny, P) & bn_lt(s_sk, R) & bn_lt(s_rnym, R);
    fe_to_mont(N.X, nx);
    fe_to_mont(N.Y, ny);
    fe_set_one(N.Z);
    dom = dom & bn_on_curve29(N.X, N.Y);
}
// Jacobian -> affine plain integers in [0, p): one inversion mod p (safegcd)
FAB_HD void bn_affine29(u256& tx, u256& ty, const jacbn& W) {
    u256 zp, zi;
    fe_from_mont(zp, W.Z);
    {
        const modinv_info PI = MODINV_BNP_INFO;
        modinv(zi, zp, PI);
    }
    fbn zm, zi2, zi3, ax, ay;
    fe_to_mont(zm, zi);
    fe_sqr(zi2, zm);               // [1x1]
    fe_mul(zi3, zi2, zm);          // [1x1]
    fe_mul(ax, W.X, zi2);          // [1x1]
    fe_mul(ay, W.Y, zi3);          // [3x1]
    fe_from_mont(tx, ax);
    fe_from_mont(ty, ay);
}

// The commitment t of the verification equation, affine, as plain integers in [0, p) - one signature per lane.
// Returns NYM_VALID when (tx, ty) is meaningful, NYM_BAD_PROOF when c >= r, NYM_NEEDS_SW outside the pinned domain.
// KTab: comb tables of HSk and HRand; QTab: per-lane store(j, point) / load(j, point), j = 1..16.
template <class KTab, class QTab>
FAB_HD uint32_t bn_nym_commitment29(u256& tx, u256& ty, const u256& nx, const u256& ny, const u256& c, const u256& s_sk,
                                    const u256& s_rnym, const KTab& hsk, const KTab& hrand, QTab& qtab) {
    uint32_t early;
    bool dom;
    jacbn N;
    bn_nym_gates29(early, dom, N, nx, ny, c, s_sk, s_rnym);

    jacbn seed, S1, S2, U, T, W;
    bool s1_inf, s2_inf, u_inf, t_inf, w_inf;
    hsk.load(0, 1u, seed.X, seed.Y);
    fe_set_one(seed.Z);
    comb_mult29(S1, s1_inf, s_sk, hsk, seed);
    comb_mult29(S2, s2_inf, s_rnym, hrand, seed);
    final_add29(U, u_inf, S1, s1_inf, S2, s2_inf);
    // c * Nym through the GLV endomorphism: c = k1 + k2 lambda, 130 doublings instead of 255
    u256 m1, m2;
    bool n1, n2, exc;
    bn_glv_decompose(m1, n1, m2, n2, c);
    const fbn BETA = {BN29_BETA_MONT};
    glv_mult29(T, t_inf, exc, m1, n1, m2, n2, N, BETA, qtab);
    bn_neg29(T);
    final_add29(W, w_inf, U, u_inf, T, t_inf);
    bn_affine29(tx, ty, W);

    if (early != NYM_VALID) return early;
    if (!dom || w_inf || exc) return NYM_NEEDS_SW;
    return NYM_VALID;
}

// ---- two lanes per signature (batches that cannot fill the chip: the idemix creators of one block) ---------------------------
// The verification equation splits into two halves with IDENTICAL instruction streams on different data:
//     even lane:  HSk   * s_sk   -  k1 * Nym            odd lane:  HRand * s_rnym  -  k2 * phi(Nym)         (c = k1 + k2 lambda)
// - one comb multiplication (32 mixed additions) and one 27-window single-scalar Booth multiplication (130 doublings, 27
// additions) each, instead of two combs and an interleaved 54-addition loop on one lane: the length of a lane's stream,
// which is what a small batch's latency is made of, drops by a third.  The single-scalar loop keeps the proof that no
// addition meets P == +-Q (|k_i| < 2^129 < r), so there is no collision flag in this form.  The lanes then exchange their
// partial sums (27 limbs by DPP) and both finish t = (even) + (odd), the inversion and the hashes; the even lane reports.
//
// Part 1 is what one lane computes on its own; the caller exchanges (DPP on the device, plain copies in the host tests) and
// calls part 2.
struct bn_nym_half {
    jacbn P;          // this lane's partial sum
    bool inf;
    uint32_t early;   // gates (identical on both lanes)
    bool dom;
};
template <class KTab, class QTab>
FAB_HD void bn_nym_split_part1(bn_nym_half& out, bool odd, const u256& nx, const u256& ny, const u256& c, const u256& s_sk, const u256& s_rnym,
                               const KTab& hsk, const KTab& hrand, QTab& qtab) {
    jacbn N;
    bn_nym_gates29(out.early, out.dom, N, nx, ny, c, s_sk, s_rnym);
    // this lane's base table and scalar
    KTab tab{odd ? hrand.w : hsk.w};
    u256 sc;
    sel256(sc, odd, s_rnym, s_sk);
    jacbn seed, S, T;
    bool s_inf, t_inf;
    hsk.load(0, 1u, seed.X, seed.Y);
    fe_set_one(seed.Z);
    comb_mult29(S, s_inf, sc, tab, seed);
    // this lane's half of c * Nym
    u256 m1, m2, m;
    bool n1, n2;
    bn_glv_decompose(m1, n1, m2, n2, c);
    sel256(m, odd, m2, m1);
    bool neg = odd ? n2 : n1;
    const fbn BETA = {BN29_BETA_MONT};
    fbn bx;
    fe_mul(bx, N.X, BETA);                       // [1x1]  phi(Nym) = (beta x, y)
    fe_sel(N.X, odd, bx, N.X);
    booth_mult29<GLV_WINDOWS>(T, t_inf, m, N, qtab);
    // partial = S - (+-T): subtracting, so the Y of T flips unless the half-scalar was negative
#pragma unroll
    for (int l = 0; l < 9; l++) T.Y.v[l] = neg ? T.Y.v[l] : -T.Y.v[l];
    final_add29(out.P, out.inf, S, s_inf, T, t_inf);
}
// mine / theirs: the two partial sums (addition is commutative: both lanes arrive at the same t)
FAB_HD uint32_t bn_nym_split_part2(u256& tx, u256& ty, const bn_nym_half& mine, const jacbn& theirs, bool theirs_inf) {
    jacbn W;
    bool w_inf;
    final_add29(W, w_inf, mine.P, mine.inf, theirs, theirs_inf);
    bn_affine29(tx, ty, W);
    if (mine.early != NYM_VALID) return mine.early;
    if (!mine.dom || w_inf) return NYM_NEEDS_SW;
    return NYM_VALID;
}

// x mod r for x < 2^256 (r > 2^255: one conditional subtraction) - the Mod of HashModOrder, idemix/util.go:49
FAB_HD void bn_mod_order(u256& out, const u256& x) {
    const u256 R = FAB_BN_R;
    u256 t;
    uint32_t br = sub256(t, x, R);
    sel256(out, br == 0, t, x);
}

}  // namespace fab
