// ECDSA P-256 verification for a REGISTERED key on EIGHT LANES per signature, in two phases - for launches that cannot fill the chip
// (a block of a few hundred transactions: sampleconfig/configtx.yaml:284 MaxMessageCount 500) and whose time is therefore the length
// of one wavefront's instruction stream.
//
// A keyed verification (p256_verify29.h p256_verify_keyed_core29) is  w = s^-1 mod n,  u1 = e w,  u2 = r w,  then
//     R = sum over 16 windows of G16[w][digit_w(u1)]  +  sum over 32 windows of K8[w][digit_w(u2)]
// - 48 precomputed points, no doublings - and a SUM of points splits over lanes in any way: every lane adds up the entries of a few
// windows (mixed additions), three rounds of lane exchanges (__shfl_xor 1, 2, 4) add the eight partial sums up.  78 000 instructions
// per wavefront on two lanes become 18 000 + 42 000 on eight, and the split is placed where the digest enters:
//     PRE   everything that does not need the digest - range gates, w (safegcd: 25 000 instructions, serial whatever the lanes do),
//           u2, T = u2 Q - runs WHILE the message is still being hashed (a creator signs a whole envelope payload: 76 SHA-256 blocks,
//           225 us on one lane); leaves w and T in 144 bytes of scratch per signature;
//     POST  u1 = e w, S = u1 G, R = S + T with the exceptional cases of the group law (ec29.h final_add29), x(R) mod n == r.
// Only POST is on the critical path behind the hash: 37 us instead of 155.
//
// Why no addition inside a sum can meet P == +-Q (the formulas of pt_add_mixed29 / pt_add29 do not cover it): every partial sum is
// a B with a = the scalar's digits on a SUBSET of windows, so 0 <= a <= k < n for the scalar k (u1 or u2, reduced mod n); two
// operands of one addition have DISJOINT window sets, a1 + a2 <= k < n.  a1 B == +-a2 B needs a1 == a2 (disjoint sets: both zero -
// the infinity flags) or a1 + a2 == n (impossible).  The kernel still tests h == 0 in every tree addition and reports the signature
// as "not decided here" if it ever happens (status 6: bccsp/sw decides) - an assertion that costs one comparison, not a code path
// that is expected to run.  S + T (different base points) goes through final_add29, which handles doubling and infinity.
#pragma once
#include "p256_verify29.h"

namespace fab {

constexpr int WIDE_LANES = 8;                // lanes per signature
constexpr int WIDE_MAX = 8192;               // signatures per launch up to which the wide form is used: 1024 wavefronts, one per SIMD
constexpr int WIDE_SCRATCH_WORDS = 36;       // per signature between the phases: T (27 limbs), w (8 words), flags
constexpr uint32_t WIDE_F_TINF = 1u, WIDE_F_EXC = 2u, WIDE_F_BADKEY = 4u;   // flags word: | early status << 8

#if defined(__HIPCC__)
// the partner's value (lane ^ mask), limb by limb
template <class F>
__device__ __forceinline__ void wide_exchange(jac_t<F>& o, const jac_t<F>& s, int mask) {
#pragma unroll
    for (int l = 0; l < 9; l++) {
        o.X.v[l] = __shfl_xor(s.X.v[l], mask, 64);
        o.Y.v[l] = __shfl_xor(s.Y.v[l], mask, 64);
        o.Z.v[l] = __shfl_xor(s.Z.v[l], mask, 64);
    }
}
// S = S + (the partner lane's partial sum), infinity flags honoured, h == 0 reported (see the header)
template <class F>
__device__ __forceinline__ void wide_tree_add(jac_t<F>& S, bool& s_inf, bool& exc, int mask) {
    jac_t<F> O, sum;
    F h, rr;
    wide_exchange(O, S, mask);
    const bool o_inf = __shfl_xor((int)s_inf, mask, 64) != 0;
    pt_add29(sum, S, O, h, rr);            // in: L(X) = 1, L(Y) <= 2, L(Z) = 1 on both sides (outputs of pt_add_mixed29 / pt_add29, or a table entry)
    const bool both = !s_inf & !o_inf;
    exc = exc | (both & fe_is_zero(h));
    sel_jac29(S, both, sum, S);
    sel_jac29(S, s_inf & !o_inf, O, S);
    s_inf = s_inf & o_inf;
}
// the entries of windows [w0, w0 + nw) of k's comb, added up on this lane
template <class Tab, class F>
__device__ __forceinline__ void wide_comb_part(jac_t<F>& S, bool& s_inf, const u256& k, const Tab& tab, const jac_t<F>& seed, int w0, int nw) {
    F ONE;
    fe_set_one(ONE);
    S = seed;
    s_inf = true;
    uint32_t nd = Tab::digit(k, w0);
    F nx, ny;
    tab.load(w0, nd ? nd : 1u, nx, ny);
#pragma unroll 1
    for (int i = 0; i < nw; i++) {
        const uint32_t d = nd;
        jac_t<F> ent, sum;
        F h, rr;
        ent.X = nx;
        ent.Y = ny;
        ent.Z = ONE;
        const int inext = i + 1 < nw ? w0 + i + 1 : w0 + i;
        nd = Tab::digit(k, inext);
        tab.load(inext, nd ? nd : 1u, nx, ny);
        pt_add_mixed29(sum, S, ent.X, ent.Y, h, rr);
        const bool take_ent = s_inf & (d != 0);
        const bool take_sum = (!s_inf) & (d != 0);
        sel_jac29(S, take_sum, sum, S);
        sel_jac29(S, take_ent, ent, S);
        s_inf = s_inf & (d == 0);
    }
}
// the eight partial sums of a signature's lanes -> their sum, on all eight lanes
template <class F>
__device__ __forceinline__ void wide_tree(jac_t<F>& S, bool& s_inf, bool& exc) {
#pragma unroll 1
    for (int mask = 1; mask < WIDE_LANES; mask <<= 1) wide_tree_add(S, s_inf, exc, mask);
}

// PRE: digest-independent half.  r, s: this signature's fields; kt: its key's comb table.  Lane `sub` (0..7) of the signature's group.
__device__ __forceinline__ void p256_wide_pre29(const u256& r, const u256& s, const GTab16& gtab, const KeyTab8& kt, uint32_t sub, bool kok, bool store,
                                                int32_t* __restrict__ scratch) {
    const fe ONE = {FE29_R1};
    const uint32_t early = range_status(r, s);
    u256 w, u2, t;
    {
        const modinv_info NI = MODINV_N_INFO;
        modinv(w, s, NI);                       // s >= n only on signatures already rejected by the low-S gate
    }
    fn_to_mont(t, r);
    fn_mul(u2, t, w);
    jac29 seed, T;
    gtab.load(0, 1u, seed.X, seed.Y);
    seed.Z = ONE;
    bool t_inf, exc = false;
    wide_comb_part(T, t_inf, u2, kt, seed, (int)sub * (KeyTab8::WINDOWS / WIDE_LANES), KeyTab8::WINDOWS / WIDE_LANES);
    wide_tree(T, t_inf, exc);
    if (store) {
#pragma unroll
        for (int l = 0; l < 9; l++) {
            scratch[l] = T.X.v[l];
            scratch[9 + l] = T.Y.v[l];
            scratch[18 + l] = T.Z.v[l];
        }
#pragma unroll
        for (int l = 0; l < 8; l++) scratch[27 + l] = (int32_t)w.w[l];
        scratch[35] = (int32_t)((t_inf ? WIDE_F_TINF : 0u) | (exc ? WIDE_F_EXC : 0u) | (kok ? 0u : WIDE_F_BADKEY) | (early << 8));
    }
}
// POST: e = hashToInt(digest).  Returns the status (ST_* of fabgpu.h, or 6 = not decided here) - the same on all eight lanes.
__device__ __forceinline__ uint32_t p256_wide_post29(const u256& e, const u256& r, const GTab16& gtab, uint32_t sub, const int32_t* __restrict__ scratch) {
    const u256 N = FAB_P256_N;
    const fe ONE = {FE29_R1};
    jac29 T;
    u256 w;
#pragma unroll
    for (int l = 0; l < 9; l++) {
        T.X.v[l] = scratch[l];
        T.Y.v[l] = scratch[9 + l];
        T.Z.v[l] = scratch[18 + l];
    }
#pragma unroll
    for (int l = 0; l < 8; l++) w.w[l] = (uint32_t)scratch[27 + l];
    const uint32_t flags = (uint32_t)scratch[35];
    const uint32_t early = flags >> 8;
    u256 ered, t, u1;
    const uint32_t br = sub256(t, e, N);        // e < 2^256 < 2n: one conditional subtraction
    sel256(ered, br == 0, t, e);
    fn_to_mont(t, ered);
    fn_mul(u1, t, w);
    jac29 seed, S, Rr;
    gtab.load(0, 1u, seed.X, seed.Y);
    seed.Z = ONE;
    bool s_inf, r_inf, exc = (flags & WIDE_F_EXC) != 0;
    wide_comb_part(S, s_inf, u1, gtab, seed, (int)sub * (GTab16::WINDOWS / WIDE_LANES), GTab16::WINDOWS / WIDE_LANES);
    wide_tree(S, s_inf, exc);
    final_add29(Rr, r_inf, S, s_inf, T, (flags & WIDE_F_TINF) != 0);
    const bool ok = x_equals_r29(Rr, r_inf, r);
    uint32_t st = ok ? ST_VALID : ST_BAD_MATH;
    if (exc) st = 6u;                           // TUPLE_ST_NEEDS_SW / "not decided here": see the header (never observed)
    if (early != ST_VALID) st = early;
    if (flags & WIDE_F_BADKEY) st = ST_OFF_CURVE;
    return st;
}
#endif  // __HIPCC__

}  // namespace fab
