// ECDSA P-256 verification with TWO LANES PER SIGNATURE (device only) - the small-batch variant of p256_verify29.h.
//
// Same algorithm, same gates, same window recodings, same tables as p256_verify_core29 (which it must agree with bit for
// bit: tests/test_gpu_parity.py runs both against the oracle); what changes is the shape of the instruction stream.  A
// 30 000-signature block is 469 wavefronts on 1024 SIMDs and one wave issues one VALU instruction per ~4.2 cycles, so the
// kernel time is the LENGTH of the per-wave stream.  Lanes 2k (E) and 2k+1 (O) share signature k: the point operations are
// the generated programs of pair29_gcn.h (732 / 1364 / 1104 instructions for dbl / add / madd - round 6; one lane: 1201 / 2374 / 1642),
// every field product is executed by both lanes on different operands, limbs cross lanes with DPP quad_perm:[1,0,3,2].
// Between operations  E holds A = X, B = Y  and  O holds B = Z.  The scalar part (gates, s^-1 mod n, u1, u2, digits) is
// computed redundantly by both lanes.
//
// Replaces crypto/ecdsa.Verify reached from bccsp/sw/ecdsa.go:56.
#pragma once
#include "p256_verify29.h"
#include "pair29_gcn.h"

namespace fab {

struct pair_pt {
    fe A, B;   // E: X, Y      O: don't-care, Z
};

__device__ __forceinline__ int32_t pair_swap_i32(int32_t v) {   // partner lane's value (quad_perm:[1,0,3,2])
    return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
}
__device__ __forceinline__ void pair_swap_fe(fe& r, const fe& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = pair_swap_i32(a.v[i]);
}
__device__ __forceinline__ void pair_sel(pair_pt& r, bool c, const pair_pt& a, const pair_pt& b) {
    fe_sel(r.A, c, a.A, b.A);
    fe_sel(r.B, c, a.B, b.B);
}

#define PAIR_TMPS __attribute__((unused)) fe tU1, tU2, tU3, tU4, tU6, tW, tH, tRR, tP1, tP2, tT0, tT1, tTD
#define PAIR_DBL(P) PAIR29_DBL((P).A, (P).B, tU1, tU2, tU3, tW, tP1, tP2, tT0, tT1, tTD)
#define PAIR_ADD(R, P, C, D) PAIR29_ADD((R).A, (R).B, (P).A, (P).B, tH, tRR, tW, tU1, tU2, tU3, tU4, tU6, tP1, tP2, tT0, tT1, tTD, C, D)
#define PAIR_MADD(R, P, C, D) PAIR29_MADD((R).A, (R).B, (P).A, (P).B, tU1, tU2, tU3, tU4, tH, tRR, tP1, tP2, tT0, tT1, tTD, C, D)

// Per-signature table j*Q (j = 1..16) in the global workspace.  Per workgroup slot: [entry][q 0..7][pair NP] x 16 bytes;
// q 0..4 = X[9] Y[9] pad, q 5..7 = Z[9] pad.  E stores / owns the X,Y quads, O the Z quads.
// Layout (FABGPU_QTAB_SIG_MAJOR, the default): per SIGNATURE contiguous - entry j of signature k is the 128-byte line
// slot + (k * 16 + j - 1) * 128: the gather of a window (both lanes of the pair, eight 16-byte cells) is exactly one cache line, fully
// used.  The round-1 layout [entry][q][pair] made the STORES of a wave contiguous but scattered a wave's gather over ~27 lines of which
// 16 bytes each were wanted: rocprofv3 FETCH_SIZE 418 MB per 30 000-tuple launch for 266 MB of useful bytes (profiles/r01_pmc_traffic.json).
#ifndef FABGPU_QTAB_SIG_MAJOR
#define FABGPU_QTAB_SIG_MAJOR 1
#endif
template <int NP>
struct PairQTab {
    uint4* pair;   // first 16-byte cell of this signature's table
    // slot: workspace of this workgroup slot (NP signatures x 16 entries x 8 cells); k: index of the signature inside the workgroup
    static __device__ __forceinline__ PairQTab of(uint4* slot, uint32_t k) {
#if FABGPU_QTAB_SIG_MAJOR
        return PairQTab{slot + (size_t)k * (16 * 8)};
#else
        return PairQTab{slot + k};
#endif
    }
#if FABGPU_QTAB_SIG_MAJOR
    __device__ __forceinline__ uint4* cell(int j, int q) const { return pair + ((size_t)(j - 1) * 8 + q); }
#else
    __device__ __forceinline__ uint4* cell(int j, int q) const { return pair + ((size_t)(j - 1) * 8 + q) * NP; }
#endif
    __device__ __forceinline__ void store_state(int j, const pair_pt& p, bool odd) const {
        if (!odd) {
            *cell(j, 0) = make_uint4(p.A.v[0], p.A.v[1], p.A.v[2], p.A.v[3]);
            *cell(j, 1) = make_uint4(p.A.v[4], p.A.v[5], p.A.v[6], p.A.v[7]);
            *cell(j, 2) = make_uint4(p.A.v[8], p.B.v[0], p.B.v[1], p.B.v[2]);
            *cell(j, 3) = make_uint4(p.B.v[3], p.B.v[4], p.B.v[5], p.B.v[6]);
            *cell(j, 4) = make_uint4(p.B.v[7], p.B.v[8], 0, 0);
        } else {
            *cell(j, 5) = make_uint4(p.B.v[0], p.B.v[1], p.B.v[2], p.B.v[3]);
            *cell(j, 6) = make_uint4(p.B.v[4], p.B.v[5], p.B.v[6], p.B.v[7]);
            *cell(j, 7) = make_uint4(p.B.v[8], 0, 0, 0);
        }
    }
    // five 16-byte loads, the same code on both lanes: E reads q = q0 + k clamped to 7, O reads q = k
    __device__ __forceinline__ void load5(uint32_t j, int q0, uint4 (&l)[5]) const {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int q = q0 + k;
            l[k] = *cell((int)j, q > 7 ? 7 : q);
        }
    }
    // state layout: E gets (X, Y), O gets B = Z
    __device__ __forceinline__ void load_state(uint32_t j, pair_pt& p, bool odd) const {
        uint4 l[5];
        load5(j, odd ? 5 : 0, l);
        p.A.v[0] = l[0].x; p.A.v[1] = l[0].y; p.A.v[2] = l[0].z; p.A.v[3] = l[0].w;
        p.A.v[4] = l[1].x; p.A.v[5] = l[1].y; p.A.v[6] = l[1].z; p.A.v[7] = l[1].w;
        p.A.v[8] = l[2].x;
        fe y;
        y.v[0] = l[2].y; y.v[1] = l[2].z; y.v[2] = l[2].w;
        y.v[3] = l[3].x; y.v[4] = l[3].y; y.v[5] = l[3].z; y.v[6] = l[3].w;
        y.v[7] = l[4].x; y.v[8] = l[4].y;
        fe_sel(p.B, odd, p.A, y);       // O: the first nine words it read are Z
    }
    // crossed layout for PAIR_ADD: E gets C = Z2, O gets C = X2, D = Y2 (E's D is garbage)
    __device__ __forceinline__ void load_crossed(uint32_t j, fe& C, fe& D, bool odd) const {
        uint4 l[5];
        load5(j, odd ? 0 : 5, l);
        C.v[0] = l[0].x; C.v[1] = l[0].y; C.v[2] = l[0].z; C.v[3] = l[0].w;
        C.v[4] = l[1].x; C.v[5] = l[1].y; C.v[6] = l[1].z; C.v[7] = l[1].w;
        C.v[8] = l[2].x;
        D.v[0] = l[2].y; D.v[1] = l[2].z; D.v[2] = l[2].w;
        D.v[3] = l[3].x; D.v[4] = l[3].y; D.v[5] = l[3].z; D.v[6] = l[3].w;
        D.v[7] = l[4].x; D.v[8] = l[4].y;
    }
};

// The same table in LDS, EIGHT entries per signature (j*Q, j = 1..8, for signed 4-bit windows): 128 signatures x 8 x 128 B = 128 KiB of
// the CU's 160 KiB - the pair kernel runs one workgroup per CU anyway.  Costs 13 more additions per verification than the 16-entry
// table with 5-bit windows (65 windows instead of 52) and saves eight table-building operations; in exchange NOTHING of the
// per-signature table touches the memory system: the global-workspace form writes 60 MB and fetches 2 x 110 MB per 30 000-tuple
// launch (58x the algorithmic bytes), which is free on an idle chip and costs 4-17 % beside a tenant that streams 1.5-5 TB/s
// (tools/gpu_hbm_tenant_probe.py).  A signature's entries are 65 cells apart (one cell of padding per signature): a wave's 32 pairs
// gather from 32 different signatures, and a 1 KiB stride would put them all on the same banks.
constexpr int PAIR_LDS_ENTRIES = 8;
constexpr int PAIR_LDS_CELLS_PER_SIG = PAIR_LDS_ENTRIES * 8 + 1;
struct PairQTabLds {
    uint4* pair;   // (LDS) first cell of this signature's table
    static __device__ __forceinline__ PairQTabLds of(uint4* lds, uint32_t k) { return PairQTabLds{lds + (size_t)k * PAIR_LDS_CELLS_PER_SIG}; }
    __device__ __forceinline__ uint4* cell(int j, int q) const { return pair + ((j - 1) * 8 + q); }
    __device__ __forceinline__ void store_state(int j, const pair_pt& p, bool odd) const {
        if (!odd) {
            *cell(j, 0) = make_uint4(p.A.v[0], p.A.v[1], p.A.v[2], p.A.v[3]);
            *cell(j, 1) = make_uint4(p.A.v[4], p.A.v[5], p.A.v[6], p.A.v[7]);
            *cell(j, 2) = make_uint4(p.A.v[8], p.B.v[0], p.B.v[1], p.B.v[2]);
            *cell(j, 3) = make_uint4(p.B.v[3], p.B.v[4], p.B.v[5], p.B.v[6]);
            *cell(j, 4) = make_uint4(p.B.v[7], p.B.v[8], 0, 0);
        } else {
            *cell(j, 5) = make_uint4(p.B.v[0], p.B.v[1], p.B.v[2], p.B.v[3]);
            *cell(j, 6) = make_uint4(p.B.v[4], p.B.v[5], p.B.v[6], p.B.v[7]);
            *cell(j, 7) = make_uint4(p.B.v[8], 0, 0, 0);
        }
        // the partner lane reads what this lane wrote (and the other way round): same wavefront, in-order LDS - only the compiler
        // must not move accesses across
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ void load5(uint32_t j, int q0, uint4 (&l)[5]) const {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int q = q0 + k;
            l[k] = *cell((int)j, q > 7 ? 7 : q);
        }
    }
    __device__ __forceinline__ void load_state(uint32_t j, pair_pt& p, bool odd) const {
        uint4 l[5];
        load5(j, odd ? 5 : 0, l);
        p.A.v[0] = l[0].x; p.A.v[1] = l[0].y; p.A.v[2] = l[0].z; p.A.v[3] = l[0].w;
        p.A.v[4] = l[1].x; p.A.v[5] = l[1].y; p.A.v[6] = l[1].z; p.A.v[7] = l[1].w;
        p.A.v[8] = l[2].x;
        fe y;
        y.v[0] = l[2].y; y.v[1] = l[2].z; y.v[2] = l[2].w;
        y.v[3] = l[3].x; y.v[4] = l[3].y; y.v[5] = l[3].z; y.v[6] = l[3].w;
        y.v[7] = l[4].x; y.v[8] = l[4].y;
        fe_sel(p.B, odd, p.A, y);
    }
    __device__ __forceinline__ void load_crossed(uint32_t j, fe& C, fe& D, bool odd) const {
        uint4 l[5];
        load5(j, odd ? 0 : 5, l);
        C.v[0] = l[0].x; C.v[1] = l[0].y; C.v[2] = l[0].z; C.v[3] = l[0].w;
        C.v[4] = l[1].x; C.v[5] = l[1].y; C.v[6] = l[1].z; C.v[7] = l[1].w;
        C.v[8] = l[2].x;
        D.v[0] = l[2].y; D.v[1] = l[2].z; D.v[2] = l[2].w;
        D.v[3] = l[3].x; D.v[4] = l[3].y; D.v[5] = l[3].z; D.v[6] = l[3].w;
        D.v[7] = l[4].x; D.v[8] = l[4].y;
    }
};

// E gets x2, O gets y2 of comb entry (window, digit) in the SAME nine registers (they are passed as both C and D of PAIR_MADD)
template <class Tab>
__device__ __forceinline__ void pair_comb_load(const int32_t* __restrict__ tab, int window, uint32_t digit, bool odd, fe& xy) {
    const int32_t* e = tab + Tab::index(window, digit) + (odd ? 9 : 0);
#pragma unroll
    for (int l = 0; l < 9; l++) xy.v[l] = e[l];
}

// w = x^-1 mod M on a lane pair (modinv30.h).  The 30 division steps of a batch are computed by both lanes (round 6: each with one
// column of the transition matrix); then the EVEN
// lane applies the transition matrix to (f, g) and the ODD lane to (d, e) - one update per lane per batch instead of two.
// Both run the (d, e) update code: for (f, g) the modulus correction is forced to zero (t (f, g) is divisible by 2^30 by
// construction, so the same shift is exact).  The result is returned on both lanes.
__device__ __forceinline__ void pair_modinv(u256& out, const u256& x, const modinv_info& mi, bool odd) {
    s30 a, b;                      // E: (f, g)      O: (d, e)
    s30 gx;
    s30_from_u256(gx, x);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        a.v[i] = odd ? 0 : mi.modulus.v[i];
        b.v[i] = odd ? (i == 0 ? 1 : 0) : gx.v[i];
    }
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        if (it >= 17) {      // (modinv30.h modinv: random inputs are done after 17 or 18 batches; g lives on the even lanes)
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) nz |= (uint32_t)b.v[i];
            if (!__any(!odd && nz != 0)) break;
        }
        int32_t fo = pair_swap_i32(a.v[0]), go = pair_swap_i32(b.v[0]);
        uint32_t f0 = (uint32_t)(odd ? fo : a.v[0]), g0 = (uint32_t)(odd ? go : b.v[0]);
        // the division steps with one column of the transition matrix per lane - E: (u, q) from (1, 0), O: (v, r) from (0, 1) - then exchanged
        int32_t ca0 = odd ? 0 : 1, cb0 = odd ? 1 : 0;
        zeta = modinv_divsteps30_column(zeta, f0, g0, ca0, cb0);
        const int32_t pa = pair_swap_i32(ca0), pb = pair_swap_i32(cb0);
        trans2x2 t;
        t.u = odd ? pa : ca0;
        t.q = odd ? pb : cb0;
        t.v = odd ? ca0 : pa;
        t.r = odd ? cb0 : pb;
        // modinv_update_de with the correction masked off on the even lane
        int32_t sa = a.v[8] >> 31, sb = b.v[8] >> 31;
        int32_t ma = (t.u & sa) + (t.v & sb);
        int32_t mb = (t.q & sa) + (t.r & sb);
        int64_t ca = (int64_t)t.u * a.v[0] + (int64_t)t.v * b.v[0];
        int64_t cb = (int64_t)t.q * a.v[0] + (int64_t)t.r * b.v[0];
        ma -= (int32_t)((mi.inv30 * (uint32_t)ca + (uint32_t)ma) & (uint32_t)MI_M30);
        mb -= (int32_t)((mi.inv30 * (uint32_t)cb + (uint32_t)mb) & (uint32_t)MI_M30);
        ma = odd ? ma : 0;
        mb = odd ? mb : 0;
        ca += (int64_t)mi.modulus.v[0] * ma;
        cb += (int64_t)mi.modulus.v[0] * mb;
        ca >>= 30;
        cb >>= 30;
#pragma unroll
        for (int i = 1; i < 9; i++) {
            ca += (int64_t)t.u * a.v[i] + (int64_t)t.v * b.v[i] + (int64_t)mi.modulus.v[i] * ma;
            cb += (int64_t)t.q * a.v[i] + (int64_t)t.r * b.v[i] + (int64_t)mi.modulus.v[i] * mb;
            a.v[i - 1] = (int32_t)((uint32_t)ca & (uint32_t)MI_M30);
            b.v[i - 1] = (int32_t)((uint32_t)cb & (uint32_t)MI_M30);
            ca >>= 30;
            cb >>= 30;
        }
        a.v[8] = (int32_t)ca;
        b.v[8] = (int32_t)cb;
    }
    // E: f = +-1 in a;  O: d in a.  The inverse is d * sign(f): normalise on O, then hand the result to E.
    int32_t fsign_e = a.v[8] >> 31;
    int32_t fsign = pair_swap_i32(fsign_e);       // O: sign of f
    modinv_normalize(a, odd ? fsign : 0, mi);
    u256 w;
    s30_to_u256(w, a);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int32_t o = pair_swap_i32((int32_t)w.w[i]);
        out.w[i] = odd ? w.w[i] : (uint32_t)o;
    }
}

// ecdsa_scalars29 with the inversion shared by the pair
__device__ __forceinline__ void pair_ecdsa_scalars29(u256& u1, u256& u2, const u256& e, const u256& r, const u256& s, bool odd) {
    const u256 N = FAB_P256_N;
    u256 w, ered, t;
    {
        const modinv_info NI = MODINV_N_INFO;
        pair_modinv(w, s, NI, odd);
    }
    uint32_t br = sub256(t, e, N);
    sel256(ered, br == 0, t, e);
    fn_to_mont(t, ered);
    fn_mul(u1, t, w);
    fn_to_mont(t, r);
    fn_mul(u2, t, w);
}

// S = k * B over a comb table of B on a lane pair (comb_mult29).  seed: any valid point in pair state.
template <class Tab>
__device__ __forceinline__ void pair_comb_mult29(pair_pt& S, bool& s_inf, const u256& k, const int32_t* __restrict__ tab, const pair_pt& seed,
                                                 bool odd) {
    const fe ONE = {FE29_R1};
    PAIR_TMPS;
    S = seed;
    s_inf = true;
    uint32_t nd = Tab::digit(k, 0);
    fe nxy;
    pair_comb_load<Tab>(tab, 0, nd ? nd : 1u, odd, nxy);
#pragma unroll 1
    for (int i = 0; i < Tab::WINDOWS; i++) {
        uint32_t d = nd;
        fe xy = nxy;
        int inext = i + 1 < Tab::WINDOWS ? i + 1 : i;
        nd = Tab::digit(k, inext);
        pair_comb_load<Tab>(tab, inext, nd ? nd : 1u, odd, nxy);
        pair_pt sum;
        PAIR_MADD(sum, S, xy, xy);
        bool take_ent = s_inf & (d != 0);
        bool take_sum = (!s_inf) & (d != 0);
        pair_sel(S, take_sum, sum, S);
        if (__any(take_ent)) {
            pair_pt ent;
            fe sy;
            pair_swap_fe(sy, xy);           // E: y2
            ent.A = xy;                     // E: x2
            fe_sel(ent.B, odd, ONE, sy);
            pair_sel(S, take_ent, ent, S);
        }
        s_inf = s_inf & (d == 0);
    }
}

// R = S + T on a lane pair with the exceptional cases of the group law (final_add29).
__device__ __forceinline__ void pair_final_add29(pair_pt& Rr, bool& r_inf, const pair_pt& S, bool s_inf, const pair_pt& T, bool t_inf, bool odd) {
    PAIR_TMPS;
    fe C, D, sa, sb;
    pair_swap_fe(sa, T.A);                  // O: X_T
    pair_swap_fe(sb, T.B);                  // E: Z_T    O: Y_T
    fe_sel(C, odd, sa, sb);
    D = sb;
    pair_pt Rp;
    PAIR_ADD(Rp, S, C, D);
    bool hz = fe_is_zero(tH);               // E: h, O: -h
    bool rz_own = fe_is_zero(tRR);          // rr lives on E
    int32_t rz_other = pair_swap_i32(rz_own ? 1 : 0);
    bool rz = odd ? (rz_other != 0) : rz_own;
    r_inf = t_inf & s_inf;
    bool use_T = s_inf & !t_inf;
    bool use_S = t_inf & !s_inf;
    bool both = !s_inf & !t_inf;
    bool use_dbl = both & hz & rz;
    r_inf = r_inf | (both & hz & !rz);
    Rr = Rp;
    if (__any(use_dbl)) {                   // S == T: only a crafted signature gets here - no wavefront of honest ones pays for the doubling
        pair_pt Rd = T;
        PAIR_DBL(Rd);
        pair_sel(Rr, use_dbl, Rd, Rr);
    }
    pair_sel(Rr, use_T, T, Rr);
    pair_sel(Rr, use_S, S, Rr);
}

// R = u1*G + u2*Q on a lane pair.  Q: affine Montgomery (both lanes hold both coordinates).  Returns the pair state of R;
// r_inf as in p256_combined_mult29.  W = width of the signed (Booth) windows over u2: 5 (a 16-entry table j*Q, 52 windows: the
// global-workspace table) or 4 (8 entries, 65 windows: the LDS table).
//
// No addition inside the loop may meet P == +-Q (the addition formulas do not handle it).  Before window i is added T = M Q with M a
// non-zero multiple of 2^W and the addend is d Q, |d| <= 2^(W-1).  M == +-d (mod n) needs M = n +- |d| (M >= 2^W > |d|), possible
// only at the last window, where M = u2 - d: M = n - d gives u2 = n, impossible; M = n + d with d < 0 gives u2 = n - 2|d| and needs
// |d| == n (mod 2^W).  n mod 32 = 17 > 16: never for W = 5.  n mod 16 = 1: for W = 4 EXACTLY ONE scalar, u2 = n - 2
// (digit -1 on top of M = n - 1: T = -Q, addend -Q).  That scalar is recognised up front and its product, -2Q, is taken from the
// table (entry 2, negated) instead of from the loop.
template <class QTab, int W = 5>
__device__ __forceinline__ void pair_combined_mult29(pair_pt& Rr, bool& r_inf, const u256& u1, const u256& u2, const fe& QX, const fe& QY,
                                                     const int32_t* __restrict__ gtab, const QTab& qtab, bool odd) {
    static_assert(W == 4 || W == 5, "signed 4- or 5-bit windows");
    constexpr int TAB = 1 << (W - 1);                 // entries j*Q, j = 1..TAB
    constexpr int NWIN = (257 + W - 1) / W;           // windows over bits -1 .. 256 of a scalar below n
    const fe ONE = {FE29_R1};
    PAIR_TMPS;
    pair_pt Qp;
    Qp.A = QX;
    fe_sel(Qp.B, odd, ONE, QY);

    // --- per-signature table j*Q, j = 1..TAB ---
    qtab.store_state(1, Qp, odd);
#pragma unroll 1
    for (int j = 2; j <= TAB; j += 2) {
        pair_pt d;
        qtab.load_state((uint32_t)(j >> 1), d, odd);
        PAIR_DBL(d);
        qtab.store_state(j, d, odd);
        if (j < TAB) {
            pair_pt d1;
            PAIR_MADD(d1, d, QX, QY);
            qtab.store_state(j + 1, d1, odd);
        }
    }

    // --- T = u2 * Q : NWIN signed W-bit windows (same recoding as p256_combined_mult29 for W = 5) ---
    uint32_t kw[9];
#pragma unroll
    for (int i = 0; i < 8; i++) kw[i] = u2.w[i];
    kw[8] = 0;
    pair_pt T = Qp;
    bool t_inf = true;
#pragma unroll 1
    for (int i = NWIN - 1; i >= 0; i--) {
        uint32_t field;                               // bits W i - 1 .. W i + W - 1 of the scalar (bit -1 = 0)
        if (i == 0) {
            field = (kw[0] << 1) & ((2u << W) - 1u);
        } else {
            int p = W * i - 1;
            uint64_t two = ((uint64_t)kw[(p >> 5) + 1] << 32) | kw[p >> 5];
            field = (uint32_t)(two >> (p & 31)) & ((2u << W) - 1u);
        }
        int32_t digit = (int32_t)((field >> 1) & (uint32_t)(TAB - 1)) + (int32_t)(field & 1u) - (int32_t)((field >> W) << (W - 1));
        bool neg = digit < 0;
        uint32_t mag = (uint32_t)(neg ? -digit : digit);
        fe C, D;
        qtab.load_crossed(mag ? mag : 1u, C, D, odd);   // issued ahead of the doublings
        if (i != NWIN - 1) {
#pragma unroll 1
            for (int k = 0; k < W; k++) PAIR_DBL(T);
        }
        {
            const int32_t nm = neg ? -1 : 0, nc = neg ? 1 : 0;        // -y = (y ^ -1) + 1: one v_xad_u32 per limb instead of a negation and a select
#pragma unroll
            for (int l = 0; l < 9; l++) D.v[l] = (D.v[l] ^ nm) + nc;  // -Y2 (lives on O)
        }
        pair_pt sum;
        PAIR_ADD(sum, T, C, D);
        bool take_ent = t_inf & (mag != 0);
        bool take_sum = (!t_inf) & (mag != 0);
        pair_sel(T, take_sum, sum, T);
        if (__any(take_ent)) {   // wave-uniform: only the first non-zero window(s) of a wave convert the entry to state layout
            pair_pt ent;
            fe sc, sd;
            pair_swap_fe(sc, C);            // E: X2     O: Z2
            pair_swap_fe(sd, D);            // E: Y2
            ent.A = sc;
            fe_sel(ent.B, odd, sc, sd);
            pair_sel(T, take_ent, ent, T);
        }
        t_inf = t_inf & (mag == 0);
    }
    if (W == 4) {
        // u2 = n - 2: the one scalar whose last addition is a doubling (see above).  (n - 2) Q = -2Q: entry 2, Y negated.
        const u256 NM2 = {{0xFC63254Fu, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu}};
        bool special = true;
#pragma unroll
        for (int l = 0; l < 8; l++) special = special & (u2.w[l] == NM2.w[l]);
        if (__any(special)) {
            pair_pt two;
            qtab.load_state(2u, two, odd);
            fe ny;
#pragma unroll
            for (int l = 0; l < 9; l++) ny.v[l] = -two.B.v[l];
            fe_sel(two.B, odd, two.B, ny);        // E holds Y in B (negate it), O holds Z in B (keep it)
            pair_sel(T, special, two, T);
        }
    }

    // --- S = u1 * G (16-bit comb), then R = S + T ---
    pair_pt S;
    bool s_inf;
    pair_comb_mult29<GTab16>(S, s_inf, u1, gtab, Qp, odd);
    pair_final_add29(Rr, r_inf, S, s_inf, T, t_inf, odd);
}

// R = u1*G + u2*Q with both points on comb tables (registered key): 16 + 32 pair mixed additions.
template <class KTab = KeyTab8>
__device__ __forceinline__ void pair_combined_mult_keyed29(pair_pt& Rr, bool& r_inf, const u256& u1, const u256& u2,
                                                           const int32_t* __restrict__ gtab, const int32_t* __restrict__ ktab, bool odd) {
    const fe ONE = {FE29_R1};
    fe gx, gy;
    GTab16 gt{gtab};
    gt.load(0, 1u, gx, gy);
    pair_pt seed, S, T;
    seed.A = gx;
    fe_sel(seed.B, odd, ONE, gy);
    bool s_inf, t_inf;
    pair_comb_mult29<KTab>(T, t_inf, u2, ktab, seed, odd);
    pair_comb_mult29<GTab16>(S, s_inf, u1, gtab, seed, odd);
    pair_final_add29(Rr, r_inf, S, s_inf, T, t_inf, odd);
}

// x(R) mod n == r on a pair: Z^2 and r Z^2 on O, the comparison with X on E.  The result is valid on the EVEN lane.
// NB the swapped operand is the MINUEND: hipcc folds the DPP move into the subtraction, and for "own - partner" it emits
// v_subrev_u32_dpp, which on MI355X does not compute src1 - dpp(src0) (probed in gputest.hip op 3; the Makefile rejects
// any build whose device code contains that opcode).
__device__ __forceinline__ bool pair_x_equals_r29(const pair_pt& Rr, bool r_inf, const u256& r) {
    const u256 N = FAB_P256_N;
    const u256 PMN = FAB_P256_P_MINUS_N;
    fe zz, rm, rhs, rhs_e, d;
    u256 r2;
    fe_sqr(zz, Rr.B);                                  // O: Z^2
    fe_to_mont(rm, r);
    fe_mul(rhs, rm, zz);
    pair_swap_fe(rhs_e, rhs);                          // E: r Z^2
    fe_sub(d, rhs_e, Rr.A);
    bool ok = fe_is_zero(d);
    add256(r2, r, N);
    fe_to_mont(rm, r2);
    fe_mul(rhs, rm, zz);
    pair_swap_fe(rhs_e, rhs);
    fe_sub(d, rhs_e, Rr.A);
    ok = ok | (lt256(r, PMN) & fe_is_zero(d));
    return ok & !r_inf;
}

// Status of one tuple, valid on the EVEN lane of the pair.
template <class QTab, int W = 5>
__device__ __forceinline__ uint32_t p256_verify_pair29(const u256& qx, const u256& qy, const u256& e, const u256& r, const u256& s,
                                                        const int32_t* __restrict__ gtab, const QTab& qtab, bool odd) {
    const u256 P = FAB_P256_P;
    uint32_t early = range_status(r, s);

    bool q_in_field = lt256(qx, P) & lt256(qy, P);
    fe QX, QY;
    fe_to_mont(QX, qx);
    fe_to_mont(QY, qy);
    bool q_ok = q_in_field & on_curve29(QX, QY);
    if (early == ST_VALID && !q_ok) early = ST_OFF_CURVE;

    u256 u1, u2;
    pair_ecdsa_scalars29(u1, u2, e, r, s, odd);

    pair_pt Rr;
    bool r_inf;
    pair_combined_mult29<QTab, W>(Rr, r_inf, u1, u2, QX, QY, gtab, qtab, odd);
    bool ok = pair_x_equals_r29(Rr, r_inf, r);
    uint32_t st = ok ? ST_VALID : ST_BAD_MATH;
    return early != ST_VALID ? early : st;
}

// Registered key (p256_verify_keyed_core29): status valid on the EVEN lane.
// ktab16: the key's 16-bit comb (round 6: FABGPU_FLAG_KEY_TABLES_16BIT; nullptr while the key has none) - a wavefront all of whose keys
// have one does 16 mixed additions for u2*Q instead of 32.
__device__ __forceinline__ uint32_t p256_verify_keyed_pair29(const u256& e, const u256& r, const u256& s, const int32_t* __restrict__ gtab,
                                                              const int32_t* __restrict__ ktab, const int32_t* __restrict__ ktab16, bool odd) {
    uint32_t early = range_status(r, s);
    u256 u1, u2;
    pair_ecdsa_scalars29(u1, u2, e, r, s, odd);
    pair_pt Rr;
    bool r_inf;
    if (__all(ktab16 != nullptr)) pair_combined_mult_keyed29<GTab16>(Rr, r_inf, u1, u2, gtab, ktab16, odd);
    else pair_combined_mult_keyed29<KeyTab8>(Rr, r_inf, u1, u2, gtab, ktab, odd);
    bool ok = pair_x_equals_r29(Rr, r_inf, r);
    uint32_t st = ok ? ST_VALID : ST_BAD_MATH;
    return early != ST_VALID ? early : st;
}

}  // namespace fab
