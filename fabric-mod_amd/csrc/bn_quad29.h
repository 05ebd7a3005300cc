// Idemix pseudonym-signature verification with FOUR LANES PER SIGNATURE (device only) - the small-batch form of bn_nym29.h.
//
// The idemix creators of one block are a few thousand signatures: 6 000 of them are 94 wavefronts in the one-lane kernel and 188 in
// the two-lane one on 1 024 SIMDs, and a wave issues one VALU instruction per ~4.3 cycles whatever it is - the kernel time is the
// LENGTH of a wave's instruction stream.  Here lanes 4k .. 4k+3 share signature k:
//     lanes 4k, 4k+1 (pair 0):  HSk   * s_sk    -  k1 * Nym                 (c = k1 + k2 lambda, bn_glv_decompose)
//     lanes 4k+2, 4k+3 (pair 1): HRand * s_rnym  -  k2 * phi(Nym)
// - the same split of the equation as bn_nym_split_part1 - and INSIDE a pair every point operation is one of the generated
// two-lanes-per-point programs of pair29_bn_gcn.h (PAIRBN_DBL 939 / PAIRBN_ADD 1796 / PAIRBN_MADD 1428 instructions against
// ~1 500 / ~3 500 / ~2 600 of the one-lane formulas), exactly as p256_pair29.h does it for P-256: E (even lane) holds A = X, B = Y,
// O (odd lane) holds B = Z between operations; limbs cross lanes with DPP quad_perm:[1,0,3,2].  The two pairs exchange their partial
// sums with quad_perm:[2,3,0,1] (E meets E, O meets O: the pair layout survives the move) and both finish t = (pair 0) + (pair 1);
// the inversion is the pair form of safegcd (p256_pair29.h pair_modinv); lane 4k hashes and reports.
//
// Same gates, same window recodings, same tables and the same statuses as bn_nym_commitment29 / bn_nym_split_part1, which it must agree
// with bit for bit: tests/test_idemix_gpu.py runs every case through all three kernels against the oracle.
// The programs are the DSL's (gen_pair_gcn.py bnpair): verified in the interpreter against big integers as a double-and-add chain
// (tests/test_pair_programs.py) and register for register on the MI355X (gputest ops 4-6).
//
// Replaces idemix/nymsignature.go:74-109 (NymSignature.Ver).
#pragma once
#include "bn_nym29.h"
#include "p256_pair29.h"      // pair_swap_i32, pair_modinv (generic in the modulus)
#include "pair29_bn_gcn.h"

namespace fab {

struct pairbn_pt {
    fbn A, B;   // E: X, Y      O: don't-care, Z
};

__device__ __forceinline__ void pairbn_swap_fe(fbn& r, const fbn& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = pair_swap_i32(a.v[i]);
}
// the other PAIR of the quad (quad_perm:[2,3,0,1]): lane 4k <-> 4k+2, 4k+1 <-> 4k+3
__device__ __forceinline__ int32_t quad_swap_i32(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false); }
__device__ __forceinline__ void quad_swap_fe(fbn& r, const fbn& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = quad_swap_i32(a.v[i]);
}
__device__ __forceinline__ void pairbn_sel(pairbn_pt& r, bool c, const pairbn_pt& a, const pairbn_pt& b) {
    fe_sel(r.A, c, a.A, b.A);
    fe_sel(r.B, c, a.B, b.B);
}

#define QB_TMPS __attribute__((unused)) fbn tU1, tU2, tU3, tU4, tU6, tW, tH, tRR, tP1, tP2, tT0, tT1, tTD
#define QB_DBL(P) PAIRBN_DBL((P).A, (P).B, tU1, tU2, tU3, tU4, tP1, tP2, tT0, tT1, tTD)
#define QB_ADD(R, P, C, D) PAIRBN_ADD((R).A, (R).B, (P).A, (P).B, tH, tRR, tW, tU1, tU2, tU3, tU4, tU6, tP1, tP2, tT0, tT1, tTD, C, D)
#define QB_MADD(R, P, C, D) PAIRBN_MADD((R).A, (R).B, (P).A, (P).B, tU1, tU2, tU3, tU4, tH, tRR, tP1, tP2, tT0, tT1, tTD, C, D)

// Per-PAIR table j*Q (j = 1..16) in the global workspace: entry j of pair k is the 128-byte line slot + (k * 16 + j - 1) * 128 - the
// layout of PairQTab (p256_pair29.h): cells 0..4 = X[9] Y[9] pad (E stores them), cells 5..7 = Z[9] pad (O).
struct PairBNQTab {
    uint4* pair;
    static __device__ __forceinline__ PairBNQTab of(uint4* slot, uint32_t k) { return PairBNQTab{slot + (size_t)k * (16 * 8)}; }
    __device__ __forceinline__ uint4* cell(int j, int q) const { return pair + ((size_t)(j - 1) * 8 + q); }
    __device__ __forceinline__ void store_state(int j, const pairbn_pt& p, bool odd) const {
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
    __device__ __forceinline__ void load5(uint32_t j, int q0, uint4 (&l)[5]) const {
#pragma unroll
        for (int k = 0; k < 5; k++) {
            int q = q0 + k;
            l[k] = *cell((int)j, q > 7 ? 7 : q);
        }
    }
    // state layout: E gets (X, Y), O gets B = Z
    __device__ __forceinline__ void load_state(uint32_t j, pairbn_pt& p, bool odd) const {
        uint4 l[5];
        load5(j, odd ? 5 : 0, l);
        p.A.v[0] = l[0].x; p.A.v[1] = l[0].y; p.A.v[2] = l[0].z; p.A.v[3] = l[0].w;
        p.A.v[4] = l[1].x; p.A.v[5] = l[1].y; p.A.v[6] = l[1].z; p.A.v[7] = l[1].w;
        p.A.v[8] = l[2].x;
        fbn y;
        y.v[0] = l[2].y; y.v[1] = l[2].z; y.v[2] = l[2].w;
        y.v[3] = l[3].x; y.v[4] = l[3].y; y.v[5] = l[3].z; y.v[6] = l[3].w;
        y.v[7] = l[4].x; y.v[8] = l[4].y;
        fe_sel(p.B, odd, p.A, y);       // O: the first nine words it read are Z
    }
    // crossed layout for QB_ADD: E gets C = Z2, O gets C = X2, D = Y2 (E's D is garbage)
    __device__ __forceinline__ void load_crossed(uint32_t j, fbn& C, fbn& D, bool odd) const {
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

// E gets x2, O gets y2 of comb entry (window, digit) in the SAME nine registers (they are passed as both C and D of QB_MADD)
__device__ __forceinline__ void pairbn_comb_load(const int32_t* __restrict__ tab, int window, uint32_t digit, bool odd, fbn& xy) {
    const int32_t* e = tab + KeyTab8::index(window, digit) + (odd ? 9 : 0);
#pragma unroll
    for (int l = 0; l < 9; l++) xy.v[l] = e[l];
}

// S = k * B over the 8-bit comb table of B on a lane pair (comb_mult29 / pair_comb_mult29).  seed: any valid point in pair state.
__device__ __forceinline__ void pairbn_comb_mult(pairbn_pt& S, bool& s_inf, const u256& k, const int32_t* __restrict__ tab, const pairbn_pt& seed,
                                                 bool odd) {
    fbn ONE;
    fe_set_one(ONE);
    QB_TMPS;
    S = seed;
    s_inf = true;
    uint32_t nd = KeyTab8::digit(k, 0);
    fbn nxy;
    pairbn_comb_load(tab, 0, nd ? nd : 1u, odd, nxy);
#pragma unroll 1
    for (int i = 0; i < KeyTab8::WINDOWS; i++) {
        uint32_t d = nd;
        fbn xy = nxy;
        int inext = i + 1 < KeyTab8::WINDOWS ? i + 1 : i;
        nd = KeyTab8::digit(k, inext);
        pairbn_comb_load(tab, inext, nd ? nd : 1u, odd, nxy);
        pairbn_pt sum;
        QB_MADD(sum, S, xy, xy);
        bool take_ent = s_inf & (d != 0);
        bool take_sum = (!s_inf) & (d != 0);
        pairbn_sel(S, take_sum, sum, S);
        if (__any(take_ent)) {
            pairbn_pt ent;
            fbn sy;
            pairbn_swap_fe(sy, xy);         // E: y2
            ent.A = xy;                     // E: x2
            fe_sel(ent.B, odd, ONE, sy);
            pairbn_sel(S, take_ent, ent, S);
        }
        s_inf = s_inf & (d == 0);
    }
}

// R = S + T on a lane pair with the exceptional cases of the group law (final_add29 / pair_final_add29).
__device__ __forceinline__ void pairbn_final_add(pairbn_pt& Rr, bool& r_inf, const pairbn_pt& S, bool s_inf, const pairbn_pt& T, bool t_inf, bool odd) {
    QB_TMPS;
    fbn C, D, sa, sb;
    pairbn_swap_fe(sa, T.A);                // O: X_T
    pairbn_swap_fe(sb, T.B);                // E: Z_T    O: Y_T
    fe_sel(C, odd, sa, sb);
    D = sb;
    pairbn_pt Rp;
    QB_ADD(Rp, S, C, D);
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
    if (__any(use_dbl)) {                   // S == T: crafted inputs only
        pairbn_pt Rd = T;
        QB_DBL(Rd);
        pairbn_sel(Rr, use_dbl, Rd, Rr);
    }
    pairbn_sel(Rr, use_T, T, Rr);
    pairbn_sel(Rr, use_S, S, Rr);
}

// T = m * Q on a lane pair, m < 2^(5 WINDOWS - 1) below the group order (booth_mult29): the pair's table j*Q in the workspace (8 pair
// doublings + 7 pair mixed additions), then WINDOWS signed 5-bit windows.  QX, QY: affine Montgomery, on both lanes.
template <int WINDOWS>
__device__ __forceinline__ void pairbn_booth_mult(pairbn_pt& T, bool& t_inf, const u256& m, const fbn& QX, const fbn& QY, const PairBNQTab& qtab,
                                                  bool odd) {
    fbn ONE;
    fe_set_one(ONE);
    QB_TMPS;
    pairbn_pt Qp;
    Qp.A = QX;
    fe_sel(Qp.B, odd, ONE, QY);
    qtab.store_state(1, Qp, odd);
#pragma unroll 1
    for (int j = 2; j <= 16; j += 2) {
        pairbn_pt d;
        qtab.load_state((uint32_t)(j >> 1), d, odd);
        QB_DBL(d);
        qtab.store_state(j, d, odd);
        if (j < 16) {
            pairbn_pt d1;
            QB_MADD(d1, d, QX, QY);
            qtab.store_state(j + 1, d1, odd);
        }
    }
    uint32_t kw[9];
#pragma unroll
    for (int i = 0; i < 8; i++) kw[i] = m.w[i];
    kw[8] = 0;
    T = Qp;
    t_inf = true;
#pragma unroll 1
    for (int i = WINDOWS - 1; i >= 0; i--) {
        int32_t digit = booth5_digit(kw, i);
        bool neg = digit < 0;
        uint32_t mag = (uint32_t)(neg ? -digit : digit);
        fbn C, D;
        qtab.load_crossed(mag ? mag : 1u, C, D, odd);   // issued ahead of the doublings
        if (i != WINDOWS - 1) {
#pragma unroll 1
            for (int k = 0; k < 5; k++) QB_DBL(T);
        }
        {
            const int32_t nm = neg ? -1 : 0, nc = neg ? 1 : 0;        // -y = (y ^ -1) + 1: one v_xad_u32 per limb instead of a negation and a select
#pragma unroll
            for (int l = 0; l < 9; l++) D.v[l] = (D.v[l] ^ nm) + nc;  // -Y2 (lives on O)
        }
        pairbn_pt sum;
        QB_ADD(sum, T, C, D);
        bool take_ent = t_inf & (mag != 0);
        bool take_sum = (!t_inf) & (mag != 0);
        pairbn_sel(T, take_sum, sum, T);
        if (__any(take_ent)) {   // wave-uniform: only the first non-zero window(s) of a wave convert the entry to state layout
            pairbn_pt ent;
            fbn sc, sd;
            pairbn_swap_fe(sc, C);          // E: X2     O: Z2
            pairbn_swap_fe(sd, D);          // E: +-Y2
            ent.A = sc;
            fe_sel(ent.B, odd, sc, sd);
            pairbn_sel(T, take_ent, ent, T);
        }
        t_inf = t_inf & (mag == 0);
    }
}

// One pair's half of the commitment (bn_nym_split_part1 in pair form).  half: which pair of the quad this lane belongs to.
struct bn_nym_quad_half {
    pairbn_pt P;
    bool inf;
    uint32_t early;
    bool dom;
};
// This pair's fixed-base term: HSk * s_sk (pair 0) or HRand * s_rnym (pair 1) over the issuer's 8-bit comb tables - 32 mixed additions,
// no doublings, no dependence on the signature's pseudonym.  Also runs on its own as idemix_nym_comb_quad_kernel, beside the
// variable-base half (idemix_kernels.hip "three launches").
__device__ __forceinline__ void bn_nym_quad_comb(pairbn_pt& S, bool& s_inf, bool odd, bool half, const u256& s_sk, const u256& s_rnym,
                                                 const int32_t* __restrict__ hsk, const int32_t* __restrict__ hrand) {
    const int32_t* tab = half ? hrand : hsk;
    u256 sc;
    sel256(sc, half, s_rnym, s_sk);
    fbn ONE, gx, gy;
    fe_set_one(ONE);
    KeyTab8 t0{hsk};
    t0.load(0, 1u, gx, gy);
    pairbn_pt seed;
    seed.A = gx;
    fe_sel(seed.B, odd, ONE, gy);
    pairbn_comb_mult(S, s_inf, sc, tab, seed, odd);
}

// The comb term as it travels from the comb launch to the commitment launch: 20 words per lane (A, B, s_inf, pad) = five uint4.
constexpr int NYM_COMB_UINT4_PER_LANE = 5;
__device__ __forceinline__ void bn_nym_comb_store(uint4* __restrict__ dst, const pairbn_pt& S, bool s_inf) {
    dst[0] = make_uint4(S.A.v[0], S.A.v[1], S.A.v[2], S.A.v[3]);
    dst[1] = make_uint4(S.A.v[4], S.A.v[5], S.A.v[6], S.A.v[7]);
    dst[2] = make_uint4(S.A.v[8], S.B.v[0], S.B.v[1], S.B.v[2]);
    dst[3] = make_uint4(S.B.v[3], S.B.v[4], S.B.v[5], S.B.v[6]);
    dst[4] = make_uint4(S.B.v[7], S.B.v[8], s_inf ? 1u : 0u, 0u);
}
__device__ __forceinline__ void bn_nym_comb_load(const uint4* __restrict__ src, pairbn_pt& S, bool& s_inf) {
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3], e = src[4];
    S.A.v[0] = a.x; S.A.v[1] = a.y; S.A.v[2] = a.z; S.A.v[3] = a.w;
    S.A.v[4] = b.x; S.A.v[5] = b.y; S.A.v[6] = b.z; S.A.v[7] = b.w;
    S.A.v[8] = c.x; S.B.v[0] = c.y; S.B.v[1] = c.z; S.B.v[2] = c.w;
    S.B.v[3] = d.x; S.B.v[4] = d.y; S.B.v[5] = d.z; S.B.v[6] = d.w;
    S.B.v[7] = e.x; S.B.v[8] = e.y;
    s_inf = e.z != 0u;
}

// comb_ready / comb_in: the comb launch's flag for this wavefront's rows and this lane's record (idemix_kernels.hip), or nullptr.  The
// variable-base half runs first; THEN the comb term is taken from the comb launch if it has arrived (it runs beside this one and is a
// quarter as long) - and computed here if it has not: the result is the same either way, only the time differs.
__device__ __forceinline__ void bn_nym_quad_part1(bn_nym_quad_half& out, bool odd, bool half, const u256& nx, const u256& ny, const u256& c,
                                                  const u256& s_sk, const u256& s_rnym, const int32_t* __restrict__ hsk,
                                                  const int32_t* __restrict__ hrand, const PairBNQTab& qtab,
                                                  uint32_t* __restrict__ comb_ready = nullptr, const uint4* __restrict__ comb_in = nullptr) {
    jacbn N;
    bn_nym_gates29(out.early, out.dom, N, nx, ny, c, s_sk, s_rnym);
    pairbn_pt S, T;
    bool s_inf, t_inf;
    if (comb_ready == nullptr) bn_nym_quad_comb(S, s_inf, odd, half, s_sk, s_rnym, hsk, hrand);
    // this pair's half of c * Nym
    u256 m1, m2, m;
    bool n1, n2;
    bn_glv_decompose(m1, n1, m2, n2, c);
    sel256(m, half, m2, m1);
    bool neg = half ? n2 : n1;
    const fbn BETA = {BN29_BETA_MONT};
    fbn bx;
    fe_mul(bx, N.X, BETA);                       // [1x1]  phi(Nym) = (beta x, y)
    fe_sel(N.X, half, bx, N.X);
    pairbn_booth_mult<GLV_WINDOWS>(T, t_inf, m, N.X, N.Y, qtab, odd);
    // partial = S - (+-T): subtracting, so the Y of T (on E; O's B is Z) flips unless the half-scalar was negative
#pragma unroll
    for (int l = 0; l < 9; l++) T.B.v[l] = (neg | odd) ? T.B.v[l] : -T.B.v[l];
    if (comb_ready != nullptr) {
        // has the comb launch delivered this wavefront's rows?  (one flag per wavefront, written behind its 64 records with release
        // semantics; a few microseconds of patience, then the term is computed here)
        uint32_t have = 0;
#pragma unroll 1
        for (int spin = 0; spin < 24 && !have; spin++) {
            have = __hip_atomic_load(comb_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 1u ? 1u : 0u;
            have = __builtin_amdgcn_readfirstlane(have);
            if (!have) __builtin_amdgcn_s_sleep(64);
        }
        if (!have) {
            // giving up: say so (0 -> 2), so that a comb launch that has not started this wavefront's rows yet skips them instead of
            // making the caller's stream wait for work nobody will read; if the records arrived this very moment (the exchange finds 1)
            // they are taken after all
            uint32_t old = 0;
            if ((threadIdx.x & 63u) == 0) old = atomicCAS(comb_ready, 0u, 2u);
            have = __builtin_amdgcn_readfirstlane(old) == 1u ? 1u : 0u;
            if (have) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        if (have) bn_nym_comb_load(comb_in, S, s_inf);
        else bn_nym_quad_comb(S, s_inf, odd, half, s_sk, s_rnym, hsk, hrand);
    }
    pairbn_final_add(out.P, out.inf, S, s_inf, T, t_inf, odd);
}

// t = (this pair's partial sum) + (the other pair's), affine, plain integers in [0, p) - meaningful on the EVEN lanes.
__device__ __forceinline__ uint32_t bn_nym_quad_part2(u256& tx, u256& ty, const bn_nym_quad_half& mine, bool odd) {
    pairbn_pt theirs, W;
    quad_swap_fe(theirs.A, mine.P.A);
    quad_swap_fe(theirs.B, mine.P.B);
    bool theirs_inf = quad_swap_i32(mine.inf ? 1 : 0) != 0;
    bool w_inf;
    pairbn_final_add(W, w_inf, mine.P, mine.inf, theirs, theirs_inf, odd);
    // one inversion mod p for the pair: Z travels to E, both lanes run the shared safegcd
    fbn zsw, z;
    pairbn_swap_fe(zsw, W.B);                    // E: Z
    fe_sel(z, odd, W.B, zsw);
    u256 zp, zi;
    fe_from_mont(zp, z);
    {
        const modinv_info PI = MODINV_BNP_INFO;
        pair_modinv(zi, zp, PI, odd);
    }
    fbn zm, zi2, zi3, ax, ay;
    fe_to_mont(zm, zi);
    fe_sqr(zi2, zm);               // [1x1]
    fe_mul(zi3, zi2, zm);          // [1x1]
    fe_mul(ax, W.A, zi2);          // [1x1]   (E: X)
    fe_mul(ay, W.B, zi3);          // [3x1]   (E: Y)
    fe_from_mont(tx, ax);
    fe_from_mont(ty, ay);
    if (mine.early != NYM_VALID) return mine.early;
    if (!mine.dom || w_inf) return NYM_NEEDS_SW;
    return NYM_VALID;
}

}  // namespace fab
