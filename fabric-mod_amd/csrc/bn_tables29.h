// Host-side precomputation of the comb tables of an idemix issuer's bases HSk and HRand (ec29.h CombTab<BITS>, fbn elements):
// T[w][d] = d * 2^(BITS w) * B, affine, Montgomery form, balanced digits - built at fabgpu_idemix_issuer_register with the SAME
// field and point code the kernels run (bn29.h / bn_nym29.h compiled for the host).  The device uses BITS = 8 (640 KiB per base,
// L2-resident, 32 mixed additions per scalar).  BITS = 16 (80 MiB per base, 16 additions) is built and checked by the CPU tests
// and was MEASURED on the device (round 1): two issuers = 320 MiB of tables no longer fit the 256 MiB Infinity Cache, the random
// 80-byte gathers pay HBM and TLB latency that one window of prefetch does not hide, and the kernel got slower (60 000
// signatures 1.55 -> 1.87 ms, 6 000 on two lanes 1.08 -> 1.35 ms) although it executes 10 % fewer instructions.
#pragma once
#include <thread>
#include <vector>

#include "bn_nym29.h"

namespace fab {

// One window of a comb table: entries d = 1 .. 2^BITS - 1 of `base` = 2^(BITS w) B (Jacobian), one inversion for the window.
template <int BITS>
inline void build_bn_comb_window(int32_t* words, int w, const jacbn& base_in) {
    typedef CombTab<BITS> Tab;
    const int E = 1 << BITS;
    const modinv_info PI = MODINV_BNP_INFO;
    // affine base of this window (so that the chain below can use mixed additions)
    fbn bx, by;
    {
        u256 zp, zinv;
        fbn zm, z2, z3;
        fe_from_mont(zp, base_in.Z);
        modinv(zinv, zp, PI);
        fe_to_mont(zm, zinv);
        fe_sqr(z2, zm);
        fe_mul(z3, z2, zm);
        fe_mul(bx, base_in.X, z2);
        fe_mul(by, base_in.Y, z3);
    }
    std::vector<jacbn> pts(E);
    std::vector<fbn> pre(E), zi(E);
    pts[1].X = bx;
    pts[1].Y = by;
    fe_set_one(pts[1].Z);
    for (int d = 2; d < E; d++) {
        if ((d & 1) == 0) {
            pt_dbl29(pts[d], pts[d >> 1]);
        } else {
            fbn h, rr;
            pt_add_mixed29(pts[d], pts[d - 1], bx, by, h, rr);
        }
    }
    // Montgomery's trick: one inversion for the window.  (Z of a doubling has L = 2: products stay far inside the bound.)
    pre[1] = pts[1].Z;
    for (int d = 2; d < E; d++) fe_mul(pre[d], pre[d - 1], pts[d].Z);
    fbn inv;
    {
        u256 ap, ai;
        fe_from_mont(ap, pre[E - 1]);
        modinv(ai, ap, PI);
        fe_to_mont(inv, ai);
    }
    for (int d = E - 1; d >= 2; d--) {
        fe_mul(zi[d], inv, pre[d - 1]);
        fe_mul(inv, inv, pts[d].Z);
    }
    zi[1] = inv;
    for (int d = 1; d < E; d++) {
        fbn z2, z3, xm, ym, fx, fy;
        u256 x, y;
        fe_sqr(z2, zi[d]);
        fe_mul(z3, z2, zi[d]);
        fe_mul(xm, pts[d].X, z2);
        fe_mul(ym, pts[d].Y, z3);
        fe_from_mont(x, xm);      // canonical, then back: table entries are the unique normalised form of the coordinate
        fe_from_mont(y, ym);
        fe_to_mont(fx, x);
        fe_to_mont(fy, y);
        int32_t* e = words + Tab::index(w, (uint32_t)d);
        for (int l = 0; l < 9; l++) {
            e[l] = fx.v[l];
            e[9 + l] = fy.v[l];
        }
    }
}

// words: CombTab<BITS>::TABLE_WORDS.  (bxp, byp): an affine point of G1, plain integers below p.  Windows are independent once
// their bases 2^(BITS w) B are known: one worker per window, up to max_threads.
template <int BITS>
inline void build_bn_comb_table(int32_t* words, const u256& bxp, const u256& byp, int max_threads = 16) {
    typedef CombTab<BITS> Tab;
    for (size_t i = 0; i < Tab::TABLE_WORDS; i++) words[i] = 0;
    std::vector<jacbn> bases(Tab::WINDOWS);
    fe_to_mont(bases[0].X, bxp);
    fe_to_mont(bases[0].Y, byp);
    fe_set_one(bases[0].Z);
    for (int w = 1; w < Tab::WINDOWS; w++) {
        jacbn t = bases[w - 1];
        for (int k = 0; k < BITS; k++) {
            jacbn d2;
            pt_dbl29(d2, t);
            t = d2;
        }
        bases[w] = t;
    }
    int nt = max_threads < 1 ? 1 : (max_threads > Tab::WINDOWS ? Tab::WINDOWS : max_threads);
    if (BITS <= 8) nt = 1;        // an 8-bit table is 8 K points: not worth threads
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            for (int w = t; w < Tab::WINDOWS; w += nt) build_bn_comb_window<BITS>(words, w, bases[w]);
        });
    for (auto& x : th) x.join();
}
inline void build_bn_comb_table8(int32_t* words, const u256& bxp, const u256& byp) { build_bn_comb_table<8>(words, bxp, byp); }

}  // namespace fab
