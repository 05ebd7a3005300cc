// Host-side precomputation of comb tables in the fe29 representation (p256_verify29.h::CombTab<BITS>):
// T[w][d] = d * 2^(BITS w) * B, affine, Montgomery form with R = 2^261, balanced 29-bit digits, for B = the generator
// (BITS = 16, once per fabgpu_init) or a registered public key (BITS = 8, fabgpu_p256_key_register).
// Built with the host u256 arithmetic (fp256.h / p256_point.h), one thread per window.
#pragma once
#include <thread>
#include <vector>

#include "p256_tables.h"
#include "p256_verify29.h"

namespace fab {

// one window: entries d = 1 .. 2^BITS - 1 of base (Jacobian, Montgomery u256), one inversion for the whole window
template <int BITS>
inline void build_comb_window(int32_t* words, int w, const jac& base) {
    const int E = 1 << BITS;
    u256 bx, by;
    jac_to_affine_mont(bx, by, base);
    std::vector<jac> pts(E);
    pts[1] = base;
    for (int d = 2; d < E; d++) {
        if ((d & 1) == 0) {
            pt_dbl(pts[d], pts[d >> 1]);
        } else {
            bool hz, rz;
            pt_add_mixed(pts[d], pts[d - 1], bx, by, hz, rz);
        }
    }
    std::vector<u256> pre(E), zi(E);   // Montgomery's trick
    pre[1] = pts[1].Z;
    for (int d = 2; d < E; d++) fp_mul(pre[d], pre[d - 1], pts[d].Z);
    u256 inv;
    fp_inv(inv, pre[E - 1]);
    for (int d = E - 1; d >= 2; d--) {
        fp_mul(zi[d], inv, pre[d - 1]);
        fp_mul(inv, inv, pts[d].Z);
    }
    zi[1] = inv;
    for (int d = 1; d < E; d++) {
        u256 zi2, zi3, xm, ym, x, y;
        fp_sqr(zi2, zi[d]);
        fp_mul(zi3, zi2, zi[d]);
        fp_mul(xm, pts[d].X, zi2);
        fp_mul(ym, pts[d].Y, zi3);
        fp_from_mont(x, xm);
        fp_from_mont(y, ym);
        fe fx, fy;
        fe_to_mont(fx, x);
        fe_to_mont(fy, y);
        int32_t* e = words + CombTab<BITS>::index(w, (uint32_t)d);
        for (int l = 0; l < 9; l++) {
            e[l] = fx.v[l];
            e[9 + l] = fy.v[l];
        }
    }
}

// words: CombTab<BITS>::TABLE_WORDS.  (bxp, byp): an affine point of the curve, plain integers.
template <int BITS>
inline void build_comb_table(int32_t* words, const u256& bxp, const u256& byp, int max_threads = 16) {
    typedef CombTab<BITS> Tab;
    for (size_t i = 0; i < Tab::TABLE_WORDS; i++) words[i] = 0;
    const u256 ONE = FAB_P256_R1;
    std::vector<jac> bases(Tab::WINDOWS);   // 2^(BITS w) B
    fp_to_mont(bases[0].X, bxp);
    fp_to_mont(bases[0].Y, byp);
    bases[0].Z = ONE;
    for (int w = 1; w < Tab::WINDOWS; w++) {
        jac t = bases[w - 1];
        for (int k = 0; k < BITS; k++) {
            jac d;
            pt_dbl(d, t);
            t = d;
        }
        bases[w] = t;
    }
    int nt = max_threads < 1 ? 1 : (max_threads > Tab::WINDOWS ? Tab::WINDOWS : max_threads);
    if ((size_t)1 << BITS <= 256) nt = 1;   // a key table is 8 K points: not worth threads
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            for (int w = t; w < Tab::WINDOWS; w += nt) build_comb_window<BITS>(words, w, bases[w]);
        });
    for (auto& x : th) x.join();
}

inline void build_g_comb_table16(int32_t* words) {
    const u256 gxp = FAB_P256_GX_PLAIN, gyp = FAB_P256_GY_PLAIN;
    build_comb_table<16>(words, gxp, gyp);
}
inline void build_key_comb_table8(int32_t* words, const u256& qx, const u256& qy) { build_comb_table<8>(words, qx, qy); }

}  // namespace fab
