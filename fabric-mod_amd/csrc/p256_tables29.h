// Host-side precomputation of 8-bit comb tables in the fe29 representation: 32 windows x 255 affine points,
// T[w][d] = d * 2^(8 w) * B for B = the generator (once per fabgpu_init) or a registered public key (fabgpu_p256_key_register), Montgomery form with R = 2^261, balanced 29-bit digits, layout p256_verify29.h::g8_index.
// Built once per fabgpu_init with the host u256 arithmetic (fp256.h / p256_point.h) and uploaded to each device.
#pragma once
#include <vector>

#include "p256_tables.h"
#include "p256_verify29.h"

namespace fab {

// words: G8_TABLE_WORDS.  (bx, by): an affine point of the curve, plain integers (the generator, or a registered public key).
inline void build_comb8_table(int32_t* words, const u256& bxp, const u256& byp) {
    for (int i = 0; i < G8_TABLE_WORDS; i++) words[i] = 0;
    const u256 ONE = FAB_P256_R1;
    jac base;  // 2^(8 w) B
    fp_to_mont(base.X, bxp);
    fp_to_mont(base.Y, byp);
    base.Z = ONE;
    std::vector<jac> pts(256);
    for (int w = 0; w < G8_WINDOWS; w++) {
        u256 bx, by;
        jac_to_affine_mont(bx, by, base);
        pts[1] = base;
        for (int d = 2; d < 256; d++) {
            if ((d & 1) == 0) {
                pt_dbl(pts[d], pts[d >> 1]);
            } else {
                bool hz, rz;
                pt_add_mixed(pts[d], pts[d - 1], bx, by, hz, rz);
            }
        }
        // one inversion for the whole window (Montgomery's trick), host only
        std::vector<u256> pre(256), zi(256);
        pre[1] = pts[1].Z;
        for (int d = 2; d < 256; d++) fp_mul(pre[d], pre[d - 1], pts[d].Z);
        u256 inv;
        fp_inv(inv, pre[255]);
        for (int d = 255; d >= 2; d--) {
            fp_mul(zi[d], inv, pre[d - 1]);
            fp_mul(inv, inv, pts[d].Z);
        }
        zi[1] = inv;
        for (int d = 1; d < 256; d++) {
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
            int32_t* e = words + g8_index(w, (uint32_t)d);
            for (int l = 0; l < 9; l++) {
                e[l] = fx.v[l];
                e[9 + l] = fy.v[l];
            }
        }
        // next window: 2^8 * base = 2 * pts[128]
        jac nb;
        pt_dbl(nb, pts[128]);
        base = nb;
    }
}
inline void build_g8_comb_table(int32_t* words) {
    const u256 gxp = FAB_P256_GX_PLAIN, gyp = FAB_P256_GY_PLAIN;
    build_comb8_table(words, gxp, gyp);
}

}  // namespace fab
