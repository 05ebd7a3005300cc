// Host-side precomputation of the generator comb table used by the verify kernel
// (64 windows x 15 affine Montgomery points: T[i][j] = j * 2^(4 i) * G), in the LDS layout of
// p256_point.h::g_index.  Built once per fabgpu_init and uploaded to each device.
#pragma once
#include <string.h>

#include <vector>

#include "p256_point.h"

namespace fab {

#define FAB_P256_GX_PLAIN {0xD898C296u, 0xF4A13945u, 0x2DEB33A0u, 0x77037D81u, 0x63A440F2u, 0xF8BCE6E5u, 0xE12C4247u, 0x6B17D1F2u}
#define FAB_P256_GY_PLAIN {0x37BF51F5u, 0xCBB64068u, 0x6B315ECEu, 0x2BCE3357u, 0x7C0F9E16u, 0x8EE7EB4Au, 0xFE1A7F9Bu, 0x4FE342E2u}

// Jacobian (Montgomery) -> affine (Montgomery); host only (uses the Fermat inversion)
inline void jac_to_affine_mont(u256& x, u256& y, const jac& a) {
    u256 zi, zi2, zi3;
    fp_inv(zi, a.Z);
    fp_sqr(zi2, zi);
    fp_mul(zi3, zi2, zi);
    fp_mul(x, a.X, zi2);
    fp_mul(y, a.Y, zi3);
}

inline void build_g_comb_table(uint32_t* words) {
    memset(words, 0, sizeof(uint32_t) * G_TABLE_WORDS);
    const u256 gxp = FAB_P256_GX_PLAIN, gyp = FAB_P256_GY_PLAIN;
    const u256 ONE = FAB_P256_R1;
    jac base;  // 2^(4 i) G
    fp_to_mont(base.X, gxp);
    fp_to_mont(base.Y, gyp);
    base.Z = ONE;
    for (int i = 0; i < G_WINDOWS; i++) {
        u256 bx, by;
        jac_to_affine_mont(bx, by, base);
        jac cur = base;
        for (int j = 1; j <= G_ENTRIES; j++) {
            u256 x, y;
            jac_to_affine_mont(x, y, cur);
            for (int l = 0; l < 8; l++) {
                words[g_index(i, 0, l, j - 1)] = x.w[l];
                words[g_index(i, 1, l, j - 1)] = y.w[l];
            }
            if (j == 1) {
                jac d;
                pt_dbl(d, cur);
                cur = d;
            } else {
                jac s;
                bool hz, rz;
                pt_add_mixed(s, cur, bx, by, hz, rz);
                cur = s;
            }
        }
        // after the loop cur = 16 * base
        base = cur;
    }
}

// table accessor over a flat word array (host memory, or LDS on the device)
struct FlatGTab {
    const uint32_t* w;
    FAB_HD void load(int window, uint32_t digit, u256& x, u256& y) const {
#pragma unroll
        for (int l = 0; l < 8; l++) {
            x.w[l] = w[g_index(window, 0, l, (int)digit - 1)];
            y.w[l] = w[g_index(window, 1, l, (int)digit - 1)];
        }
    }
};

}  // namespace fab
