// Host-side precomputation of the generator comb table in the fe29 representation (64 windows x 15 affine points,
// T[i][j] = j * 2^(4 i) * G, Montgomery form with R = 2^261, balanced 29-bit digits), in the LDS layout of
// p256_verify29.h::g29_index.  Built once per fabgpu_init from the u256 table (p256_tables.h) and uploaded to each device.
#pragma once
#include <vector>

#include "p256_tables.h"
#include "p256_verify29.h"

namespace fab {

inline void build_g_comb_table29(int32_t* words) {
    std::vector<uint32_t> old(G_TABLE_WORDS);
    build_g_comb_table(old.data());
    FlatGTab gt{old.data()};
    for (int i = 0; i < G29_TABLE_WORDS; i++) words[i] = 0;
    for (int w = 0; w < G29_WINDOWS; w++) {
        for (int j = 1; j <= G29_ENTRIES; j++) {
            u256 xm, ym, x, y;
            gt.load(w, (uint32_t)j, xm, ym);
            fp_from_mont(x, xm);
            fp_from_mont(y, ym);
            fe fx, fy;
            fe_to_mont(fx, x);
            fe_to_mont(fy, y);
            for (int l = 0; l < 9; l++) {
                words[g29_index(w, 0, l, j - 1)] = fx.v[l];
                words[g29_index(w, 1, l, j - 1)] = fy.v[l];
            }
        }
    }
}

}  // namespace fab
