// Host-side precomputation of the 8-bit comb tables of an idemix issuer's bases HSk and HRand (ec29.h CombTab<8>, fbn
// elements): T[w][d] = d * 2^(8 w) * B, affine, Montgomery form, balanced digits - built at fabgpu_idemix_issuer_register
// with the SAME field and point code the kernels run (bn29.h / bn_nym29.h compiled for the host).
#pragma once
#include <vector>

#include "bn_nym29.h"

namespace fab {

// words: KeyTab8::TABLE_WORDS.  (bxp, byp): an affine point of G1, plain integers below p.
inline void build_bn_comb_table8(int32_t* words, const u256& bxp, const u256& byp) {
    typedef KeyTab8 Tab;
    const int E = 256;
    for (size_t i = 0; i < Tab::TABLE_WORDS; i++) words[i] = 0;
    jacbn base;
    fe_to_mont(base.X, bxp);
    fe_to_mont(base.Y, byp);
    fe_set_one(base.Z);
    const modinv_info PI = MODINV_BNP_INFO;
    std::vector<jacbn> pts(E);
    std::vector<fbn> pre(E), zi(E);
    for (int w = 0; w < Tab::WINDOWS; w++) {
        // affine base of this window (so that the chain below can use mixed additions)
        fbn bx, by;
        {
            u256 zp, zinv;
            fbn zm, z2, z3;
            fe_from_mont(zp, base.Z);
            modinv(zinv, zp, PI);
            fe_to_mont(zm, zinv);
            fe_sqr(z2, zm);
            fe_mul(z3, z2, zm);
            fe_mul(bx, base.X, z2);
            fe_mul(by, base.Y, z3);
        }
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
        for (int k = 0; k < 8; k++) {
            jacbn d2;
            pt_dbl29(d2, base);
            base = d2;
        }
    }
}

}  // namespace fab
