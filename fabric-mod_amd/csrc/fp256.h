// 256-bit modular arithmetic for secp256r1 (NIST P-256) on 8 x 32-bit limbs.
//
// Written for the CDNA4 integer VALU: the only wide multiplier the ISA offers is
// v_mad_u64_u32 (32x32 -> 64 plus a 64-bit addend, carry-out in VCC), so numbers are 8 little-
// endian u32 limbs and every partial product is one v_mad_u64_u32 plus one v_addc_co_u32 into a
// 96-bit column accumulator (product scanning).  Field elements mod p live in the Montgomery
// domain (R = 2^256); p = 2^256 - 2^224 + 2^192 + 2^96 - 1 satisfies p = -1 mod 2^96, so the
// Montgomery quotient digit is the accumulator limb itself and q*p is added with shifts only.
// Scalars mod the group order n use generic Montgomery reduction (n has no special form).
//
// The same header compiles for the host (synthetic-block generator, table precomputation,
// CPU-side KeyImport gate), where the MAC falls back to portable 64-bit C.
//
// Replaces, on the device, the arithmetic that the reference reaches through
// bccsp/sw/ecdsa.go:56 (Go crypto/ecdsa.Verify -> crypto/elliptic p256 assembly).
#pragma once
#include <stdint.h>

#if defined(__HIP__)
#define FAB_HD __host__ __device__ inline __attribute__((always_inline))
#define FAB_D __device__ inline __attribute__((always_inline))        // device only (generated one29_gcn.h and what calls it)
#else
#define FAB_HD inline __attribute__((always_inline))
#endif
// The multiply / square bodies are real functions in device code (one copy each, arguments and result in
// VGPRs): fully inlined, the verify kernel was 380 KB of straight-line code and thrashed the 64 KB
// instruction cache (measured 4x slower than its instruction count, profiles/r01_bench_v0_inlined.txt).
#if defined(__HIP_DEVICE_COMPILE__)
#define FAB_FN __device__ __attribute__((noinline))
#elif defined(__HIP__)
#define FAB_FN __host__ __device__ inline
#else
#define FAB_FN inline
#endif

namespace fab {

struct u256 {
    uint32_t w[8];
};

// ---- constants (little-endian limbs) -------------------------------------------------------
#define FAB_P256_P {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000001u, 0xFFFFFFFFu}
#define FAB_P256_N {0xFC632551u, 0xF3B9CAC2u, 0xA7179E84u, 0xBCE6FAADu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x00000000u, 0xFFFFFFFFu}
// n >> 1 (bccsp/utils/ecdsa.go:27-33 curveHalfOrders)
#define FAB_P256_HALF_N {0x7E3192A8u, 0x79DCE561u, 0xD38BCF42u, 0xDE737D56u, 0xFFFFFFFFu, 0x7FFFFFFFu, 0x80000000u, 0x7FFFFFFFu}
// p - n  (x mod n == r  <=>  x == r  or  (r < p - n and x == r + n))
#define FAB_P256_P_MINUS_N {0x039CDAAEu, 0x0C46353Du, 0x58E8617Bu, 0x43190553u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u}
// R mod p (Montgomery one), R^2 mod p, curve b in Montgomery form
#define FAB_P256_R1 {0x00000001u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFEu, 0x00000000u}
#define FAB_P256_R2 {0x00000003u, 0x00000000u, 0xFFFFFFFFu, 0xFFFFFFFBu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFDu, 0x00000004u}
#define FAB_P256_B_MONT {0x29C4BDDFu, 0xD89CDF62u, 0x78843090u, 0xACF005CDu, 0xF7212ED6u, 0xE5A220ABu, 0x04874834u, 0xDC30061Du}
// R mod n, R^2 mod n, -n^-1 mod 2^32
#define FAB_P256_N_R1 {0x039CDAAFu, 0x0C46353Du, 0x58E8617Bu, 0x43190552u, 0x00000000u, 0x00000000u, 0xFFFFFFFFu, 0x00000000u}
#define FAB_P256_N_R2 {0xBE79EEA2u, 0x83244C95u, 0x49BD6FA6u, 0x4699799Cu, 0x2B6BEC59u, 0x2845B239u, 0xF3D95620u, 0x66E12D94u}
#define FAB_P256_N0INV 0xEE00BC4Fu

// ---- carry primitives -------------------------------------------------------------------------
FAB_HD uint32_t addc(uint32_t a, uint32_t b, uint32_t& c) {
    uint32_t co;
    uint32_t r = __builtin_addc(a, b, c, &co);
    c = co;
    return r;
}
FAB_HD uint32_t subb(uint32_t a, uint32_t b, uint32_t& br) {
    uint32_t bo;
    uint32_t r = __builtin_subc(a, b, br, &bo);
    br = bo;
    return r;
}

// (ex:hi:lo) += a * b      -- the inner-loop MAC
FAB_HD void mac(uint32_t& lo, uint32_t& hi, uint32_t& ex, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t acc = ((uint64_t)hi << 32) | lo;
    // gfx950 hazard: a VALU write of VCC/SGPR needs 2 wait states before a VALU reads it as carry-in
    // (LLVM GCNHazardRecognizer VALUWriteSGPRVALURead); hipcc does not pad inside asm statements.
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ex)
        : "v"(a), "v"(b)
        : "vcc");
    lo = (uint32_t)acc;
    hi = (uint32_t)(acc >> 32);
#else
    uint64_t acc = ((uint64_t)hi << 32) | lo;
    uint64_t p = (uint64_t)a * b;
    acc += p;
    ex += (acc < p);
    lo = (uint32_t)acc;
    hi = (uint32_t)(acc >> 32);
#endif
}
// (hi:lo) = a * b + (hi:lo), caller guarantees no overflow
FAB_HD void mac_nc(uint32_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
    uint64_t acc = (((uint64_t)hi << 32) | lo) + (uint64_t)a * b;
    lo = (uint32_t)acc;
    hi = (uint32_t)(acc >> 32);
}

// ---- plain 256-bit helpers ---------------------------------------------------------------------
FAB_HD uint32_t add256(u256& r, const u256& a, const u256& b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = addc(a.w[i], b.w[i], c);
    return c;
}
FAB_HD uint32_t sub256(u256& r, const u256& a, const u256& b) {
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = subb(a.w[i], b.w[i], br);
    return br;
}
FAB_HD bool is_zero(const u256& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.w[i];
    return o == 0;
}
FAB_HD bool eq256(const u256& a, const u256& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.w[i] ^ b.w[i];
    return o == 0;
}
// a < b
FAB_HD bool lt256(const u256& a, const u256& b) {
    u256 t;
    return sub256(t, a, b) != 0;
}
FAB_HD void sel256(u256& r, bool c, const u256& a, const u256& b) {  // r = c ? a : b
#pragma unroll
    for (int i = 0; i < 8; i++) r.w[i] = c ? a.w[i] : b.w[i];
}
FAB_HD u256 zero256() {
    u256 z;
#pragma unroll
    for (int i = 0; i < 8; i++) z.w[i] = 0;
    return z;
}
// 32 big-endian bytes (the C-ABI field format) <-> limbs
FAB_HD void from_be32(u256& r, const uint8_t* be) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t* p = be + 4 * (7 - i);
        r.w[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    }
}
FAB_HD void to_be32(uint8_t* be, const u256& a) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint8_t* p = be + 4 * (7 - i);
        p[0] = (uint8_t)(a.w[i] >> 24);
        p[1] = (uint8_t)(a.w[i] >> 16);
        p[2] = (uint8_t)(a.w[i] >> 8);
        p[3] = (uint8_t)a.w[i];
    }
}

// ---- 256 x 256 -> 512 (product scanning) ----------------------------------------------------------
FAB_HD void mul512(uint32_t t[16], const u256& a, const u256& b) {
    uint32_t lo = 0, hi = 0, ex = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int j = k - i;
            if (j >= 0 && j < 8) mac(lo, hi, ex, a.w[i], b.w[j]);
        }
        t[k] = lo;
        lo = hi;
        hi = ex;
        ex = 0;
    }
    t[15] = lo;
}
// a^2: off-diagonal products once, doubled, plus the diagonal
FAB_HD void sqr512(uint32_t t[16], const u256& a) {
    uint32_t lo = 0, hi = 0, ex = 0;
    t[0] = 0;
#pragma unroll
    for (int k = 1; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int j = k - i;
            if (j > i && j < 8) mac(lo, hi, ex, a.w[i], a.w[j]);
        }
        t[k] = lo;
        lo = hi;
        hi = ex;
        ex = 0;
    }
    t[15] = lo;
    // double
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = addc(t[k], t[k], c);
    // add diagonal squares
    c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t dl = 0, dh = 0;
        mac_nc(dl, dh, a.w[i], a.w[i]);
        t[2 * i] = addc(t[2 * i], dl, c);
        t[2 * i + 1] = addc(t[2 * i + 1], dh, c);
    }
}

// ---- Montgomery reduction mod p: r = T / 2^256 mod p, T < p * 2^256 ----------------------------------
// Round i: q = t[i]; T += q * p * 2^(32 i) with q*p = q*2^256 - q*2^224 + q*2^192 + q*2^96 - q.
// The "- q" cancels t[i]; (q*2^256 - q*2^224) is the non-negative 64-bit value q*(2^32-1) at limb i+7.
FAB_HD void redc_p(u256& r, uint32_t t[16]) {
    uint32_t top = 0;  // limb 16
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t q = t[i];
        uint32_t br = 0;
        uint32_t vlo = subb(0u, q, br);  // (q << 32) - q
        uint32_t vhi = subb(q, 0u, br);
        uint32_t c = 0;
        t[i + 3] = addc(t[i + 3], q, c);
        t[i + 4] = addc(t[i + 4], 0u, c);
        t[i + 5] = addc(t[i + 5], 0u, c);
        t[i + 6] = addc(t[i + 6], q, c);
        t[i + 7] = addc(t[i + 7], vlo, c);
        if (i + 8 < 16) {
            t[i + 8] = addc(t[i + 8], vhi, c);
#pragma unroll
            for (int k = i + 9; k < 16; k++) t[k] = addc(t[k], 0u, c);
            top += c;
        } else {
            top = top + vhi + c;  // i == 8 never happens (i < 8 -> i + 8 <= 15)
        }
    }
    // result = t[8..15] + top*2^256, < 2p; subtract p once if needed
    const u256 P = FAB_P256_P;
    u256 x, y;
#pragma unroll
    for (int i = 0; i < 8; i++) x.w[i] = t[8 + i];
    uint32_t br = sub256(y, x, P);
    sel256(r, (top != 0) | (br == 0), y, x);
}

FAB_FN u256 fp_mul_fn(u256 a, u256 b) {
    uint32_t t[16];
    u256 r;
    mul512(t, a, b);
    redc_p(r, t);
    return r;
}
FAB_FN u256 fp_sqr_fn(u256 a) {
    uint32_t t[16];
    u256 r;
    sqr512(t, a);
    redc_p(r, t);
    return r;
}
FAB_HD void fp_mul(u256& r, const u256& a, const u256& b) { r = fp_mul_fn(a, b); }
FAB_HD void fp_sqr(u256& r, const u256& a) { r = fp_sqr_fn(a); }
FAB_HD void fp_add(u256& r, const u256& a, const u256& b) {
    const u256 P = FAB_P256_P;
    u256 t, u;
    uint32_t c = add256(t, a, b);
    uint32_t br = sub256(u, t, P);
    sel256(r, (c != 0) | (br == 0), u, t);
}
FAB_HD void fp_sub(u256& r, const u256& a, const u256& b) {
    u256 t;
    uint32_t br = sub256(t, a, b);
    uint32_t m = 0u - br;  // all-ones when a < b: add p back
    uint32_t c = 0;
    r.w[0] = addc(t.w[0], m, c);
    r.w[1] = addc(t.w[1], m, c);
    r.w[2] = addc(t.w[2], m, c);
    r.w[3] = addc(t.w[3], 0u, c);
    r.w[4] = addc(t.w[4], 0u, c);
    r.w[5] = addc(t.w[5], 0u, c);
    r.w[6] = addc(t.w[6], m & 1u, c);
    r.w[7] = addc(t.w[7], m, c);
}
FAB_HD void fp_dbl(u256& r, const u256& a) { fp_add(r, a, a); }
FAB_HD void fp_to_mont(u256& r, const u256& a) {
    const u256 R2 = FAB_P256_R2;
    fp_mul(r, a, R2);
}
FAB_HD void fp_from_mont(u256& r, const u256& a) {
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t[i] = a.w[i];
        t[8 + i] = 0;
    }
    redc_p(r, t);
}
// a^(p-2): only used off the hot path (host table build, tests); the verify kernel never inverts mod p.
FAB_HD void fp_inv(u256& r, const u256& a) {
    const u256 P = FAB_P256_P;
    const u256 ONE = FAB_P256_R1;
    u256 acc = ONE;
    for (int i = 255; i >= 0; i--) {
        fp_sqr(acc, acc);
        uint32_t e = (i == 0) ? 1u : ((i == 1) ? 0u : ((P.w[i >> 5] >> (i & 31)) & 1u));  // p - 2: bit0 = 1, bit1 = 0
        if (e) fp_mul(acc, acc, a);
    }
    r = acc;
}

// ---- generic Montgomery arithmetic mod n (group order) ---------------------------------------------------
FAB_HD void redc_n(u256& r, uint32_t t[16]) {
    const u256 N = FAB_P256_N;
    uint32_t top = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t q = t[i] * FAB_P256_N0INV;
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t v = (uint64_t)q * N.w[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)v;
            carry = (uint32_t)(v >> 32);
        }
        uint32_t c = 0;
        if (i + 8 < 16) {
            t[i + 8] = addc(t[i + 8], carry, c);
#pragma unroll
            for (int k = i + 9; k < 16; k++) t[k] = addc(t[k], 0u, c);
            top += c;
        } else {
            top += carry;
        }
    }
    u256 x, y;
#pragma unroll
    for (int i = 0; i < 8; i++) x.w[i] = t[8 + i];
    uint32_t br = sub256(y, x, N);
    sel256(r, (top != 0) | (br == 0), y, x);
}
FAB_FN u256 fn_mul_fn(u256 a, u256 b) {
    uint32_t t[16];
    u256 r;
    mul512(t, a, b);
    redc_n(r, t);
    return r;
}
FAB_HD void fn_mul(u256& r, const u256& a, const u256& b) { r = fn_mul_fn(a, b); }
FAB_HD void fn_sqr(u256& r, const u256& a) { r = fn_mul_fn(a, a); }
FAB_HD void fn_to_mont(u256& r, const u256& a) {
    const u256 R2 = FAB_P256_N_R2;
    fn_mul(r, a, R2);
}
FAB_HD void fn_from_mont(u256& r, const u256& a) {
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        t[i] = a.w[i];
        t[8 + i] = 0;
    }
    redc_n(r, t);
}
FAB_HD void fn_add(u256& r, const u256& a, const u256& b) {
    const u256 N = FAB_P256_N;
    u256 t, u;
    uint32_t c = add256(t, a, b);
    uint32_t br = sub256(u, t, N);
    sel256(r, (c != 0) | (br == 0), u, t);
}
// a^(n-2) mod n, Montgomery domain in and out, 4-bit fixed window (n is public: no secret-dependent flow)
FAB_HD void fn_inv(u256& r, const u256& a) {
    const u256 N = FAB_P256_N;
    u256 e = N;
    e.w[0] -= 2;  // n - 2 (no borrow: low limb is ...2551)
    u256 tab[16];
    tab[0] = FAB_P256_N_R1;
    tab[1] = a;
    for (int i = 2; i < 16; i++) fn_mul(tab[i], tab[i - 1], a);
    u256 acc = tab[(e.w[7] >> 28) & 15];
    for (int i = 62; i >= 0; i--) {
        fn_sqr(acc, acc);
        fn_sqr(acc, acc);
        fn_sqr(acc, acc);
        fn_sqr(acc, acc);
        uint32_t d = (e.w[i >> 3] >> ((i & 7) * 4)) & 15;
        fn_mul(acc, acc, tab[d]);
    }
    r = acc;
}

}  // namespace fab
