// GPU-SIDE TEST HOOKS - built into libfabgpu_gputest.so, never into the product library.  They run single generated
// instruction streams (pair29_gcn.h) on one wavefront so that tests/test_gpu_parity.py can compare every output register
// with the reference interpreter of gcn_dsl.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "p256_pair29.h"
#include "pair29_bn_gcn.h"   // the BN pair programs of bn_quad29.h: validated here register for register
#include "p256_tables29.h"
#include "bn_nym29.h"
#include "bn_quad29.h"
#include "bn_tables29.h"
#include "device_common.h"

using namespace fab;

// in : per lane A[9] B[9] C[9] D[9];  out: per lane A[9] B[9] H[9] RR[9]
__global__ void __launch_bounds__(64, 1) gputest_pair_kernel(int op, const int32_t* __restrict__ in, int32_t* __restrict__ out) {
    const int32_t* p = in + threadIdx.x * 36;
    pair_pt P;
    fe C, D;
    for (int i = 0; i < 9; i++) {
        P.A.v[i] = p[i];
        P.B.v[i] = p[9 + i];
        C.v[i] = p[18 + i];
        D.v[i] = p[27 + i];
    }
    PAIR_TMPS;
    for (int i = 0; i < 9; i++) tH.v[i] = tRR.v[i] = 0;
    if (op == 3) {   // ISA probe: out A = v_subrev_u32_dpp(A, B), out B = v_sub_u32_dpp(A, B), out H = v_add_u32_dpp(A, B), out RR = v_mov_b32_dpp(A)
        fe a = P.A, b = P.B;
        for (int i = 0; i < 9; i++) {
            asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_sub_u32_dpp %1, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_u32_dpp %2, %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 4"
                         : "=&v"(P.A.v[i]), "=&v"(P.B.v[i]), "=&v"(tH.v[i]), "=&v"(tRR.v[i])
                         : "v"(a.v[i]), "v"(b.v[i]));
        }
    } else if (op == 0) {
        PAIR_DBL(P);
    } else if (op == 1) {
        pair_pt R;
        PAIR_ADD(R, P, C, D);
        P = R;
    } else if (op == 2) {
        pair_pt R;
        PAIR_MADD(R, P, C, D);
        P = R;
    } else if (op == 4) {           // the same three programs for FP256BN's field and a = 0 (pair29_bn_gcn.h)
        PAIRBN_DBL(P.A, P.B, tU1, tU2, tU3, tU4, tP1, tP2, tT0, tT1, tTD);
    } else if (op == 5) {
        pair_pt R;
        PAIRBN_ADD(R.A, R.B, P.A, P.B, tH, tRR, tW, tU1, tU2, tU3, tU4, tU6, tP1, tP2, tT0, tT1, tTD, C, D);
        P = R;
    } else {
        pair_pt R;
        PAIRBN_MADD(R.A, R.B, P.A, P.B, tU1, tU2, tU3, tU4, tH, tRR, tP1, tP2, tT0, tT1, tTD, C, D);
        P = R;
    }
    int32_t* o = out + threadIdx.x * 36;
    for (int i = 0; i < 9; i++) {
        o[i] = P.A.v[i];
        o[9 + i] = P.B.v[i];
        o[18 + i] = tH.v[i];
        o[27 + i] = tRR.v[i];
    }
}

extern "C" int gputest_pair_op(int op, const int32_t* in, int32_t* out) {
    int32_t *din = nullptr, *dout = nullptr;
    const size_t bytes = 64 * 36 * sizeof(int32_t);
    if (hipMalloc((void**)&din, bytes) != hipSuccess || hipMalloc((void**)&dout, bytes) != hipSuccess) return -1;
    hipMemcpy(din, in, bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gputest_pair_kernel, dim3(1), dim3(64), 0, 0, op, din, dout);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost);
    hipFree(din);
    hipFree(dout);
    return rc;
}

// R = u1*G + u2*Q through pair_combined_mult29 on one wavefront (32 signatures).  in: 32 x (u1, u2, qx, qy) big-endian 32-byte
// fields;  out: per lane A[9] B[9] r_inf
__global__ void __launch_bounds__(64, 1) gputest_pair_combined_kernel(const uint8_t* __restrict__ in, const int32_t* __restrict__ gtab,
                                                                       uint4* __restrict__ qws, int32_t* __restrict__ out) {
    const bool odd = (threadIdx.x & 1) != 0;
    const uint32_t k = threadIdx.x >> 1;
    u256 u1, u2, qx, qy;
    from_be32(u1, in + 128 * k);
    from_be32(u2, in + 128 * k + 32);
    from_be32(qx, in + 128 * k + 64);
    from_be32(qy, in + 128 * k + 96);
    fe QX, QY;
    fe_to_mont(QX, qx);
    fe_to_mont(QY, qy);
    PairQTab<32> qtab = PairQTab<32>::of(qws, k);
    pair_pt R;
    bool inf;
    pair_combined_mult29(R, inf, u1, u2, QX, QY, gtab, qtab, odd);
    int32_t* o = out + threadIdx.x * 19;
    for (int i = 0; i < 9; i++) {
        o[i] = R.A.v[i];
        o[9 + i] = R.B.v[i];
    }
    o[18] = inf ? 1 : 0;
}

extern "C" int gputest_pair_combined(const uint8_t* in, int32_t* out) {
    std::vector<int32_t> tab(GTab16::TABLE_WORDS);
    build_g_comb_table16(tab.data());
    uint8_t* din = nullptr;
    int32_t *dtab = nullptr, *dout = nullptr;
    uint4* dws = nullptr;
    if (hipMalloc((void**)&din, 32 * 128) != hipSuccess || hipMalloc((void**)&dtab, sizeof(int32_t) * GTab16::TABLE_WORDS) != hipSuccess ||
        hipMalloc((void**)&dout, 64 * 19 * 4) != hipSuccess || hipMalloc((void**)&dws, (size_t)16 * 8 * 32 * 16) != hipSuccess)
        return -1;
    hipMemcpy(din, in, 32 * 128, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), sizeof(int32_t) * GTab16::TABLE_WORDS, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gputest_pair_combined_kernel, dim3(1), dim3(64), 0, 0, din, dtab, dws, dout);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    hipMemcpy(out, dout, 64 * 19 * 4, hipMemcpyDeviceToHost);
    hipFree(din); hipFree(dtab); hipFree(dout); hipFree(dws);
    return rc;
}

// Debug probe of p256_verify_pair29: same statements, intermediate values written out.
// in: 32 x (qx, qy, e, r, s);  out per lane: u1[8] u2[8] A[9] B[9] flags(st, ok1, ok2, r_inf, early)
__global__ void __launch_bounds__(64, 1) gputest_pair_verify_kernel(const uint8_t* __restrict__ in, const int32_t* __restrict__ gtab,
                                                                     uint4* __restrict__ qws, int32_t* __restrict__ out) {
    const bool odd = (threadIdx.x & 1) != 0;
    const uint32_t k = threadIdx.x >> 1;
    u256 qx, qy, e, r, s;
    from_be32(qx, in + 160 * k);
    from_be32(qy, in + 160 * k + 32);
    from_be32(e, in + 160 * k + 64);
    from_be32(r, in + 160 * k + 96);
    from_be32(s, in + 160 * k + 128);
    PairQTab<32> qtab = PairQTab<32>::of(qws, k);
    const u256 P = FAB_P256_P;
    const u256 N = FAB_P256_N;
    uint32_t early = range_status(r, s);
    bool q_in_field = lt256(qx, P) & lt256(qy, P);
    fe QX, QY;
    fe_to_mont(QX, qx);
    fe_to_mont(QY, qy);
    bool q_ok = q_in_field & on_curve29(QX, QY);
    if (early == ST_VALID && !q_ok) early = ST_OFF_CURVE;
    u256 w, u1, u2, ered, t;
    {
        const modinv_info NI = MODINV_N_INFO;
        modinv(w, s, NI);
    }
    uint32_t br = sub256(t, e, N);
    sel256(ered, br == 0, t, e);
    fn_to_mont(t, ered);
    fn_mul(u1, t, w);
    fn_to_mont(t, r);
    fn_mul(u2, t, w);
    pair_pt Rr;
    bool r_inf;
    pair_combined_mult29(Rr, r_inf, u1, u2, QX, QY, gtab, qtab, odd);
    fe zz, rm, rhs, rhs_e, d;
    fe_sqr(zz, Rr.B);
    fe_to_mont(rm, r);
    fe_mul(rhs, rm, zz);
    pair_swap_fe(rhs_e, rhs);
    fe_sub(d, Rr.A, rhs_e);
    bool ok1 = fe_is_zero(d);
    uint32_t st = p256_verify_pair29(qx, qy, e, r, s, gtab, qtab, odd);
    int32_t* o = out + threadIdx.x * 48;
    for (int i = 0; i < 8; i++) {
        o[i] = (int32_t)u1.w[i];
        o[8 + i] = (int32_t)u2.w[i];
    }
    for (int i = 0; i < 9; i++) {
        o[16 + i] = Rr.A.v[i];
        o[25 + i] = Rr.B.v[i];
    }
    o[34] = (int32_t)st;
    o[35] = ok1;
    o[36] = r_inf;
    o[37] = (int32_t)early;
    for (int i = 0; i < 9; i++) o[38 + i] = rhs_e.v[i];
}

extern "C" int gputest_pair_verify(const uint8_t* in, int32_t* out) {
    std::vector<int32_t> tab(GTab16::TABLE_WORDS);
    build_g_comb_table16(tab.data());
    uint8_t* din = nullptr;
    int32_t *dtab = nullptr, *dout = nullptr;
    uint4* dws = nullptr;
    if (hipMalloc((void**)&din, 32 * 160) != hipSuccess || hipMalloc((void**)&dtab, sizeof(int32_t) * GTab16::TABLE_WORDS) != hipSuccess ||
        hipMalloc((void**)&dout, 64 * 48 * 4) != hipSuccess || hipMalloc((void**)&dws, (size_t)16 * 8 * 32 * 16) != hipSuccess)
        return -1;
    hipMemcpy(din, in, 32 * 160, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab.data(), sizeof(int32_t) * GTab16::TABLE_WORDS, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gputest_pair_verify_kernel, dim3(1), dim3(64), 0, 0, din, dtab, dws, dout);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    hipMemcpy(out, dout, 64 * 48 * 4, hipMemcpyDeviceToHost);
    hipFree(din); hipFree(dtab); hipFree(dout); hipFree(dws);
    return rc;
}

// The commitment t of the pseudonym-signature equation as the DEVICE computes it (bn_nym29.h), exposed on its own so that tests can
// compare it with vectors made by an independent implementation (tests/golden/idemix_nym_kats.json): one wave, up to 64 signatures
// with one lane each (split == 0), 32 with two lanes each (split == 1) or 16 with four lanes each (split == 2).  in: n x 5 big-endian fields (nym_x, nym_y, c, s_sk, s_rnym);
// out: n x (tx[32] ty[32]) big-endian, st: n status words.
__global__ void __launch_bounds__(64, 1) gputest_nym_commitment_kernel(int split, uint32_t n, const uint8_t* __restrict__ in, const int32_t* __restrict__ hskt,
                                                                        const int32_t* __restrict__ hrandt, uint4* __restrict__ qws, uint8_t* __restrict__ out,
                                                                        uint32_t* __restrict__ st_out) {
    GlobalQTab29<64> qtab = GlobalQTab29<64>::of(qws, threadIdx.x);
    KeyTab8 hsk{hskt}, hrand{hrandt};
    const bool odd = (threadIdx.x & 1u) != 0;
    uint32_t i = split == 2 ? threadIdx.x >> 2 : (split ? threadIdx.x >> 1 : threadIdx.x);
    bool active = i < n;
    uint32_t ic = active ? i : n - 1;
    u256 nx, ny, c, ssk, srn, tx, ty;
    from_be32(nx, in + 160 * ic);
    from_be32(ny, in + 160 * ic + 32);
    from_be32(c, in + 160 * ic + 64);
    from_be32(ssk, in + 160 * ic + 96);
    from_be32(srn, in + 160 * ic + 128);
    uint32_t st;
    if (split == 2) {            // four lanes per signature (bn_quad29.h): 16 signatures on the wave, lane 4k reports
        PairBNQTab pq = PairBNQTab::of(qws, threadIdx.x >> 1);
        bn_nym_quad_half mine;
        bn_nym_quad_part1(mine, odd, (threadIdx.x & 2u) != 0, nx, ny, c, ssk, srn, hskt, hrandt, pq);
        st = bn_nym_quad_part2(tx, ty, mine, odd);
        if (active && (threadIdx.x & 3u) == 0) {
            to_be32(out + 64 * i, tx);
            to_be32(out + 64 * i + 32, ty);
            st_out[i] = st;
        }
        return;
    }
    if (!split) {
        st = bn_nym_commitment29(tx, ty, nx, ny, c, ssk, srn, hsk, hrand, qtab);
    } else {
        bn_nym_half mine;
        bn_nym_split_part1(mine, odd, nx, ny, c, ssk, srn, hsk, hrand, qtab);
        jacbn theirs;
        for (int l = 0; l < 9; l++) {
            theirs.X.v[l] = lane_pair_swap(mine.P.X.v[l]);
            theirs.Y.v[l] = lane_pair_swap(mine.P.Y.v[l]);
            theirs.Z.v[l] = lane_pair_swap(mine.P.Z.v[l]);
        }
        bool theirs_inf = lane_pair_swap(mine.inf ? 1 : 0) != 0;
        st = bn_nym_split_part2(tx, ty, mine, theirs, theirs_inf);
    }
    if (active && (!split || !odd)) {
        to_be32(out + 64 * i, tx);
        to_be32(out + 64 * i + 32, ty);
        st_out[i] = st;
    }
}

extern "C" int gputest_nym_commitment(int split, uint32_t n, const uint8_t* hsk_xy64, const uint8_t* hrand_xy64, const uint8_t* in, uint8_t* out, uint32_t* st) {
    if (n == 0 || n > (split == 2 ? 16u : (split ? 32u : 64u))) return -3;
    std::vector<int32_t> t1(KeyTab8::TABLE_WORDS), t2(KeyTab8::TABLE_WORDS);
    u256 x, y;
    from_be32(x, hsk_xy64); from_be32(y, hsk_xy64 + 32);
    build_bn_comb_table8(t1.data(), x, y);
    from_be32(x, hrand_xy64); from_be32(y, hrand_xy64 + 32);
    build_bn_comb_table8(t2.data(), x, y);
    const size_t tb = sizeof(int32_t) * KeyTab8::TABLE_WORDS;
    int32_t *d1 = nullptr, *d2 = nullptr;
    uint8_t *din = nullptr, *dout = nullptr;
    uint32_t* dst = nullptr;
    uint4* dws = nullptr;
    if (hipMalloc((void**)&d1, tb) != hipSuccess || hipMalloc((void**)&d2, tb) != hipSuccess || hipMalloc((void**)&din, 160 * n) != hipSuccess ||
        hipMalloc((void**)&dout, 64 * n) != hipSuccess || hipMalloc((void**)&dst, 4 * n) != hipSuccess ||
        hipMalloc((void**)&dws, (size_t)16 * 8 * 64 * 16) != hipSuccess)
        return -1;
    hipMemcpy(d1, t1.data(), tb, hipMemcpyHostToDevice);
    hipMemcpy(d2, t2.data(), tb, hipMemcpyHostToDevice);
    hipMemcpy(din, in, 160 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(gputest_nym_commitment_kernel, dim3(1), dim3(64), 0, 0, split, n, din, d1, d2, dws, dout, dst);
    int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    hipMemcpy(out, dout, 64 * n, hipMemcpyDeviceToHost);
    hipMemcpy(st, dst, 4 * n, hipMemcpyDeviceToHost);
    hipFree(d1); hipFree(d2); hipFree(din); hipFree(dout); hipFree(dst); hipFree(dws);
    return rc;
}

// ---- the integer multiply-accumulate ceiling of this chip, SUSTAINED ---------------------------------------------------------------
// What roofline.frac of the verify kernels is priced against (bench.py): every SIMD of the chip issuing nothing but independent
// v_mad_i64_i32 - the instruction the field products are made of - for several milliseconds, at 1, 2 or 4 wavefronts per SIMD.  Short
// bursts (ubench.hip: 40-130 us) run at the boost clock; a kernel that keeps the multiplier array busy for milliseconds runs at
// whatever clock the power budget leaves, and that is the ceiling a 0.7 ms verify launch actually lives under.
// Reports wall time (HIP events), the shader-clock ticks one wavefront counted (s_memtime) and the MACs retired.
__global__ void __launch_bounds__(1024) gputest_mac_ceiling_kernel(uint32_t iters, uint32_t seed, uint64_t* __restrict__ ticks) {
    int32_t a = (int32_t)(seed + threadIdx.x), b = (int32_t)(seed * 3 + 1);
    int64_t x0 = a, x1 = b, x2 = a ^ 77, x3 = threadIdx.x, x4 = a + 1, x5 = b + 2, x6 = a + 3, x7 = b + 4;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; it++) {
#define MAC8 "v_mad_i64_i32 %0, s[10:11], %8, %9, %0\n v_mad_i64_i32 %1, s[10:11], %8, %9, %1\n v_mad_i64_i32 %2, s[10:11], %8, %9, %2\n v_mad_i64_i32 %3, s[10:11], %8, %9, %3\n" \
             "v_mad_i64_i32 %4, s[10:11], %8, %9, %4\n v_mad_i64_i32 %5, s[10:11], %8, %9, %5\n v_mad_i64_i32 %6, s[10:11], %8, %9, %6\n v_mad_i64_i32 %7, s[10:11], %8, %9, %7\n"
        asm volatile(MAC8 MAC8 MAC8 MAC8 MAC8 MAC8 MAC8 MAC8
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                     : "v"(a), "v"(b)
                     : "s10", "s11");
#undef MAC8
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t sink = (uint64_t)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7);
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) + (sink == 0x1234567 ? 1 : 0);
}

// waves_per_simd in {1, 2, 4}; iters x 64 MAC instructions per wavefront.  out: [0] wall ms, [1] mean s_memtime ticks per wavefront,
// [2] MACs retired (lanes x instructions), [3] wavefronts.  0 ok.
extern "C" int gputest_mac_ceiling(int waves_per_simd, uint32_t iters, double* out4) {
    if (!out4 || (waves_per_simd != 1 && waves_per_simd != 2 && waves_per_simd != 4)) return 1;
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 2;
    const int cus = prop.multiProcessorCount, threads = 256 * waves_per_simd, waves = cus * 4 * waves_per_simd;
    uint64_t* d = nullptr;
    if (hipMalloc(&d, sizeof(uint64_t) * waves) != hipSuccess) return 3;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(gputest_mac_ceiling_kernel, dim3(cus), dim3(threads), 0, 0, iters / 16 + 1, 12345u, d);     // warm-up
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gputest_mac_ceiling_kernel, dim3(cus), dim3(threads), 0, 0, iters, 12345u, d);
    hipEventRecord(e1, 0);
    int rc = hipEventSynchronize(e1) == hipSuccess ? 0 : 4;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(waves);
    if (rc == 0 && hipMemcpy(h.data(), d, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost) != hipSuccess) rc = 5;
    double sum = 0;
    for (uint64_t v : h) sum += (double)v;
    out4[0] = ms;
    out4[1] = sum / waves;
    out4[2] = (double)waves * 64.0 * 64.0 * (double)iters;
    out4[3] = waves;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    return rc;
}
