// Instruction-cost micro-benchmarks for the integer big-number path on gfx950.
// Prints cycles per wave-instruction for the ops the P-256 kernel is built from, at 1/2/4 waves per SIMD.
// Results are recorded in DESIGN.md ("Measured instruction costs").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
// operands: %0-%7 = x0..x7 (64-bit rw), %8-%11 = y0..y3 (32-bit rw), %12 = a, %13 = b (32-bit inputs)
#define OPS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) \
            : "v"(a), "v"(b) : "vcc", "s10", "s11", "s12", "s13", "s14", "s15"

constexpr int ITERS = 256;

template <int K>
__global__ void bench(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    uint64_t x0 = a, x1 = b, x2 = a ^ 77, x3 = threadIdx.x, x4 = a + 1, x5 = b + 2, x6 = a + 3, x7 = b + 4;
    uint32_t y0 = a ^ 5, y1 = b ^ 6, y2 = seed, y3 = seed + 9;
    double f0 = a, f1 = b, f2 = 1.0000001, f3 = 3, f4 = 5, f5 = 7;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (K == 0) { asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %12, %13, %0\n v_mad_u64_u32 %1, s[10:11], %12, %13, %1\n v_mad_u64_u32 %2, s[10:11], %12, %13, %2\n v_mad_u64_u32 %3, s[10:11], %12, %13, %3\n") OPS); }
        else if (K == 1) { asm volatile(REP64("v_mad_u64_u32 %0, s[10:11], %12, %13, %0\n") OPS); }
        else if (K == 2) { asm volatile(REP64("v_mad_u64_u32 %0, vcc, %12, %13, %0\n s_nop 1\n v_addc_co_u32 %8, vcc, 0, %8, vcc\n") OPS); }
        else if (K == 3) { asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %12, %13, %0\n v_mad_u64_u32 %1, s[12:13], %12, %13, %1\n v_mad_u64_u32 %2, s[14:15], %12, %13, %2\n v_addc_co_u32 %8, s[10:11], 0, %8, s[10:11]\n v_addc_co_u32 %9, s[12:13], 0, %9, s[12:13]\n v_addc_co_u32 %10, s[14:15], 0, %10, s[14:15]\n") OPS); }
        else if (K == 4) { asm volatile(REP64("v_addc_co_u32 %8, vcc, %12, %8, vcc\n s_nop 1\n") OPS); }
        else if (K == 5) { asm volatile(REP16("v_add_u32 %8, %12, %8\n v_add_u32 %9, %13, %9\n v_add_u32 %10, %12, %10\n v_add_u32 %11, %13, %11\n") OPS); }
        else if (K == 6) { asm volatile(REP16("v_mul_lo_u32 %8, %12, %8\n v_mul_lo_u32 %9, %13, %9\n v_mul_lo_u32 %10, %12, %10\n v_mul_lo_u32 %11, %13, %11\n") OPS); }
        else if (K == 7) { asm volatile(REP16("v_mul_hi_u32 %8, %12, %8\n v_mul_hi_u32 %9, %13, %9\n v_mul_hi_u32 %10, %12, %10\n v_mul_hi_u32 %11, %13, %11\n") OPS); }
        else if (K == 8) { asm volatile(REP16("v_mad_u32_u24 %8, %12, %13, %8\n v_mad_u32_u24 %9, %13, %12, %9\n v_mad_u32_u24 %10, %12, %13, %10\n v_mad_u32_u24 %11, %13, %12, %11\n") OPS); }
        else if (K == 9) { asm volatile(REP64("s_nop 0\n") OPS); }
        else if (K == 10) { asm volatile(REP16("v_addc_co_u32 %8, s[10:11], %12, %8, s[10:11]\n v_addc_co_u32 %9, s[12:13], %12, %9, s[12:13]\n v_addc_co_u32 %10, s[14:15], %12, %10, s[14:15]\n v_add_u32 %11, %13, %11\n") OPS); }
        else if (K == 11) { asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %12, %13, %0\n v_mad_u64_u32 %1, s[12:13], %12, %13, %1\n s_nop 0\n v_addc_co_u32 %8, s[10:11], 0, %8, s[10:11]\n v_addc_co_u32 %9, s[12:13], 0, %9, s[12:13]\n v_mad_u64_u32 %0, s[10:11], %12, %13, %0\n v_mad_u64_u32 %1, s[12:13], %12, %13, %1\n s_nop 0\n v_addc_co_u32 %8, s[10:11], 0, %8, s[10:11]\n v_addc_co_u32 %9, s[12:13], 0, %9, s[12:13]\n") OPS); }
        else if (K == 12) { asm volatile(REP16("v_mul_u32_u24 %8, %12, %8\n v_mul_hi_u32_u24 %9, %13, %9\n v_mul_u32_u24 %10, %12, %10\n v_mul_hi_u32_u24 %11, %13, %11\n") OPS); }
        else if (K == 13) { asm volatile(REP16("v_lshl_add_u64 %0, %1, 0, %0\n v_lshl_add_u64 %2, %3, 0, %2\n v_lshl_add_u64 %4, %5, 0, %4\n v_lshl_add_u64 %6, %7, 0, %6\n") OPS); }
        else if (K == 14) { asm volatile(REP64("v_mad_u64_u32 %0, vcc, %12, %13, %0\n v_addc_co_u32 %8, vcc, 0, %8, vcc\n") OPS); }
        else if (K == 15) { asm volatile(REP16("v_add_co_u32 %8, vcc, %12, %8\n v_addc_co_u32 %9, vcc, %13, %9, vcc\n v_addc_co_u32 %10, vcc, %12, %10, vcc\n v_addc_co_u32 %11, vcc, %13, %11, vcc\n") OPS); }
        else if (K == 16) { asm volatile(REP16("v_alignbit_b32 %8, %12, %8, 7\n v_alignbit_b32 %9, %13, %9, 9\n v_alignbit_b32 %10, %12, %10, 11\n v_alignbit_b32 %11, %13, %11, 13\n") OPS); }
        else if (K == 17) { asm volatile(REP16("v_bfi_b32 %8, %12, %8, %13\n v_add3_u32 %9, %13, %9, %12\n v_bfi_b32 %10, %12, %10, %13\n v_add3_u32 %11, %13, %11, %12\n") OPS); }
        else if (K == 18) { asm volatile(REP16("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n") : "+v"(f0), "+v"(f1), "+v"(f4), "+v"(f5) : "v"(f2), "v"(f3)); }
        else if (K == 19) { if ((threadIdx.x & 63) < 32) { asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %12, %13, %0\n v_mad_u64_u32 %1, s[10:11], %12, %13, %1\n v_mad_u64_u32 %2, s[10:11], %12, %13, %2\n v_mad_u64_u32 %3, s[10:11], %12, %13, %3\n") OPS); } }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t sink = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ y0 ^ y1 ^ y2 ^ y3 ^ (uint64_t)f0 ^ (uint64_t)f1 ^ (uint64_t)f4 ^ (uint64_t)f5;
    if (threadIdx.x % 64 == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = (t1 - t0) + (sink == 0x1234567 ? 1 : 0);
}

struct Case { const char* name; int insts; void (*fn)(uint64_t*, uint32_t); };

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    int cus = prop.multiProcessorCount;
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, sizeof(uint64_t) * cus * 64));
    Case cases[] = {
        {"v_mad_u64_u32 x4 indep", 64, bench<0>},
        {"v_mad_u64_u32 dependent", 64, bench<1>},
        {"MAC mad;s_nop1;addc (1 chain)", 64, bench<2>},
        {"MAC 3 chains interleaved (per MAC)", 48, bench<3>},
        {"v_addc_co chain + s_nop 1", 64, bench<4>},
        {"v_add_u32 x4 indep", 64, bench<5>},
        {"v_mul_lo_u32 x4 indep", 64, bench<6>},
        {"v_mul_hi_u32 x4 indep", 64, bench<7>},
        {"v_mad_u32_u24 x4 indep", 64, bench<8>},
        {"s_nop 0", 64, bench<9>},
        {"addc 3 chains interleaved + 1 add (per inst)", 64, bench<10>},
        {"MAC 2 chains + s_nop0 (per MAC)", 64, bench<11>},
        {"v_mul_u32_u24 + v_mul_hi_u32_u24 x2 indep", 64, bench<12>},
        {"v_lshl_add_u64 x4 indep", 64, bench<13>},
        {"MAC without nop, 1 chain (hazard-unsafe timing probe)", 64, bench<14>},
        {"v_add_co/v_addc_co pair chain no nop (timing probe)", 64, bench<15>},
        {"v_alignbit_b32 x4 indep", 64, bench<16>},
        {"v_bfi/v_add3 mix x4 indep", 64, bench<17>},
        {"v_fma_f64 x4 indep", 64, bench<18>},
        {"v_mad_u64_u32 x4 indep, lanes 0-31 only", 64, bench<19>},
    };
    for (int wps : {1, 2, 4}) {
        printf("--- %d wave(s) per SIMD (block = %d threads, 1 block per CU) ---\n", wps, 256 * wps);
        for (auto& c : cases) {
            hipLaunchKernelGGL(c.fn, dim3(cus), dim3(256 * wps), 0, 0, d_out, 12345u);
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(c.fn, dim3(cus), dim3(256 * wps), 0, 0, d_out, 12345u);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> h(cus * 4 * wps);
            CHECK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
            double avg = 0; for (auto v : h) avg += v; avg /= h.size();
            double per = avg / (double)(ITERS * c.insts);
            printf("%-56s memtime-ticks/inst %8.3f | wall-us %8.1f | ns/inst/wave %7.3f | ns/inst per SIMD slot %7.3f\n", c.name, per, ms * 1e3,
                   ms * 1e6 / (double)(ITERS * c.insts), ms * 1e6 / (double)(ITERS * c.insts) / wps);
        }
    }
    return 0;
}
