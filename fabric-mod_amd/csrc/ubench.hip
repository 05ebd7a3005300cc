// Instruction-cost micro-benchmarks for the integer big-number path on gfx950.
// Prints cycles per wave-instruction (s_memtime, shader clock) for the ops the P-256 kernel is built from,
// at 1/2/4 waves per SIMD.  Results are recorded in DESIGN.md ("Measured instruction costs").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

constexpr int ITERS = 256;

template <int K>
__global__ void bench(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed ^ 77, d = threadIdx.x;
    uint64_t x0 = a, x1 = b, x2 = c, x3 = d, x4 = a + 1, x5 = b + 2, x6 = c + 3, x7 = d + 4;
    double f0 = a, f1 = b, f2 = 1.0000001, f3 = 3;
    uint64_t t0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (K == 0) {  // 8 independent v_mad_u64_u32 (throughput)
            asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n v_mad_u64_u32 %1, s[10:11], %8, %9, %1\n"
                               "v_mad_u64_u32 %2, s[10:11], %8, %9, %2\n v_mad_u64_u32 %3, s[10:11], %8, %9, %3\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s10", "s11");
        } else if (K == 1) {  // dependent v_mad_u64_u32 chain (latency)
            asm volatile(REP64("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s10", "s11");
        } else if (K == 2) {  // MAC as used: mad ; s_nop 1 ; addc   (one chain)
            asm volatile(REP64("v_mad_u64_u32 %0, vcc, %8, %9, %0\n s_nop 1\n v_addc_co_u32 %10, vcc, 0, %10, vcc\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c) : "vcc");
        } else if (K == 3) {  // 3 interleaved MAC chains, distinct SGPR carries, no nops
            asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n v_mad_u64_u32 %1, s[12:13], %8, %9, %1\n v_mad_u64_u32 %2, s[14:15], %8, %9, %2\n"
                               "v_addc_co_u32 %10, s[10:11], 0, %10, s[10:11]\n v_addc_co_u32 %11, s[12:13], 0, %11, s[12:13]\n v_addc_co_u32 %12, s[14:15], 0, %12, s[14:15]\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed)
                         : "s10", "s11", "s12", "s13", "s14", "s15");
        } else if (K == 4) {  // carry chain: add_co ; s_nop 1 ; addc ; s_nop 1 ...
            asm volatile(REP64("v_addc_co_u32 %10, vcc, %8, %10, vcc\n s_nop 1\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c) : "vcc");
        } else if (K == 5) {  // plain independent 32-bit adds (full-rate reference)
            asm volatile(REP16("v_add_u32 %10, %8, %10\n v_add_u32 %11, %9, %11\n v_add_u32 %12, %8, %12\n v_add_u32 %13, %9, %13\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed), "v"(seed) : "vcc");
        } else if (K == 6) {  // v_mul_lo_u32 independent
            asm volatile(REP16("v_mul_lo_u32 %10, %8, %10\n v_mul_lo_u32 %11, %9, %11\n v_mul_lo_u32 %12, %8, %12\n v_mul_lo_u32 %13, %9, %13\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed), "v"(seed) : "vcc");
        } else if (K == 7) {  // v_mul_hi_u32 independent
            asm volatile(REP16("v_mul_hi_u32 %10, %8, %10\n v_mul_hi_u32 %11, %9, %11\n v_mul_hi_u32 %12, %8, %12\n v_mul_hi_u32 %13, %9, %13\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed), "v"(seed) : "vcc");
        } else if (K == 8) {  // v_mad_u32_u24 independent
            asm volatile(REP16("v_mad_u32_u24 %10, %8, %9, %10\n v_mad_u32_u24 %11, %9, %8, %11\n v_mad_u32_u24 %12, %8, %9, %12\n v_mad_u32_u24 %13, %9, %8, %13\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed), "v"(seed) : "vcc");
        } else if (K == 9) {  // v_fma_f64 independent x4
            asm volatile(REP16("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3\n")
                         : "+v"(f0), "+v"(f1), "+v"(x2), "+v"(x3) : "v"(f2), "v"(f3));
        } else if (K == 10) {  // s_nop 0 x64
            asm volatile(REP64("s_nop 0\n"));
        } else if (K == 11) {  // carry chain without nops, independent pairs interleaved 3-way via SGPRs
            asm volatile(REP16("v_addc_co_u32 %10, s[10:11], %8, %10, s[10:11]\n v_addc_co_u32 %11, s[12:13], %8, %11, s[12:13]\n v_addc_co_u32 %12, s[14:15], %8, %12, s[14:15]\n v_add_u32 %13, %9, %13\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed), "v"(seed)
                         : "s10", "s11", "s12", "s13", "s14", "s15");
        } else if (K == 12) {  // 2 interleaved MAC chains + s_nop 0
            asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n v_mad_u64_u32 %1, s[12:13], %8, %9, %1\n s_nop 0\n"
                               "v_addc_co_u32 %10, s[10:11], 0, %10, s[10:11]\n v_addc_co_u32 %11, s[12:13], 0, %11, s[12:13]\n"
                               "v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n v_mad_u64_u32 %1, s[12:13], %8, %9, %1\n s_nop 0\n"
                               "v_addc_co_u32 %10, s[10:11], 0, %10, s[10:11]\n v_addc_co_u32 %11, s[12:13], 0, %11, s[12:13]\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed)
                         : "s10", "s11", "s12", "s13", "s14", "s15");
        } else if (K == 13) {  // v_mul_u32_u24 + v_mul_hi_u32_u24 pairs
            asm volatile(REP16("v_mul_u32_u24 %10, %8, %10\n v_mul_hi_u32_u24 %11, %9, %11\n v_mul_u32_u24 %12, %8, %12\n v_mul_hi_u32_u24 %13, %9, %13\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(c), "v"(d), "v"(seed), "v"(seed) : "vcc");
        } else if (K == 14) {  // v_lshlrev_b64 / 64-bit add via v_lshl_add_u64
            asm volatile(REP16("v_lshl_add_u64 %0, %1, 0, %0\n v_lshl_add_u64 %2, %3, 0, %2\n v_lshl_add_u64 %4, %5, 0, %4\n v_lshl_add_u64 %6, %7, 0, %6\n")
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
        } else if (K == 15) {  // half wave active? same as K==0 but measured with exec = low 32 lanes (set by caller via threadIdx)
            if (threadIdx.x % 64 < 32)
                asm volatile(REP16("v_mad_u64_u32 %0, s[10:11], %8, %9, %0\n v_mad_u64_u32 %1, s[10:11], %8, %9, %1\n"
                                   "v_mad_u64_u32 %2, s[10:11], %8, %9, %2\n v_mad_u64_u32 %3, s[10:11], %8, %9, %3\n")
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "s10", "s11");
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t sink = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ (uint64_t)c ^ (uint64_t)d ^ (uint64_t)f0 ^ (uint64_t)f1;
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
        {"v_mad_u64_u32 x4 indep", 64, bench<0>}, {"v_mad_u64_u32 dependent", 64, bench<1>},
        {"MAC mad;s_nop1;addc (1 chain)", 64, bench<2>}, {"MAC 3 chains interleaved (per MAC)", 48, bench<3>},
        {"v_addc_co chain + s_nop 1", 64, bench<4>}, {"v_add_u32 indep", 64, bench<5>},
        {"v_mul_lo_u32 indep", 64, bench<6>}, {"v_mul_hi_u32 indep", 64, bench<7>},
        {"v_mad_u32_u24 indep", 64, bench<8>}, {"v_fma_f64 indep", 64, bench<9>}, {"s_nop 0", 64, bench<10>},
        {"addc 3 chains interleaved (per inst of 4)", 64, bench<11>}, {"MAC 2 chains + s_nop0 (per MAC)", 64, bench<12>},
        {"v_mul(_hi)_u32_u24 indep", 64, bench<13>}, {"v_lshl_add_u64 indep", 64, bench<14>},
        {"v_mad_u64_u32 x4, 32 of 64 lanes active", 64, bench<15>},
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
            // s_memtime ticks at a fixed 100 MHz reference on some parts: also report wall-derived cycles at 2.4 GHz
            double wall_cyc = ms * 1e-3 * 2.4e9 / (double)(ITERS * c.insts);
            printf("%-44s memtime/inst %8.3f   wall@2.4GHz/inst(per wave) %8.3f   per-SIMD issue %8.3f\n", c.name, per, wall_cyc, wall_cyc / wps);
        }
    }
    return 0;
}
