// Kernels of the eight-lanes-per-signature keyed verification (p256_wide29.h) and the hash-only kernel that feeds its second phase.
// A TU of its own: the point arithmetic is inlined generated asm, and kernels.hip is large enough.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_common.h"
#include "kernels.h"
#include "p256_wide29.h"
#include "sha256_coop.h"

namespace fab {

// One wavefront per workgroup: a 100-transaction block is 50 wavefronts - each gets a CU's SIMD to itself wherever the dispatcher
// puts it; 8 192 signatures are 1 024, one per SIMD.
constexpr int WIDE_BLOCK = 64;

template <int W>
__global__ void __launch_bounds__(64 * W, 1) p256_wide_pre_kernel(uint32_t n, const uint32_t* __restrict__ key_id, uint32_t nkeys,
                                                                      const int32_t* const* __restrict__ ktabs, const uint8_t* __restrict__ r,
                                                                      const uint8_t* __restrict__ s, const int32_t* __restrict__ gtab,
                                                                      int32_t* __restrict__ scratch) {
    GTab16 gt{gtab};
    constexpr uint32_t PER = WIDE_BLOCK / WIDE_LANES;
    const uint32_t lane = threadIdx.x & 63u, waves = blockDim.x >> 6;      // a tile = one wavefront's eight signatures
    const uint32_t sub = lane & (WIDE_LANES - 1);
    const uint32_t ntiles = (n + PER - 1) / PER;
    for (uint32_t tile = blockIdx.x * waves + (threadIdx.x >> 6); tile < ntiles; tile += gridDim.x * waves) {
        const uint32_t i = tile * PER + (lane >> 3);
        const bool active = i < n;
        const uint32_t ic = active ? i : (n - 1);
        const uint32_t kid = key_id[ic];
        const bool kok = kid < nkeys;
        KeyTab8 kt{ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0)]};
        u256 vr, vs;
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        p256_wide_pre29(vr, vs, gt, kt, sub, kok, active && sub == 0, scratch + (size_t)WIDE_SCRATCH_WORDS * ic);
    }
}

// e: 32-byte big-endian digests by row.  verdict8: one byte per eight signatures (bit k of byte j = signature 8 j + k), i.e. the byte
// view of the usual verdict words.
template <int W>
__global__ void __launch_bounds__(64 * W, 1) p256_wide_post_kernel(uint32_t n, const uint8_t* __restrict__ e, const uint8_t* __restrict__ r,
                                                                       const int32_t* __restrict__ gtab, const int32_t* __restrict__ scratch,
                                                                       uint8_t* __restrict__ verdict8, uint8_t* __restrict__ status) {
    GTab16 gt{gtab};
    constexpr uint32_t PER = WIDE_BLOCK / WIDE_LANES;
    const uint32_t lane = threadIdx.x & 63u, waves = blockDim.x >> 6;
    const uint32_t sub = lane & (WIDE_LANES - 1);
    const uint32_t ntiles = (n + PER - 1) / PER;
    for (uint32_t tile = blockIdx.x * waves + (threadIdx.x >> 6); tile < ntiles; tile += gridDim.x * waves) {
        const uint32_t i = tile * PER + (lane >> 3);
        const bool active = i < n;
        const uint32_t ic = active ? i : (n - 1);
        u256 ve, vr;
        load_be_field(ve, e, ic);
        load_be_field(vr, r, ic);
        const uint32_t st = p256_wide_post29(ve, vr, gt, sub, scratch + (size_t)WIDE_SCRATCH_WORDS * ic);
        // eight verdicts per wavefront, on lanes 0, 8, .. 56: squeeze them into one byte
        uint64_t x = __ballot(active && sub == 0 && st == 0u) & 0x0101010101010101ull;
        x = (x | (x >> 7)) & 0x0003000300030003ull;
        x = (x | (x >> 14)) & 0x0000000f0000000full;
        x = (x | (x >> 28)) & 0xffull;
        if (lane == 0) {
            verdict8[tile] = (uint8_t)x;       // (PER == 8: tile j covers signatures 8 j .. 8 j + 7)
            if (tile == ntiles - 1)            // the rest of the last 64-bit verdict word: nobody's signatures
                for (uint32_t b = ntiles; b < ((n + 63) / 64) * 8; b++) verdict8[b] = 0;
        }
        if (status != nullptr && active && sub == 0) status[i] = (uint8_t)st;
    }
}

// SHA-256 of n (possibly prefixed) messages -> n x 32 digest bytes (pre.digests): the hash half of the fused kernels alone, one
// message per lane; the wide verification takes digests from memory because its first phase runs beside this kernel, not behind it.
__global__ void __launch_bounds__(64) sha256_messages_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                             const uint32_t* __restrict__ off, sha_prefixes pre) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n;
    const uint32_t ic = active ? i : (n - 1);
    uint32_t h[8];
    sha256_message(arena32, arena_words, off, pre, ic, active, h);
    emit_digest(pre, i, active, h);
}

// The same digests with eight lanes on a message (sha256_coop.h): no mid-states - a prefixed message is hashed whole, prefix first.
template <int W>
__global__ void __launch_bounds__(64 * W) sha256_messages_coop_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                                  const uint32_t* __restrict__ off, sha_prefixes pre) {
    extern __shared__ uint32_t lds_all[];                                  // SHAC_LDS_WORDS per wavefront, or more (placement: see PLACEMENT below)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* lds = lds_all + wave * SHAC_LDS_WORDS;
    const uint32_t i = (blockIdx.x * (blockDim.x >> 6) + wave) * SHAC_PER_WAVE + (lane >> 3);
    const bool active = i < n;
    const uint32_t ic = active ? i : (n - 1);
    const uint32_t sb = off[pre.spans ? 2 * ic : ic], se = off[pre.spans ? 2 * ic + 1 : ic + 1];
    uint32_t ps = 0, pl = 0;
    if (pre.pre_idx != nullptr) {
        const uint32_t pi = pre.pre_idx[ic];
        if (pi < pre.m) {
            ps = pre.pre_off[pre.spans ? 2 * pi : pi];
            const uint32_t pe = pre.pre_off[pre.spans ? 2 * pi + 1 : pi + 1];
            pl = pe >= ps ? pe - ps : 0;
        }
    }
    uint32_t h[8];
    sha256_coop(arena32, arena_words, ps, pl, sb, se >= sb ? se - sb : 0, active, lds, lane, h);
    if (active && (lane & (SHAC_LANES - 1)) == 0 && pre.digests != nullptr) sha256_coop_store(pre.digests, i, h);
}

// A launch too large for eight lanes per message: one lane per message - but not for its LONG messages.  The launch lasts as long as
// its longest message (a wavefront runs until its last lane is through), and one creator with a 5 KB certificate makes three of a
// block's messages that long - its payload, its TxID check (nonce || creator), an endorsement if it endorses - which held a
// 10 000-transaction block's hash checks for 310 us instead of 115 (round 4: the `one_oversize_identity` leg, 1.09x a friendly block).
// So the messages of more than `long_over` bytes are taken out and hashed on eight lanes each: the first `scan_wgs` workgroups look at
// 64 messages per wavefront - a lane each - and hash the first EIGHT long ones among them (none, nearly always: they leave after one
// ballot; one round at most, so that a launch whose messages are all "long" by the launcher's guess loses nothing); the workgroups
// behind them are the ordinary lane-per-message ones, cut into the same groups of 64, and skip exactly those.  One launch: the long
// ones start first.
__global__ void __launch_bounds__(256) sha256_mixed_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                           const uint32_t* __restrict__ off, uint32_t pairs, uint32_t long_over, uint32_t scan_wgs,
                                                           uint32_t* __restrict__ digests) {
    extern __shared__ uint32_t lds_all[];                                  // SHAC_LDS_WORDS per wavefront of the workgroup
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    auto span = [&](uint32_t i, uint32_t& start, uint32_t& len) {
        start = off[pairs ? 2 * i : i];
        const uint32_t end = off[pairs ? 2 * i + 1 : i + 1];
        len = end >= start ? end - start : 0;
    };
    if (blockIdx.x < scan_wgs) {
        uint32_t* lds = lds_all + wave * SHAC_LDS_WORDS;
        const uint32_t i0 = (blockIdx.x * waves + wave) * 64u, mine = i0 + lane;
        uint32_t st0 = 0, ln0 = 0;
        if (mine < n) span(mine, st0, ln0);
        uint64_t longs = __ballot(mine < n && ln0 > long_over);
        if (longs == 0ull) return;
        uint32_t pick = 64;
        for (uint32_t g = 0; g < (uint32_t)SHAC_PER_WAVE && longs; g++) {    // (wavefront-uniform) the first eight, one per lane group
            const uint32_t b = (uint32_t)__builtin_ctzll(longs);
            longs &= longs - 1;
            if ((lane >> 3) == g) pick = b;
        }
        const bool active = pick < 64;
        const uint32_t i = i0 + (active ? pick : 0u);
        uint32_t start = 0, len = 0;
        if (active) span(i, start, len);
        uint32_t h[8];
        sha256_coop(arena32, arena_words, 0, 0, start, len, active, lds, lane, h);
        if (active && (lane & (SHAC_LANES - 1)) == 0) sha256_coop_store(digests, i, h);
        return;
    }
    const uint32_t i = (blockIdx.x - scan_wgs) * blockDim.x + threadIdx.x;
    uint32_t start = 0, len = 0;
    if (i < n) span(i, start, len);
    // (this wavefront's 64 messages are the 64 a scan wavefront looked at: the first eight long ones are that one's)
    const uint64_t longs = __ballot(i < n && len > long_over);
    const bool taken = i < n && len > long_over && __builtin_popcountll(longs & ((1ull << lane) - 1ull)) < SHAC_PER_WAVE;
    const bool active = i < n && !taken;
    uint32_t h[8];
    sha256_lane<true>(arena32, arena_words, start, len, active, h);
    if (active) sha256_coop_store(digests, i, h);
}

// PLACEMENT of the small launches (the wide kernels, the eight-lane hashes): their wavefronts are long serial instruction streams, a
// SIMD runs one such stream at full speed and two at half each, and the dispatcher fills a CU as long as a workgroup fits - dozens of
// one-wavefront workgroups, on whichever SIMDs.  Two levers, both measured (tools/gpu_probe_small2.py, device phase of a pass):
//   * workgroups of SEVERAL wavefronts - a workgroup's wavefronts go to different SIMDs of its CU: two per workgroup for the wide
//     kernels (four measured the same, within noise), four for the hashes: 500 tx 0.498 -> 0.406 ms, 1 000 tx
//     0.599 -> 0.562 (same box, same call; 100 tx 0.345 -> 0.328);
//   * unused dynamic LDS, so that a CU takes no more wavefronts of these launches than it should: `cap` wavefronts per CU = what
//     runs at the same time over all the launches that do (a pass tells: its `pre`, its hashes and its hash checks run side by side)
//     / 256 CUs, rounded up to 1, 2, 4 or 8; a workgroup of W wavefronts asks for W / cap of the CU's 160 KB.  One workgroup per CU
//     while that is possible (100 tx: 0.35 -> 0.30 ms against four one-wavefront workgroups per CU), one wavefront per SIMD up to
//     1 024 of them, two up to 2 048, nothing beyond.
uint32_t spread_waves_per_cu(uint32_t wavefronts) {
    const uint32_t per_cu = (wavefronts + 255) / 256;
    return per_cu <= 1 ? 1u : (per_cu <= 2 ? 2u : (per_cu <= 4 ? 4u : (per_cu <= 8 ? 8u : SPREAD_NONE)));
}
// the dynamic LDS a launch of W-wavefront workgroups asks for under `cap` (0: from its own wavefront count)
// (static_lds: what the kernel's workgroup holds anyway - the wide kernels keep 2 KB per wavefront - so that the sum stays inside a CU)
static uint32_t spread_bytes(uint32_t cap, uint32_t own_wavefronts, uint32_t W, uint32_t static_lds) {
    if (cap == 0) cap = spread_waves_per_cu(own_wavefronts);
    if (cap == SPREAD_NONE) return 0;
    const uint32_t wgs_per_cu = cap > W ? cap / W : 1u;
    const uint32_t room = ((160u << 10) / wgs_per_cu) - (4u << 10);       // 1 per CU: 156 KB, 2: 76 KB, 4: 36 KB, 8: 16 KB
    return room > static_lds ? room - static_lds : 0u;
}
constexpr uint32_t WIDE_WG_WAVES = 2, COOP_WG_WAVES = 4, WIDE_STATIC_LDS_PER_WAVE = 2048;

static_assert(WIDE_BLOCK / WIDE_LANES == 8, "one verdict byte per tile");
static_assert(WIDE_LAUNCH_MAX == WIDE_MAX && WIDE_SCRATCH_BYTES == 4 * WIDE_SCRATCH_WORDS, "kernels.h restates p256_wide29.h for the host");

hipError_t launch_p256_wide_pre(uint32_t n, const void* key_id, uint32_t nkeys, const void* ktabs, const void* r, const void* s, const void* gtab,
                                void* scratch, hipStream_t st, uint32_t lds_spread) {
    if (n == 0) return hipSuccess;
    const uint32_t tiles = (n + 7) / 8, W = tiles >= WIDE_WG_WAVES ? WIDE_WG_WAVES : 1u, wgs = (tiles + W - 1) / W;
    dim3 grid(wgs < 4096u ? wgs : 4096u), block(64 * W);
    auto k = W == 2 ? p256_wide_pre_kernel<2> : p256_wide_pre_kernel<1>;
    hipLaunchKernelGGL(k, grid, block, spread_bytes(lds_spread, tiles, W, W * WIDE_STATIC_LDS_PER_WAVE), st, n, (const uint32_t*)key_id, nkeys, (const int32_t* const*)ktabs, (const uint8_t*)r,
                       (const uint8_t*)s, (const int32_t*)gtab, (int32_t*)scratch);
    return hipGetLastError();
}
hipError_t launch_p256_wide_post(uint32_t n, const void* e, const void* r, const void* gtab, const void* scratch, void* verdict_bits, void* status,
                                 hipStream_t st, uint32_t lds_spread) {
    if (n == 0) return hipSuccess;
    const uint32_t tiles = (n + 7) / 8, W = tiles >= WIDE_WG_WAVES ? WIDE_WG_WAVES : 1u, wgs = (tiles + W - 1) / W;
    dim3 grid(wgs < 4096u ? wgs : 4096u), block(64 * W);
    auto k = W == 2 ? p256_wide_post_kernel<2> : p256_wide_post_kernel<1>;
    hipLaunchKernelGGL(k, grid, block, spread_bytes(lds_spread, tiles, W, W * WIDE_STATIC_LDS_PER_WAVE), st, n, (const uint8_t*)e, (const uint8_t*)r, (const int32_t*)gtab, (const int32_t*)scratch,
                       (uint8_t*)verdict_bits, (uint8_t*)status);
    return hipGetLastError();
}
hipError_t launch_sha256_messages(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const ShaPrefixArgs& pa, hipStream_t st) {
    if (n == 0) return hipSuccess;
    sha_prefixes pre;
    pre.pre_idx = (pa.m && pa.pre_idx) ? (const uint32_t*)pa.pre_idx : nullptr;
    pre.pre_off = (const uint32_t*)pa.pre_off;
    pre.mid = (const uint32_t*)pa.mid_scratch;
    pre.m = pre.pre_idx ? pa.m : 0;
    pre.spans = pa.spans ? 1u : 0u;
    pre.digests = (uint32_t*)pa.digests;
    dim3 grid((n + 63) / 64), block(64);
    hipLaunchKernelGGL(sha256_messages_kernel, grid, block, 0, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4), (const uint32_t*)off, pre);
    return hipGetLastError();
}
hipError_t launch_sha256_mixed(uint32_t n, const void* arena, size_t arena_bytes, const void* off, bool pairs, void* digests, hipStream_t st,
                               uint32_t lds_reserve) {
    if (n == 0) return hipSuccess;
    // (a launch that keeps CUs to itself - lds_reserve, ShaPrefixArgs - brings four wavefronts per workgroup, one per SIMD of its CU)
    const uint32_t block = lds_reserve ? 256u : 64u, wgs = (n + block - 1) / block;
    const uint32_t own = (block / 64u) * (uint32_t)SHAC_LDS_WORDS * 4u;
    // "long": a quarter above what the messages average if they fill their arena (they do: a block's payloads, a gather scratch), and at
    // least 2 KiB.  A friendly 10 000-transaction block: payloads of 5.0 KB, threshold 6.3 KB; its hash checks: 1.3 KB, threshold 2 KiB.
    const uint64_t mean = arena_bytes / n;
    const uint32_t long_over = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, std::max<uint64_t>(2048, mean + mean / 4));
    hipLaunchKernelGGL(sha256_mixed_kernel, dim3(2 * wgs), dim3(block), lds_reserve > own ? lds_reserve : own, st, n, (const uint32_t*)arena,
                       (uint32_t)((arena_bytes + 3) / 4), (const uint32_t*)off, pairs ? 1u : 0u, long_over, wgs, (uint32_t*)digests);
    return hipGetLastError();
}
hipError_t launch_sha256_messages_coop(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const ShaPrefixArgs& pa, hipStream_t st,
                                       uint32_t lds_spread) {
    if (n == 0) return hipSuccess;
    sha_prefixes pre;
    pre.pre_idx = (pa.m && pa.pre_idx) ? (const uint32_t*)pa.pre_idx : nullptr;
    pre.pre_off = (const uint32_t*)pa.pre_off;
    pre.mid = nullptr;
    pre.m = pre.pre_idx ? pa.m : 0;
    pre.spans = pa.spans ? 1u : 0u;
    pre.digests = (uint32_t*)pa.digests;
    const uint32_t nwaves = (n + SHAC_PER_WAVE - 1) / SHAC_PER_WAVE, W = nwaves >= COOP_WG_WAVES ? COOP_WG_WAVES : 1u;
    dim3 grid((nwaves + W - 1) / W), block(64 * W);
    const uint32_t want = spread_bytes(lds_spread, nwaves, W, 0), own = W * (uint32_t)SHAC_LDS_WORDS * 4;
    const uint32_t lds = want > own ? want : own;
    auto k = W == 4 ? sha256_messages_coop_kernel<4> : sha256_messages_coop_kernel<1>;
    hipLaunchKernelGGL(k, grid, block, lds, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4), (const uint32_t*)off, pre);
    return hipGetLastError();
}

// see warm_kernel_functions_kernels (kernels.hip)
int warm_kernel_functions_wide() {
    int ok = 0;
    hipFuncAttributes a;
    const void* fns[] = {(const void*)p256_wide_pre_kernel<1>, (const void*)p256_wide_pre_kernel<2>, (const void*)p256_wide_post_kernel<1>,
                         (const void*)p256_wide_post_kernel<2>, (const void*)sha256_messages_kernel, (const void*)sha256_messages_coop_kernel<1>,
                         (const void*)sha256_messages_coop_kernel<4>, (const void*)sha256_mixed_kernel};
    for (const void* f : fns) ok += hipFuncGetAttributes(&a, f) == hipSuccess ? 1 : 0;
    return ok;
}

}  // namespace fab
