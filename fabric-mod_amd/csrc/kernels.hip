// HIP kernels for gfx950 (MI355X / CDNA4): batched SHA-256 (with shared-prefix mid-states), batched ECDSA P-256 verify -
// one signature per lane, or two lanes per signature for batches that cannot fill the chip - for fresh and for registered
// public keys, and the fused hash+verify kernels.  64-lane wavefronts, verdicts packed with a wave ballot.  No MFMA: this is
// integer VALU work (v_mad_i64_i32 on 29-bit signed limbs for the big-number part - fe29.h, generated streams in fe29_gcn.h /
// pair29_gcn.h - and v_alignbit / v_xor / v_add for SHA-256).
//
// Replaces (reference, all CPU): bccsp/sw/hash.go:29-33, bccsp/sw/ecdsa.go:41-57 -> crypto/ecdsa.Verify,
// called per signature from msp/identities.go:169-196.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "device_common.h"
#include "kernels.h"
#include "p256_pair29.h"

namespace fab {

// (SHA-256 per lane, field loads, verdict packing and the per-lane table workspace: device_common.h)

__global__ void __launch_bounds__(256) sha256_midstate_kernel(uint32_t m, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                               const uint32_t* __restrict__ pre_off, uint32_t spans, uint32_t* __restrict__ mid) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = p < m;
    uint32_t pc = active ? p : (m - 1);
    uint32_t start = pre_off[spans ? 2 * pc : pc], len = pre_off[spans ? 2 * pc + 1 : pc + 1] - start;
    uint32_t nfull = active ? (len >> 6) : 0, maxfull = nfull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t other = __shfl_xor(maxfull, o, 64);
        maxfull = other > maxfull ? other : maxfull;
    }
    maxfull = __builtin_amdgcn_readfirstlane(maxfull);
    uint32_t h[8];
    sha256_iv(h);
    const uint32_t shift = start & 3u;
    const uint32_t last_word = arena_words ? arena_words - 1 : 0;
    uint32_t nxt[17];
    auto fetch = [&](uint32_t blk_, uint32_t (&dst)[17]) {
        const uint32_t wi = (start + (blk_ << 6)) >> 2;
#pragma unroll
        for (int k = 0; k < 17; k++) {
            uint32_t idx = wi + k;
            idx = idx < last_word ? idx : last_word;
            dst[k] = arena32[idx];
        }
    };
    if (maxfull) fetch(0, nxt);
    for (uint32_t blk = 0; blk < maxfull; blk++) {             // (the next block's loads are in flight while this one is compressed)
        uint32_t w[16], raw[17];
#pragma unroll
        for (int k = 0; k < 17; k++) raw[k] = nxt[k];
        if (blk + 1 < maxfull) fetch(blk + 1, nxt);
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = __builtin_bswap32(__builtin_amdgcn_alignbyte(raw[k + 1], raw[k], shift));
        if (blk < nfull) sha256_compress(h, w);
    }
    if (active) {
        uint4* o = reinterpret_cast<uint4*>(mid + 8 * (size_t)p);
        o[0] = make_uint4(h[0], h[1], h[2], h[3]);
        o[1] = make_uint4(h[4], h[5], h[6], h[7]);
    }
}
// (hash-only launches: wide_kernels.hip - eight lanes per message for small ones, sha256_mixed_kernel beyond)

// ------------------------------------------------------------------------------------------------
// ECDSA P-256 verify
// ------------------------------------------------------------------------------------------------
// Persistent workgroups: a bounded number of slots, each walking the BLOCK-signature tiles  blockIdx.x, blockIdx.x + gridDim.x, ...
// (the per-lane j*Q workspace is sized by slots, not by the batch).
// One 256-thread workgroup per CU = one wave per SIMD.  A second wave per SIMD was measured (BLOCK = 512, round-1 PMC runs in
// profiles/): each wave then takes 1.5x the cycles, but the chip also drops from ~2.0 to ~1.6 GHz - the integer multiplier
// array is power-limited - so whole-job throughput does not move; one wave per SIMD keeps the latency of a block minimal.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2) p256_verify_kernel(uint32_t n, const uint8_t* __restrict__ qx, const uint8_t* __restrict__ qy,
                                                                    const uint8_t* __restrict__ e, const uint8_t* __restrict__ r,
                                                                    const uint8_t* __restrict__ s, const int32_t* __restrict__ gtab,
                                                                    uint4* __restrict__ qws, uint64_t* __restrict__ verdict_bits,
                                                                    uint8_t* __restrict__ status) {
    GlobalQTab29<BLOCK> qtab = GlobalQTab29<BLOCK>::of(qws + (size_t)blockIdx.x * (QWS_UINT4_PER_LANE * BLOCK), threadIdx.x);
    GTab16 gt{gtab};
    const uint32_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * BLOCK + threadIdx.x;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        u256 vqx, vqy, ve, vr, vs;
        load_be_field(vqx, qx, ic);
        load_be_field(vqy, qy, ic);
        load_be_field(ve, e, ic);
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = p256_verify_core29(vqx, vqy, ve, vr, vs, gt, qtab);
        emit_verdict(i, active, st, verdict_bits, status);
    }
}

// Two lanes per signature (p256_pair29.h): 128 signatures per 256-thread workgroup, for batches that cannot fill the chip
// with one signature per lane.  Lane 2k / 2k+1 of a wave own signature k; the even lane carries the verdict.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 1) p256_verify_pair_kernel(uint32_t n, const uint8_t* __restrict__ qx, const uint8_t* __restrict__ qy,
                                                                         const uint8_t* __restrict__ e, const uint8_t* __restrict__ r,
                                                                         const uint8_t* __restrict__ s, const int32_t* __restrict__ gtab,
                                                                         uint4* __restrict__ qws, uint64_t* __restrict__ verdict_bits,
                                                                         uint8_t* __restrict__ status) {
    constexpr int NP = BLOCK / 2;
    const bool odd = (threadIdx.x & 1) != 0;
    const uint32_t pairidx = threadIdx.x >> 1;
    PairQTab<NP> qtab = PairQTab<NP>::of(qws + (size_t)blockIdx.x * (QWS_PAIR_UINT4_PER_SIG * NP), pairidx);
    uint32_t* verdict32 = reinterpret_cast<uint32_t*>(verdict_bits);
    const uint32_t ntiles = (n + NP - 1) / NP;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * NP + pairidx;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        u256 vqx, vqy, ve, vr, vs;
        load_be_field(vqx, qx, ic);
        load_be_field(vqy, qy, ic);
        load_be_field(ve, e, ic);
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = p256_verify_pair29(vqx, vqy, ve, vr, vs, gtab, qtab, odd);
        pair_emit_verdict(i, n, active, odd, st, verdict32, status);
    }
}

// The same with the per-signature table in LDS (PairQTabLds: 8 entries, signed 4-bit windows) instead of the global workspace: no
// table traffic at all (p256_pair29.h says what that is worth).  Dynamic LDS: 128 signatures x 1040 bytes.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 1) p256_verify_pair_lds_kernel(uint32_t n, const uint8_t* __restrict__ qx, const uint8_t* __restrict__ qy,
                                                                             const uint8_t* __restrict__ e, const uint8_t* __restrict__ r,
                                                                             const uint8_t* __restrict__ s, const int32_t* __restrict__ gtab,
                                                                             uint64_t* __restrict__ verdict_bits, uint8_t* __restrict__ status) {
    extern __shared__ uint4 pair_lds[];
    constexpr int NP = BLOCK / 2;
    const bool odd = (threadIdx.x & 1) != 0;
    const uint32_t pairidx = threadIdx.x >> 1;
    PairQTabLds qtab = PairQTabLds::of(pair_lds, pairidx);
    uint32_t* verdict32 = reinterpret_cast<uint32_t*>(verdict_bits);
    const uint32_t ntiles = (n + NP - 1) / NP;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * NP + pairidx;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        u256 vqx, vqy, ve, vr, vs;
        load_be_field(vqx, qx, ic);
        load_be_field(vqy, qy, ic);
        load_be_field(ve, e, ic);
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = p256_verify_pair29<PairQTabLds, 4>(vqx, vqy, ve, vr, vs, gtab, qtab, odd);
        pair_emit_verdict(i, n, active, odd, st, verdict32, status);
    }
}

// Registered public keys (fabgpu_p256_key_register): every signature names a key whose 8-bit comb table is resident on the
// device, so u2*Q is 32 mixed additions like u1*G: no doublings, no per-lane table, no workspace.  ktabs[KTAB_STRIDE k] = table of key k,
// ktabs[KTAB_STRIDE k + 1] = its 16-bit comb or nullptr (round 6, FABGPU_FLAG_KEY_TABLES_16BIT: 16 mixed additions when a whole wavefront has them).
// An out-of-range key id reports status 4 ("use bccsp/sw"), never a verdict.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2) p256_verify_keyed_kernel(uint32_t n, const uint32_t* __restrict__ key_id, uint32_t nkeys,
                                                                          const int32_t* const* __restrict__ ktabs, const uint8_t* __restrict__ e,
                                                                          const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                                                                          const int32_t* __restrict__ gtab, uint64_t* __restrict__ verdict_bits,
                                                                          uint8_t* __restrict__ status) {
    GTab16 gt{gtab};
    const uint32_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * BLOCK + threadIdx.x;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        uint32_t kid = key_id[ic];
        bool kok = kid < nkeys;
        KeyTab8 kt{ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0)]};
        const int32_t* kt16 = ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0) + 1];
        u256 ve, vr, vs;
        load_be_field(ve, e, ic);
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = __all(kt16 != nullptr) ? p256_verify_keyed_core29(ve, vr, vs, gt, GTab16{kt16}) : p256_verify_keyed_core29(ve, vr, vs, gt, kt);
        if (!kok) st = ST_OFF_CURVE;
        emit_verdict(i, active, st, verdict_bits, status);
    }
}
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2) p256_verify_keyed_pair_kernel(uint32_t n, const uint32_t* __restrict__ key_id, uint32_t nkeys,
                                                                               const int32_t* const* __restrict__ ktabs, const uint8_t* __restrict__ e,
                                                                               const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                                                                               const int32_t* __restrict__ gtab, uint64_t* __restrict__ verdict_bits,
                                                                               uint8_t* __restrict__ status) {
    constexpr int NP = BLOCK / 2;
    const bool odd = (threadIdx.x & 1) != 0;
    const uint32_t pairidx = threadIdx.x >> 1;
    uint32_t* verdict32 = reinterpret_cast<uint32_t*>(verdict_bits);
    const uint32_t ntiles = (n + NP - 1) / NP;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * NP + pairidx;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        uint32_t kid = key_id[ic];
        bool kok = kid < nkeys;
        const int32_t* kt = ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0)];
        const int32_t* kt16 = ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0) + 1];
        u256 ve, vr, vs;
        load_be_field(ve, e, ic);
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = p256_verify_keyed_pair29(ve, vr, vs, gtab, kt, kt16, odd);
        if (!kok) st = ST_OFF_CURVE;
        pair_emit_verdict(i, n, active, odd, st, verdict32, status);
    }
}

// identity.Verify fused, registered keys: SHA-256 then the keyed core; the digest stays in registers.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2) sha256_p256_verify_keyed_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                                                 const uint32_t* __restrict__ off, const uint32_t* __restrict__ key_id,
                                                                                 uint32_t nkeys, const int32_t* const* __restrict__ ktabs,
                                                                                 const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                                                                                 const int32_t* __restrict__ gtab, uint64_t* __restrict__ verdict_bits,
                                                                                 uint8_t* __restrict__ status, sha_prefixes pre) {
    GTab16 gt{gtab};
    const uint32_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * BLOCK + threadIdx.x;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        uint32_t h[8];
        sha256_message(arena32, arena_words, off, pre, ic, active, h);
        emit_digest(pre, i, active, h);
        uint32_t kid = key_id[ic];
        bool kok = kid < nkeys;
        KeyTab8 kt{ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0)]};
        const int32_t* kt16 = ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0) + 1];
        u256 ve, vr, vs;
#pragma unroll
        for (int k = 0; k < 8; k++) ve.w[k] = h[7 - k];
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = __all(kt16 != nullptr) ? p256_verify_keyed_core29(ve, vr, vs, gt, GTab16{kt16}) : p256_verify_keyed_core29(ve, vr, vs, gt, kt);
        if (!kok) st = ST_OFF_CURVE;
        emit_verdict(i, active, st, verdict_bits, status);
    }
}
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2) sha256_p256_verify_keyed_pair_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                                                      const uint32_t* __restrict__ off, const uint32_t* __restrict__ key_id,
                                                                                      uint32_t nkeys, const int32_t* const* __restrict__ ktabs,
                                                                                      const uint8_t* __restrict__ r, const uint8_t* __restrict__ s,
                                                                                      const int32_t* __restrict__ gtab, uint64_t* __restrict__ verdict_bits,
                                                                                      uint8_t* __restrict__ status, sha_prefixes pre) {
    constexpr int NP = BLOCK / 2;
    const bool odd = (threadIdx.x & 1) != 0;
    const uint32_t pairidx = threadIdx.x >> 1;
    uint32_t* verdict32 = reinterpret_cast<uint32_t*>(verdict_bits);
    const uint32_t ntiles = (n + NP - 1) / NP;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * NP + pairidx;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        uint32_t h[8];
        sha256_message(arena32, arena_words, off, pre, ic, active, h);
        emit_digest(pre, i, active && !odd, h);
        uint32_t kid = key_id[ic];
        bool kok = kid < nkeys;
        const int32_t* kt = ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0)];
        const int32_t* kt16 = ktabs[KTAB_STRIDE * (size_t)(kok ? kid : 0) + 1];
        u256 ve, vr, vs;
#pragma unroll
        for (int k = 0; k < 8; k++) ve.w[k] = h[7 - k];
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = p256_verify_keyed_pair29(ve, vr, vs, gtab, kt, kt16, odd);
        if (!kok) st = ST_OFF_CURVE;
        pair_emit_verdict(i, n, active, odd, st, verdict32, status);
    }
}

// identity.Verify fused, two lanes per signature: both lanes of a pair hash the (same) message - the hash is 18 % of the
// stream and does not split across lanes - and keep the digest in registers.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 1) sha256_p256_verify_pair_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                                                const uint32_t* __restrict__ off, const uint8_t* __restrict__ qx,
                                                                                const uint8_t* __restrict__ qy, const uint8_t* __restrict__ r,
                                                                                const uint8_t* __restrict__ s, const int32_t* __restrict__ gtab,
                                                                                uint4* __restrict__ qws, uint64_t* __restrict__ verdict_bits,
                                                                                uint8_t* __restrict__ status, sha_prefixes pre) {
    constexpr int NP = BLOCK / 2;
    const bool odd = (threadIdx.x & 1) != 0;
    const uint32_t pairidx = threadIdx.x >> 1;
    PairQTab<NP> qtab = PairQTab<NP>::of(qws + (size_t)blockIdx.x * (QWS_PAIR_UINT4_PER_SIG * NP), pairidx);
    uint32_t* verdict32 = reinterpret_cast<uint32_t*>(verdict_bits);
    const uint32_t ntiles = (n + NP - 1) / NP;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * NP + pairidx;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        uint32_t h[8];
        sha256_message(arena32, arena_words, off, pre, ic, active, h);
        emit_digest(pre, i, active && !odd, h);
        u256 vqx, vqy, ve, vr, vs;
#pragma unroll
        for (int k = 0; k < 8; k++) ve.w[k] = h[7 - k];
        load_be_field(vqx, qx, ic);
        load_be_field(vqy, qy, ic);
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = p256_verify_pair29(vqx, vqy, ve, vr, vs, gtab, qtab, odd);
        pair_emit_verdict(i, n, active, odd, st, verdict32, status);
    }
}

// identity.Verify fused: e = SHA-256(msg) stays in registers
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2) sha256_p256_verify_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words,
                                                                           const uint32_t* __restrict__ off, const uint8_t* __restrict__ qx,
                                                                           const uint8_t* __restrict__ qy, const uint8_t* __restrict__ r,
                                                                           const uint8_t* __restrict__ s, const int32_t* __restrict__ gtab,
                                                                           uint4* __restrict__ qws, uint64_t* __restrict__ verdict_bits,
                                                                           uint8_t* __restrict__ status, sha_prefixes pre) {
    GlobalQTab29<BLOCK> qtab = GlobalQTab29<BLOCK>::of(qws + (size_t)blockIdx.x * (QWS_UINT4_PER_LANE * BLOCK), threadIdx.x);
    GTab16 gt{gtab};
    const uint32_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * BLOCK + threadIdx.x;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        uint32_t h[8];
        sha256_message(arena32, arena_words, off, pre, ic, active, h);
        emit_digest(pre, i, active, h);
        u256 vqx, vqy, ve, vr, vs;
#pragma unroll
        for (int k = 0; k < 8; k++) ve.w[k] = h[7 - k];   // digest big-endian -> integer limbs
        load_be_field(vqx, qx, ic);
        load_be_field(vqy, qy, ic);
        load_be_field(vr, r, ic);
        load_be_field(vs, s, ic);
        uint32_t st = p256_verify_core29(vqx, vqy, ve, vr, vs, gt, qtab);
        emit_verdict(i, active, st, verdict_bits, status);
    }
}

// Stitches the pieces of gathered messages (fabgpu_identity_batch.gather_spans) into consecutive bytes: one WAVEFRONT per message,
// lane l moving bytes l, l + 64, ... of each piece (byte granularity because the pieces sit at arbitrary offsets of the block
// buffer; coalesced 64-byte rows).  A lane-per-message byte loop was measured first: its ~1500 dependent load/store round
// trips per lane cost 2 ms per 10 000-transaction block.
__global__ void __launch_bounds__(256) gather_spans_kernel(uint32_t n, const uint8_t* __restrict__ arena, uint32_t arena_bytes,
                                                            const uint32_t* __restrict__ spans, const uint32_t* __restrict__ out_off,
                                                            uint8_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= n) return;
    uint32_t o = out_off[i];
    const uint32_t oe = out_off[i + 1];
    for (int p = 0; p < 3; p++) {
        uint32_t s = spans[6 * (size_t)i + 2 * p], e = spans[6 * (size_t)i + 2 * p + 1];
        e = e < arena_bytes ? e : arena_bytes;
        if (e <= s) continue;
        uint32_t len = e - s;
        len = len < oe - o ? len : oe - o;
        for (uint32_t b = lane; b < len; b += 64) out[o + b] = arena[s + b];
        o += len;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers (host)
// ------------------------------------------------------------------------------------------------
// Runs the mid-state kernel for a prefixed batch (no-op otherwise) and returns the descriptor the fused kernels take.
static sha_prefixes launch_midstates(const void* arena, size_t arena_bytes, const ShaPrefixArgs& pa, hipStream_t st) {
    sha_prefixes pre{nullptr, nullptr, nullptr, 0, pa.spans ? 1u : 0u, (uint32_t*)pa.digests};
    if (pa.m == 0 || pa.pre_idx == nullptr) return pre;
    if (!pa.mid_ready) {
        dim3 grid((pa.m + 255) / 256), block(256);
        hipLaunchKernelGGL(sha256_midstate_kernel, grid, block, 0, st, pa.m, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                           (const uint32_t*)pa.pre_off, pre.spans, (uint32_t*)pa.mid_scratch);
    }
    pre.pre_idx = (const uint32_t*)pa.pre_idx;
    pre.pre_off = (const uint32_t*)pa.pre_off;
    pre.mid = (const uint32_t*)pa.mid_scratch;
    pre.m = pa.m;
    return pre;
}
hipError_t launch_sha256_midstates(const void* arena, size_t arena_bytes, const ShaPrefixArgs& pa, hipStream_t st) {
    if (pa.m == 0) return hipSuccess;
    dim3 grid((pa.m + 255) / 256), block(256);
    hipLaunchKernelGGL(sha256_midstate_kernel, grid, block, pa.lds_reserve, st, pa.m, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                       (const uint32_t*)pa.pre_off, pa.spans ? 1u : 0u, (uint32_t*)pa.mid_scratch);
    return hipGetLastError();
}
hipError_t launch_sha256_batch(uint32_t n, const void* arena, size_t arena_bytes, const void* off, void* digests, hipStream_t st, uint32_t lds_spread) {
    if (n == 0) return hipSuccess;
    if (n <= SHA_COOP_MAX) {                    // a launch that cannot fill the chip: eight lanes per message (sha256_coop.h)
        ShaPrefixArgs pa;
        pa.digests = digests;
        return launch_sha256_messages_coop(n, arena, arena_bytes, off, pa, st, lds_spread);
    }
    return launch_sha256_mixed(n, arena, arena_bytes, off, false, digests, st);
}
hipError_t launch_sha256_spans(uint32_t n, const void* arena, size_t arena_bytes, const void* spans, void* digests, hipStream_t st, uint32_t lds_reserve,
                               uint32_t lds_spread) {
    if (n == 0) return hipSuccess;
    if (n <= SHA_COOP_MAX && lds_reserve == 0) {
        ShaPrefixArgs pa;
        pa.spans = true;
        pa.digests = digests;
        return launch_sha256_messages_coop(n, arena, arena_bytes, spans, pa, st, lds_spread);
    }
    return launch_sha256_mixed(n, arena, arena_bytes, spans, true, digests, st, lds_reserve);
}
hipError_t launch_gather_sha256(uint32_t n, const void* arena, size_t arena_bytes, const void* spans, const void* out_off, void* scratch,
                                size_t scratch_bytes, void* digests, hipStream_t st, uint32_t lds_reserve, uint32_t lds_spread) {
    if (n == 0) return hipSuccess;
    dim3 grid((n + 3) / 4), block(256);   // four wavefronts = four messages per workgroup
    hipLaunchKernelGGL(gather_spans_kernel, grid, block, lds_reserve, st, n, (const uint8_t*)arena, (uint32_t)arena_bytes, (const uint32_t*)spans,
                       (const uint32_t*)out_off, (uint8_t*)scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (lds_reserve) return launch_sha256_mixed(n, scratch, scratch_bytes, out_off, false, digests, st, lds_reserve);   // keeping to its own CUs
    return launch_sha256_batch(n, scratch, scratch_bytes, out_off, digests, st, lds_spread);
}

VerifyGeom verify_geom(uint32_t n, bool allow_pair) {
    VerifyGeom g;
    g.block = VERIFY_BLOCK;
    g.pair = allow_pair && n <= (uint32_t)VERIFY_PAIR_MAX;
    uint32_t per_wg = g.pair ? VERIFY_BLOCK / 2 : VERIFY_BLOCK;
    uint32_t tiles = (n + per_wg - 1) / per_wg;
    g.wgs = tiles < (uint32_t)VERIFY_MAX_WGS ? tiles : (uint32_t)VERIFY_MAX_WGS;
    return g;
}
// Where the two-lanes-per-signature verify-only kernel keeps its per-signature table when the context does not say
// (FABGPU_FLAG_PAIR_TABLE_LDS / _GLOBAL in fabgpu_cfg.flags force one; bench.py --pair-table for A/B runs): -1 = by batch size (launch_p256_verify).
int pair_table_default() { return -1; }
size_t pair_table_lds_bytes() { return (size_t)(VERIFY_BLOCK / 2) * PAIR_LDS_CELLS_PER_SIG * 16; }
size_t verify_workspace_bytes(uint32_t n, bool allow_pair) {
    VerifyGeom g = verify_geom(n, allow_pair);
    if (g.pair) return (size_t)g.wgs * (g.block / 2) * QWS_PAIR_UINT4_PER_SIG * 16;
    return (size_t)g.wgs * g.block * QWS_UINT4_PER_LANE * 16;
}
hipError_t launch_p256_verify(uint32_t n, const void* qx, const void* qy, const void* e, const void* r, const void* s,
                              const void* gtab, void* qws, void* verdict_bits, void* status, bool allow_pair, hipStream_t st, uint32_t lds_reserve, int table_lds) {
    if (n == 0) return hipSuccess;
    VerifyGeom g = verify_geom(n, allow_pair);
    dim3 grid(g.wgs), block(g.block);
    // The per-signature table of the pair kernel: in LDS when the launch is large (measured on MI355X, tools/gpu_pair_table_ab.py and
    // tools/gpu_pmc_traffic.sh: at 30 000 tuples the two forms take the same time - 0.631 / 0.629 ms back to back - and the LDS form
    // moves 94 MB through the memory system per launch instead of 279 MB); in the global workspace for smaller ones, where the LDS
    // form's 13 extra additions show as latency (10 000 tuples: 0.624 against 0.614 ms; 1 000: 0.618 against 0.603 ms).
    if (table_lds < 0) table_lds = n > (uint32_t)PAIR_TABLE_LDS_FROM ? 1 : 0;
    if (g.pair && table_lds) {
        hipLaunchKernelGGL(p256_verify_pair_lds_kernel<VERIFY_BLOCK>, grid, block, (size_t)(VERIFY_BLOCK / 2) * PAIR_LDS_CELLS_PER_SIG * 16, st, n, (const uint8_t*)qx,
                           (const uint8_t*)qy, (const uint8_t*)e, (const uint8_t*)r, (const uint8_t*)s, (const int32_t*)gtab, (uint64_t*)verdict_bits, (uint8_t*)status);
        return hipGetLastError();
    }
    if (g.pair) {
        hipLaunchKernelGGL(p256_verify_pair_kernel<VERIFY_BLOCK>, grid, block, lds_reserve, st, n, (const uint8_t*)qx, (const uint8_t*)qy, (const uint8_t*)e,
                           (const uint8_t*)r, (const uint8_t*)s, (const int32_t*)gtab, (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(p256_verify_kernel<VERIFY_BLOCK>, grid, block, lds_reserve, st, n, (const uint8_t*)qx, (const uint8_t*)qy, (const uint8_t*)e,
                       (const uint8_t*)r, (const uint8_t*)s, (const int32_t*)gtab, (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status);
    return hipGetLastError();
}
hipError_t launch_sha256_p256_verify(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* qx,
                                     const void* qy, const void* r, const void* s, const void* gtab, void* qws,
                                     void* verdict_bits, void* status, bool allow_pair, const ShaPrefixArgs& pa, hipStream_t st) {
    if (n == 0) return hipSuccess;
    sha_prefixes pre = launch_midstates(arena, arena_bytes, pa, st);
    VerifyGeom g = verify_geom(n, allow_pair);
    dim3 grid(g.wgs), block(g.block);
    if (g.pair) {
        hipLaunchKernelGGL(sha256_p256_verify_pair_kernel<VERIFY_BLOCK>, grid, block, pa.lds_reserve, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                           (const uint32_t*)off, (const uint8_t*)qx, (const uint8_t*)qy, (const uint8_t*)r, (const uint8_t*)s,
                           (const int32_t*)gtab, (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status, pre);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(sha256_p256_verify_kernel<VERIFY_BLOCK>, grid, block, pa.lds_reserve, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                       (const uint32_t*)off, (const uint8_t*)qx, (const uint8_t*)qy, (const uint8_t*)r, (const uint8_t*)s,
                       (const int32_t*)gtab, (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status, pre);
    return hipGetLastError();
}

hipError_t launch_p256_verify_keyed(uint32_t n, const void* key_id, uint32_t nkeys, const void* ktabs, const void* e, const void* r, const void* s,
                                    const void* gtab, void* verdict_bits, void* status, bool allow_pair, hipStream_t st, uint32_t lds_reserve) {
    if (n == 0) return hipSuccess;
    VerifyGeom g = verify_geom(n, allow_pair);
    dim3 grid(g.wgs), block(g.block);
    if (g.pair)
        hipLaunchKernelGGL(p256_verify_keyed_pair_kernel<VERIFY_BLOCK>, grid, block, lds_reserve, st, n, (const uint32_t*)key_id, nkeys, (const int32_t* const*)ktabs,
                           (const uint8_t*)e, (const uint8_t*)r, (const uint8_t*)s, (const int32_t*)gtab, (uint64_t*)verdict_bits, (uint8_t*)status);
    else
        hipLaunchKernelGGL(p256_verify_keyed_kernel<VERIFY_BLOCK>, grid, block, lds_reserve, st, n, (const uint32_t*)key_id, nkeys, (const int32_t* const*)ktabs,
                           (const uint8_t*)e, (const uint8_t*)r, (const uint8_t*)s, (const int32_t*)gtab, (uint64_t*)verdict_bits, (uint8_t*)status);
    return hipGetLastError();
}

hipError_t launch_sha256_p256_verify_keyed(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* key_id, uint32_t nkeys,
                                           const void* ktabs, const void* r, const void* s, const void* gtab, void* verdict_bits, void* status,
                                           bool allow_pair, const ShaPrefixArgs& pa, hipStream_t st) {
    if (n == 0) return hipSuccess;
    sha_prefixes pre = launch_midstates(arena, arena_bytes, pa, st);
    VerifyGeom g = verify_geom(n, allow_pair);
    dim3 grid(g.wgs), block(g.block);
    if (g.pair)
        hipLaunchKernelGGL(sha256_p256_verify_keyed_pair_kernel<VERIFY_BLOCK>, grid, block, pa.lds_reserve, st, n, (const uint32_t*)arena,
                           (uint32_t)((arena_bytes + 3) / 4), (const uint32_t*)off, (const uint32_t*)key_id, nkeys, (const int32_t* const*)ktabs,
                           (const uint8_t*)r, (const uint8_t*)s, (const int32_t*)gtab, (uint64_t*)verdict_bits, (uint8_t*)status, pre);
    else
        hipLaunchKernelGGL(sha256_p256_verify_keyed_kernel<VERIFY_BLOCK>, grid, block, pa.lds_reserve, st, n, (const uint32_t*)arena,
                           (uint32_t)((arena_bytes + 3) / 4), (const uint32_t*)off, (const uint32_t*)key_id, nkeys, (const int32_t* const*)ktabs,
                           (const uint8_t*)r, (const uint8_t*)s, (const int32_t*)gtab, (uint64_t*)verdict_bits, (uint8_t*)status, pre);
    return hipGetLastError();
}

// The runtime resolves a kernel FUNCTION (symbol lookup, kernel object, argument layout) at its first launch, on the launching thread -
// after the code object of its translation unit is loaded, which the provider's construction already rehearses with one launch per unit.
// The keyed kernels are first launched by the second block of a fresh provider (its identities earn their tables during the first):
// asking for every function's attributes now moves that work to construction too (GPUCSP::Preallocate).  Returns how many resolved.
int warm_kernel_functions_kernels() {
    int ok = 0;
    hipFuncAttributes a;
    const void* fns[] = {(const void*)p256_verify_kernel<VERIFY_BLOCK>, (const void*)p256_verify_pair_kernel<VERIFY_BLOCK>,
                         (const void*)p256_verify_pair_lds_kernel<VERIFY_BLOCK>, (const void*)p256_verify_keyed_kernel<VERIFY_BLOCK>,
                         (const void*)p256_verify_keyed_pair_kernel<VERIFY_BLOCK>, (const void*)sha256_p256_verify_keyed_kernel<VERIFY_BLOCK>,
                         (const void*)sha256_p256_verify_keyed_pair_kernel<VERIFY_BLOCK>, (const void*)sha256_p256_verify_pair_kernel<VERIFY_BLOCK>,
                         (const void*)sha256_p256_verify_kernel<VERIFY_BLOCK>, (const void*)sha256_midstate_kernel, (const void*)gather_spans_kernel};
    for (const void* f : fns) ok += hipFuncGetAttributes(&a, f) == hipSuccess ? 1 : 0;
    return ok;
}

}  // namespace fab
