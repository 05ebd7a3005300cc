// Kernels of the device-side block walk (block_walk_dev.h) for gfx950.  This is HBM / latency-bound byte work - protobuf framing,
// DER, byte comparisons - so the rules are the memory ones: the marshalled block is read where fabgpu_arena_stage put it (once by the
// walk: field headers only, the payloads are skipped), one lane per envelope for the walk (25 nested messages per transaction, a
// dependent chain per lane, 10 000 chains in flight), one WAVEFRONT per tuple where whole byte strings are touched (identity
// hash + comparison: 64-byte coalesced rows), SoA outputs written where the fused verify launch reads them - nothing of the
// submission crosses PCIe.  No MFMA, no LDS beyond the scan.
//
// The walk itself is walk::walk_envelope of block_walk_core.h - the SAME template the host walker instantiates - run twice: once with
// a counting emitter, once (after an exclusive scan of the counts) with an emitter that writes at the assigned offsets, which keeps
// the host walker's order: envelope order, then emission order inside the envelope.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fabgpu.h"
#include "block_prepass.h"
#include "block_walk_dev.h"

namespace fab {

using bccsp::BlockHashCheck;
using bccsp::BlockTuple;
using bccsp::Span;

namespace {

using bccsp::walk::CountEmitter;
using bccsp::walk::WriteEmitter;

__global__ void __launch_bounds__(64) walk_count_kernel(WalkArrays a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_env) return;
    uint32_t off = a.env_spans[2 * e], len = a.env_spans[2 * e + 1];
    if (off > a.block_len || len > a.block_len - off) off = len = 0;       // (the list comes from the host's lister)
    CountEmitter em;
    uint8_t type = 255, understood = 0;
    bccsp::walk::walk_envelope(a.block, a.block + off, len, e, em, type, understood);
    a.counts[e] = make_uint4(em.nt, em.np, em.nc, em.gb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)em.gb);
    a.tx_type[e] = type;
    a.tx_understood[e] = understood;
}

// Exclusive prefix sums of the per-envelope counts: ONE workgroup (10 000 envelopes are ten per thread; the block-wide part is a
// 1024-entry Hillis-Steele scan in LDS).
__global__ void __launch_bounds__(1024) walk_scan_kernel(uint32_t n, const uint4* __restrict__ counts, uint4* __restrict__ bases, uint32_t* __restrict__ cbase,
                                                         WalkTotals* __restrict__ totals) {
    __shared__ uint32_t st[1024], sp[1024], sc[1024], sk[1024];
    __shared__ uint64_t sg[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t t = 0, p = 0, c = 0, k = 0;
    uint64_t g = 0;
    for (uint32_t i = lo; i < hi; i++) {
        const uint4 v = counts[i];
        t += v.x; p += v.y; c += v.z; g += v.w;
        k += v.x ? 1u : 0u;
    }
    st[tid] = t; sp[tid] = p; sc[tid] = c; sg[tid] = g; sk[tid] = k;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t vt = 0, vp = 0, vc = 0, vk = 0;
        uint64_t vg = 0;
        if (tid >= o) { vt = st[tid - o]; vp = sp[tid - o]; vc = sc[tid - o]; vg = sg[tid - o]; vk = sk[tid - o]; }
        __syncthreads();
        st[tid] += vt; sp[tid] += vp; sc[tid] += vc; sg[tid] += vg; sk[tid] += vk;
        __syncthreads();
    }
    uint32_t bt = st[tid] - t, bp = sp[tid] - p, bc = sc[tid] - c, bk = sk[tid] - k;
    uint64_t bg = sg[tid] - g;
    for (uint32_t i = lo; i < hi; i++) {
        const uint4 v = counts[i];
        bases[i] = make_uint4(bt, bp, bc, (uint32_t)bg);
        cbase[i] = bk;
        bt += v.x; bp += v.y; bc += v.z; bg += v.w;
        bk += v.x ? 1u : 0u;
    }
    if (tid == 1023) {
        totals->tuples = st[1023];
        totals->prefixes = sp[1023];
        totals->checks = sc[1023];
        totals->creators = sk[1023];
        totals->gather_bytes = sg[1023];
    }
}

__global__ void __launch_bounds__(64) walk_emit_kernel(WalkArrays a, uint32_t n_checks, uint32_t gather_total) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) a.gather_off[n_checks] = gather_total;
    if (e >= a.n_env) return;
    const uint4 cnt = a.counts[e];
    if (cnt.x == 0 && cnt.y == 0 && cnt.z == 0) return;
    const uint4 base = a.bases[e];
    uint32_t off = a.env_spans[2 * e], len = a.env_spans[2 * e + 1];
    if (off > a.block_len || len > a.block_len - off) off = len = 0;
    WriteEmitter em{a.tuples, a.pre_off2, a.checks, a.gather_spans, a.gather_off, base.x, base.y, base.z, base.w, cnt.x, cnt.y, cnt.z, a.creator_spans, a.cbase[e]};
    uint8_t type = 255, understood = 0;
    bccsp::walk::walk_envelope(a.block, a.block + off, len, e, em, type, understood);
}

// identity bytes -> index of the provider's cache entry (0xFFFFFFFF: not in the table).  One wavefront per tuple: one coalesced row
// for the hash (length + last 64 bytes), then the ~800 bytes of the SerializedIdentity against the candidate's in 256-byte rows
// (a dword per lane; the block side sits at an arbitrary byte offset: unaligned dword loads, which global memory serves).
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__global__ void __launch_bounds__(256) walk_identity_kernel(WalkArrays a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= a.n_tuples) return;
    const Span id = a.tuples[i].identity;
    uint32_t found = 0xFFFFFFFFu;
    if (a.id_mask != 0 && id.off <= a.arena_len && id.len <= a.arena_len - id.off) {
        const uint8_t* p = a.block + id.off;
        const uint32_t m = id.len < 64 ? id.len : 64;
        uint64_t h = 0xCBF29CE484222325ull;
        if (lane < m) h = bccsp::walk::id_stream_fold(h, p[id.len - m + lane]);
        uint64_t term = h * bccsp::walk::id_stream_const(lane);
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t lo32 = __shfl_xor((uint32_t)term, o, 64), hi32 = __shfl_xor((uint32_t)(term >> 32), o, 64);
            term += ((uint64_t)hi32 << 32) | lo32;
        }
        const uint64_t hash = bccsp::walk::id_hash_finish(term, id.len);
        const uint32_t ndw = id.len >> 2, rest = id.len & 3u;
        uint32_t slot = (uint32_t)hash & a.id_mask;
        for (uint32_t probes = 0; probes <= a.id_mask; probes++) {
            const uint32_t e = a.id_slots[slot];
            if (e == 0) break;
            const DevIdEntry* ent = a.id_entries + (e - 1);
            if (ent->hash == hash && ent->len == id.len) {
                const uint8_t* q = a.id_bytes + ent->off;                  // (4-byte aligned: walk_idtab_set's callers lay the bytes out so)
                bool differs = false;
                for (uint32_t w = lane; w < ndw; w += 64)
                    differs |= *reinterpret_cast<const u32_unaligned*>(p + 4 * (size_t)w) != *reinterpret_cast<const u32_unaligned*>(q + 4 * (size_t)w);
                if (lane < rest) differs |= p[4 * (size_t)ndw + lane] != q[4 * (size_t)ndw + lane];
                if (__ballot(differs) == 0) {
                    found = e - 1;
                    break;
                }
            }
            slot = (slot + 1) & a.id_mask;
        }
    }
    if (lane == 0) a.id_idx[i] = found;
}

// The gates of one tuple and its row of the submission arrays.  A tuple the device does not decide still gets a well-formed row
// (r = s = 1 under the generator as key, like the host's fillers): there is no compaction - tuple i is submission i - and its
// verdict is ignored in favour of gate_st.
__global__ void __launch_bounds__(256) walk_gate_kernel(WalkArrays a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_tuples) return;
    // the generator of P-256: key of the filler rows
    const uint8_t GX[32] = {0x6b, 0x17, 0xd1, 0xf2, 0xe1, 0x2c, 0x42, 0x47, 0xf8, 0xbc, 0xe6, 0xe5, 0x63, 0xa4, 0x40, 0xf2,
                            0x77, 0x03, 0x7d, 0x81, 0x2d, 0xeb, 0x33, 0xa0, 0xf4, 0xa1, 0x39, 0x45, 0xd8, 0x98, 0xc2, 0x96};
    const uint8_t GY[32] = {0x4f, 0xe3, 0x42, 0xe2, 0xfe, 0x1a, 0x7f, 0x9b, 0x8e, 0xe7, 0xeb, 0x4a, 0x7c, 0x0f, 0x9e, 0x16,
                            0x2b, 0xce, 0x33, 0x57, 0x6b, 0x31, 0x5e, 0xce, 0xcb, 0xb6, 0x40, 0x68, 0x37, 0xbf, 0x51, 0xf5};
    const BlockTuple t = a.tuples[i];
    // the row of this tuple (WalkArrays::row_of): creators first when the submission is split
    uint32_t row = i;
    if (a.split) {
        // creator tuples at indices <= i: those of the envelopes before this one, plus this envelope's (its first tuple)
        const uint32_t before = i < a.n_dev_tuples ? a.cbase[t.tx] : a.n_creators;
        const bool creator = i < a.n_dev_tuples && i == a.bases[t.tx].x;
        row = creator ? before : a.n_creators + (i - before - (i < a.n_dev_tuples ? 1u : 0u));
    }
    a.row_of[i] = row;
    a.off2[2 * (size_t)row] = t.suffix.len ? t.suffix.off : 0;
    a.off2[2 * (size_t)row + 1] = t.suffix.len ? t.suffix.off + t.suffix.len : 0;
    a.pre_idx[row] = t.prefix_index >= 0 ? (uint32_t)t.prefix_index : 0xFFFFFFFFu;
    uint8_t r32[32], s32[32];
    for (int k = 0; k < 32; k++) r32[k] = s32[k] = 0;
    r32[31] = s32[31] = 1;
    uint8_t gst;
    bool submit = false;
    const uint32_t idx = a.id_idx[i];
    const DevIdEntry* ent = idx != 0xFFFFFFFFu ? a.id_entries + idx : nullptr;
    bool unknown = false, declined = false;
    if (!ent) {
        unknown = true;
        gst = bccsp::TUPLE_ST_NEEDS_SW;                             // (the host walk takes the whole block when this count is not zero)
    } else if (!ent->p256) {
        gst = bccsp::TUPLE_ST_NEEDS_SW;
    } else if (t.sig.len == 0) {
        gst = bccsp::TUPLE_ST_EMPTY_SIG;
    } else {
        uint8_t g = bccsp::walk::GATE_DECLINED;
        if (t.sig.off <= a.arena_len && t.sig.len <= a.arena_len - t.sig.off) g = bccsp::walk::gate_sig_fast(a.block + t.sig.off, t.sig.len, r32, s32);
        if (g == bccsp::walk::GATE_SUBMIT) {
            gst = FABGPU_ST_VALID;
            submit = true;
        } else if (g == bccsp::walk::GATE_HIGH_S) {
            gst = FABGPU_ST_HIGH_S;
        } else {
            declined = true;
            gst = bccsp::TUPLE_ST_BAD_DER;                          // (never reported: a declined signature sends the block to the host walk)
        }
        if (!submit) {
            for (int k = 0; k < 32; k++) r32[k] = s32[k] = 0;
            r32[31] = s32[31] = 1;
        }
    }
    const bool keyed = submit && ent->key_id >= 0;
    {   // the summary: one atomic per wavefront and counter, not one per tuple
        const uint64_t live = __ballot(true);
        const bool first = (uint32_t)(__ffsll((unsigned long long)live) - 1) == (threadIdx.x & 63u);
        const uint32_t nu = __popcll(__ballot(unknown)), nd = __popcll(__ballot(declined)), ns = __popcll(__ballot(submit)), nk = __popcll(__ballot(submit && !keyed));
        if (first) {
            if (nu) atomicAdd(&a.summary->n_unknown_identity, nu);
            if (nd) atomicAdd(&a.summary->n_declined, nd);
            if (ns) atomicAdd(&a.summary->n_submitted, ns);
            if (nk) atomicAdd(&a.summary->n_unkeyed, nk);
        }
    }
    a.key_id[row] = keyed ? (uint32_t)ent->key_id : 0u;
    uint8_t* qx = a.qx + 32 * (size_t)row;
    uint8_t* qy = a.qy + 32 * (size_t)row;
    uint8_t* r = a.r + 32 * (size_t)row;
    uint8_t* s = a.s + 32 * (size_t)row;
    for (int k = 0; k < 32; k++) {
        qx[k] = submit ? ent->qx[k] : GX[k];
        qy[k] = submit ? ent->qy[k] : GY[k];
        r[k] = r32[k];
        s[k] = s32[k];
    }
    a.gate_st[i] = gst;
}

// per-transaction evidence bits
enum : uint32_t { M_BAD_CREATOR = 1, M_BAD_END = 2, M_SW = 4, M_BAD_TXID = 8, M_BAD_PHASH = 16 };

__global__ void __launch_bounds__(256) walk_status_kernel(WalkArrays a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_tuples) return;
    const uint8_t gst = a.gate_st[i];
    const BlockTuple t = a.tuples[i];
    uint8_t st = gst;
    uint8_t hashed = 0;
    const uint32_t row = a.row_of[i];
    if (a.row_digests) {
        const uint4* src = reinterpret_cast<const uint4*>(a.row_digests + 32 * (size_t)row);
        uint4* dst = reinterpret_cast<uint4*>(a.tuple_digests + 32 * (size_t)i);
        dst[0] = src[0];
        dst[1] = src[1];
    }
    if (gst == FABGPU_ST_VALID) {                                    // the device decided: exactly PreVerifyParsed's mapping
        const bool in_c = a.split && row < a.n_creators;
        const uint32_t j = in_c ? row : row - (a.split ? a.n_creators : 0u);
        const bool bit = ((in_c ? a.verdict_bits_c : a.verdict_bits)[j >> 6] >> (j & 63)) & 1;
        const uint8_t ds = a.dev_status[row];
        st = (bit && ds == FABGPU_ST_VALID) ? FABGPU_ST_VALID : (ds == FABGPU_ST_VALID ? FABGPU_ST_BAD_MATH : ds);
        hashed = 1;
    }
    a.tuple_status[i] = st;
    a.tuple_hashed[i] = hashed;
    if (st != FABGPU_ST_VALID && t.tx < a.n_env)                     // block-level tuples (orderer signatures) do not flag a transaction
        atomicOr(&a.tx_mask[t.tx], st == bccsp::TUPLE_ST_NEEDS_SW ? M_SW : (t.kind == bccsp::TUPLE_CREATOR ? M_BAD_CREATOR : M_BAD_END));
}

// does the digest equal what the block says (block_prepass.cpp HashCheckMatches: lowercase hex for the TxID, raw bytes for the proposal hash)
__global__ void __launch_bounds__(256) walk_checks_kernel(WalkArrays a, uint32_t n_checks) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_checks) return;
    const BlockHashCheck hc = a.checks[j];
    const uint8_t* d = a.gather_digests + 32 * (size_t)j;
    const uint8_t* e = a.block + hc.expect.off;
    bool ok;
    if (hc.kind == bccsp::HASH_PROPOSAL) {
        ok = hc.expect.len == 32;
        for (int k = 0; ok && k < 32; k++) ok = e[k] == d[k];
    } else {
        ok = hc.expect.len == 64;
        for (int k = 0; ok && k < 32; k++) {
            const uint32_t hi = d[k] >> 4, lo = d[k] & 15;
            ok = e[2 * k] == (uint8_t)(hi < 10 ? '0' + hi : 'a' + hi - 10) && e[2 * k + 1] == (uint8_t)(lo < 10 ? '0' + lo : 'a' + lo - 10);
        }
    }
    if (!ok && hc.tx < a.n_env) atomicOr(&a.tx_mask[hc.tx], hc.kind == bccsp::HASH_TXID ? M_BAD_TXID : M_BAD_PHASH);
}

// in the order ValidateTransaction and then VSCC would reject (PreVerifyParsed): not understood > bad creator signature > bad TxID >
// bad proposal hash > bad endorsement > "ask bccsp/sw" > all valid
__global__ void __launch_bounds__(256) walk_txflags_kernel(WalkArrays a) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_env) return;
    const uint32_t m = a.tx_mask[t];
    uint8_t f;
    if (!a.tx_understood[t]) f = bccsp::TX_NOT_UNDERSTOOD;
    else if (m & M_BAD_CREATOR) f = bccsp::TX_BAD_CREATOR_SIGNATURE;
    else if (m & M_BAD_TXID) f = bccsp::TX_BAD_TXID;
    else if (m & M_BAD_PHASH) f = bccsp::TX_BAD_PROPOSAL_HASH;
    else if (m & M_BAD_END) f = bccsp::TX_BAD_ENDORSEMENT;
    else if (m & M_SW) f = bccsp::TX_NEEDS_SW;
    else f = bccsp::TX_ALL_SIGNATURES_VALID;
    a.tx_flags[t] = f;
}

}  // namespace

hipError_t launch_walk_count(const WalkArrays& a, hipStream_t st) {
    if (a.n_env) {
        hipLaunchKernelGGL(walk_count_kernel, dim3((a.n_env + 63) / 64), dim3(64), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(walk_scan_kernel, dim3(1), dim3(1024), 0, st, a.n_env, (const uint4*)a.counts, a.bases, a.cbase, a.totals);
    return hipGetLastError();
}
hipError_t launch_walk_emit(const WalkArrays& a, const WalkTotals& t, hipStream_t st) {
    const uint32_t n = a.n_env ? a.n_env : 1;                       // (lane 0 closes gather_off even without envelopes)
    hipLaunchKernelGGL(walk_emit_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a, t.checks, (uint32_t)t.gather_bytes);
    return hipGetLastError();
}
hipError_t launch_walk_gate(const WalkArrays& a, hipStream_t st) {
    if (a.n_tuples == 0) return hipSuccess;
    hipLaunchKernelGGL(walk_identity_kernel, dim3((a.n_tuples + 3) / 4), dim3(256), 0, st, a);   // four wavefronts = four tuples per workgroup
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(walk_gate_kernel, dim3((a.n_tuples + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_walk_flags(const WalkArrays& a, uint32_t n_checks, hipStream_t st) {
    hipError_t e = hipSuccess;
    if (a.n_tuples) {
        hipLaunchKernelGGL(walk_status_kernel, dim3((a.n_tuples + 255) / 256), dim3(256), 0, st, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess && n_checks) {
        hipLaunchKernelGGL(walk_checks_kernel, dim3((n_checks + 255) / 256), dim3(256), 0, st, a, n_checks);
        e = hipGetLastError();
    }
    if (e == hipSuccess && a.n_env) {
        hipLaunchKernelGGL(walk_txflags_kernel, dim3((a.n_env + 255) / 256), dim3(256), 0, st, a);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace fab
