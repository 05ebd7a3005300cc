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
#include "p256_verify29.h"   // on_curve29: the curve-membership gate of a certificate key the device decodes itself

namespace fab {

using bccsp::BlockHashCheck;
using bccsp::BlockTuple;
using bccsp::Span;

namespace {

// (a wavefront's LDS accesses execute in order: between "these lanes wrote" and "that lane reads" only the compiler must be held)
__device__ __forceinline__ void wave_lds_sync_early() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

using bccsp::walk::CountEmitter;
using bccsp::walk::StashEmitter;
using bccsp::walk::WriteEmitter;

__global__ void __launch_bounds__(64) walk_count_kernel(WalkArrays a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_env) return;
    uint32_t off = a.env_spans[2 * e], len = a.env_spans[2 * e + 1];
    if (off > a.block_len || len > a.block_len - off) off = len = 0;       // (the list comes from the host's lister)
    uint8_t type = 255, understood = 0;
    uint32_t nt, np, nc;
    uint64_t gb;
    if (a.stash) {
        bccsp::walk::StashEmitter em{a.stash + e};
        bccsp::walk::walk_envelope(a.block, a.block + off, len, e, em, type, understood);
        nt = em.nt; np = em.np; nc = em.nc; gb = em.gb;
        a.stash[e].over = em.fits() ? 0u : 1u;
    } else {
        CountEmitter em;
        bccsp::walk::walk_envelope(a.block, a.block + off, len, e, em, type, understood);
        nt = em.nt; np = em.np; nc = em.nc; gb = em.gb;
    }
    a.counts[e] = make_uint4(nt, np, nc, gb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)gb);
    a.tx_type[e] = type;
    a.tx_understood[e] = understood;
}

// The same for a block of at most WALK_STAGED_MAX envelopes: ONE WAVEFRONT per envelope.  A lane that walks its envelope in global
// memory follows ~75 dependent byte loads, each an L2 round trip - 35 us per envelope whatever the block's size, in front of everything
// else a small block's pass does.  Here the 64 lanes copy the envelope into LDS first (one coalesced round trip) and lane 0 walks the
// copy (spans stay block-relative: the walker only ever subtracts its `block` argument from pointers it derived from `env`).  An
// envelope that does not fit the window is walked in place.  Beyond a few thousand envelopes the lane-per-envelope kernel is the
// better one: ten thousand one-lane wavefronts are more issue slots than the chip has to spare (measured in round 4: 104 -> 206 us).
constexpr uint32_t WALK_STAGE_BYTES = 8192, WALK_STAGED_MAX = 2048;
__global__ void __launch_bounds__(256) walk_count_staged_kernel(WalkArrays a) {
    __shared__ uint4 window[4][WALK_STAGE_BYTES / 16];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t e = blockIdx.x * 4 + wave;
    if (e >= a.n_env) return;
    uint32_t off = a.env_spans[2 * e], len = a.env_spans[2 * e + 1];
    if (off > a.block_len || len > a.block_len - off) off = len = 0;
    const uint32_t a0 = off & ~15u, nvec = (off + len - a0 + 15u) >> 4;       // 16-byte vectors that cover the envelope (the block's allocation is padded)
    const bool fits = nvec <= WALK_STAGE_BYTES / 16;
    if (fits) {
        const uint4* src = reinterpret_cast<const uint4*>(a.block + a0);
        for (uint32_t v = lane; v < nvec; v += 64) window[wave][v] = src[v];
        wave_lds_sync_early();
    }
    if (lane != 0) return;
    uint8_t type = 255, understood = 0;
    bccsp::walk::StashEmitter em{a.stash + e};
    if (fits) {
        const uint8_t* env = reinterpret_cast<const uint8_t*>(window[wave]) + (off - a0);
        bccsp::walk::walk_envelope(env - off, env, len, e, em, type, understood);
    } else {
        bccsp::walk::walk_envelope(a.block, a.block + off, len, e, em, type, understood);
    }
    a.stash[e].over = em.fits() ? 0u : 1u;
    a.counts[e] = make_uint4(em.nt, em.np, em.nc, em.gb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)em.gb);
    a.tx_type[e] = type;
    a.tx_understood[e] = understood;
}

// Exclusive prefix sums of the per-envelope counts: ONE workgroup (10 000 envelopes are ten per thread).
__global__ void __launch_bounds__(1024) walk_scan_kernel(uint32_t n, const uint4* __restrict__ counts, uint4* __restrict__ bases, uint32_t* __restrict__ cbase,
                                                         WalkTotals* __restrict__ totals, WalkTotals* __restrict__ host_totals, uint32_t* __restrict__ host_flag,
                                                         uint32_t seq) {
    // the block-wide part: an inclusive scan inside each wavefront by lane shuffles (six steps), the sixteen wavefronts' totals through
    // LDS - two barriers instead of the twenty of a 1024-entry Hillis-Steele scan in LDS.  (Measured: no change where it was hoped for -
    // 31 us for 10 000 envelopes either way: the one workgroup shares its CU with the creators' payload hashes and waits for issue slots.)
    __shared__ uint32_t wt[16], wp[16], wc[16], wk[16];
    __shared__ uint64_t wg[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t t = 0, p = 0, c = 0, k = 0;
    uint64_t g = 0;
    for (uint32_t i = lo; i < hi; i += 4) {                             // (four independent loads at a time)
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = i + j < hi ? counts[i + j] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t += v[j].x; p += v[j].y; c += v[j].z; g += v[j].w;
            k += v[j].x ? 1u : 0u;
        }
    }
    uint32_t it = t, ip = p, ic = c, ik = k;                              // inclusive sums
    uint64_t ig = g;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t vt = __shfl_up(it, o, 64), vp = __shfl_up(ip, o, 64), vc = __shfl_up(ic, o, 64), vk = __shfl_up(ik, o, 64);
        const uint32_t glo = __shfl_up((uint32_t)ig, o, 64), ghi = __shfl_up((uint32_t)(ig >> 32), o, 64);
        if (lane >= (uint32_t)o) { it += vt; ip += vp; ic += vc; ik += vk; ig += ((uint64_t)ghi << 32) | glo; }
    }
    if (lane == 63) { wt[wave] = it; wp[wave] = ip; wc[wave] = ic; wk[wave] = ik; wg[wave] = ig; }
    __syncthreads();
    for (uint32_t w = 0; w < wave; w++) { it += wt[w]; ip += wp[w]; ic += wc[w]; ik += wk[w]; ig += wg[w]; }
    uint32_t bt = it - t, bp = ip - p, bc = ic - c, bk = ik - k;
    uint64_t bg = ig - g;
    for (uint32_t i = lo; i < hi; i++) {
        const uint4 v = counts[i];
        bases[i] = make_uint4(bt, bp, bc, (uint32_t)bg);
        cbase[i] = bk;
        bt += v.x; bp += v.y; bc += v.z; bg += v.w;
        bk += v.x ? 1u : 0u;
    }
    if (tid == 1023) {                                             // (the last thread's inclusive sums are the totals)
        totals->tuples = it;
        totals->prefixes = ip;
        totals->checks = ic;
        totals->creators = ik;
        totals->gather_bytes = ig;
        if (host_totals) {                                         // the host sizes everything else from these: it polls host_flag
            host_totals->tuples = it;
            host_totals->prefixes = ip;
            host_totals->checks = ic;
            host_totals->creators = ik;
            host_totals->gather_bytes = ig;
            __threadfence_system();
            __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void __launch_bounds__(64) walk_emit_kernel(WalkArrays a, uint32_t n_checks, uint32_t gather_total) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e == 0) a.gather_off[n_checks] = gather_total;
    if (e >= a.n_env) return;
    const uint4 cnt = a.counts[e];
    const uint4 base = a.bases[e];
    uint32_t off = a.env_spans[2 * e], len = a.env_spans[2 * e + 1];
    if (off > a.block_len || len > a.block_len - off) off = len = 0;
    WriteEmitter em{a.tuples, a.pre_off2, a.checks, a.gather_spans, a.gather_off, base.x, base.y, base.z, base.w, cnt.x, cnt.y, cnt.z, a.creator_spans, a.cbase[e]};
    if (a.stash && a.stash[e].over == 0) {
        // what the count kernel kept for this envelope, to the places the scan assigned (the emitter's own stores: one body of code
        // for "where a record goes"); the prefix indices become global here
        const bccsp::walk::EnvStash& st = a.stash[e];
        for (uint32_t k = 0; k < cnt.y; k++) em.add_prefix(st.p[k]);
        for (uint32_t k = 0; k < cnt.x; k++) {
            BlockTuple t = st.t[k];
            if (t.prefix_index >= 0) t.prefix_index += (int32_t)base.y;
            em.add_tuple(t);
        }
        for (uint32_t k = 0; k < cnt.z; k++) em.add_check(st.c[k]);
        return;
    }
    uint8_t type = 255, understood = 0;
    bccsp::walk::walk_envelope(a.block, a.block + off, len, e, em, type, understood);
    // the counts this envelope's slots were sized from (the count kernel's, or the host's - WalkRequest::host_counts) against this walk
    if (em.nt != cnt.x || em.np != cnt.y || em.nc != cnt.z || (cnt.w != 0xFFFFFFFFu && em.g != cnt.w) || type != a.tx_type[e] ||
        understood != a.tx_understood[e])
        atomicAdd(&a.summary->n_outline_differs, 1u);
}

// per-tuple notes of the gate kernel for the summary (WalkArrays::tflags)
enum : uint8_t { TF_SUBMIT = 1, TF_KEYED = 2, TF_UNKNOWN = 4, TF_GENERAL = 8, TF_OUTLINE = 16, TF_UNDECIDED = 32, TF_NYM = 64 };
// gate_st of an idemix creator whose pseudonym signature went to the nym kernel (internal: walk_status_kernel replaces it)
constexpr uint8_t GATE_ST_NYM = 250;

// ---- one wavefront per tuple: identity lookup, signature gate, submission row -----------------------------------------------------
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
// n/2 of P-256 (bccsp/utils/ecdsa.go:30-37 curveHalfOrders) and the generator (key of the filler rows), big-endian bytes
__constant__ uint8_t C_HALF_N[32] = {0x7f, 0xff, 0xff, 0xff, 0x80, 0x00, 0x00, 0x00, 0x7f, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
                                     0xde, 0x73, 0x7d, 0x56, 0xd3, 0x8b, 0xcf, 0x42, 0x79, 0xdc, 0xe5, 0x61, 0x7e, 0x31, 0x92, 0xa8};
__constant__ uint8_t C_GXY[64] = {0x6b, 0x17, 0xd1, 0xf2, 0xe1, 0x2c, 0x42, 0x47, 0xf8, 0xbc, 0xe6, 0xe5, 0x63, 0xa4, 0x40, 0xf2,
                                  0x77, 0x03, 0x7d, 0x81, 0x2d, 0xeb, 0x33, 0xa0, 0xf4, 0xa1, 0x39, 0x45, 0xd8, 0x98, 0xc2, 0x96,
                                  0x4f, 0xe3, 0x42, 0xe2, 0xfe, 0x1a, 0x7f, 0x9b, 0x8e, 0xe7, 0xeb, 0x4a, 0x7c, 0x0f, 0x9e, 0x16,
                                  0x2b, 0xce, 0x33, 0x57, 0x6b, 0x31, 0x5e, 0xce, 0xcb, 0xb6, 0x40, 0x68, 0x37, 0xbf, 0x51, 0xf5};

// identity bytes -> index of the provider's cache entry (0xFFFFFFFF: not in the table): one coalesced row for the hash (length + last
// 64 bytes), then the ~800 bytes of the SerializedIdentity against the candidate's in 256-byte rows (a dword per lane; the block side
// sits at an arbitrary byte offset: unaligned dword loads, which global memory serves).  Wave-uniform in, wave-uniform out.
__device__ __forceinline__ uint32_t wave_identity_lookup(const WalkArrays& a, const Span id, uint32_t lane) {
    uint32_t found = 0xFFFFFFFFu;
    if (a.id_mask == 0 || a.id_slots == nullptr || id.off > a.arena_len || id.len > a.arena_len - id.off) return found;
    const uint8_t* p = a.block + id.off;
    uint64_t term = bccsp::walk::id_lane_term(p, id.len, lane, a.id_seed);
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t lo32 = __shfl_xor((uint32_t)term, o, 64), hi32 = __shfl_xor((uint32_t)(term >> 32), o, 64);
        term += ((uint64_t)hi32 << 32) | lo32;
    }
    const uint64_t hash = bccsp::walk::id_hash_finish(term, id.len);
    const uint32_t ndw = id.len >> 2, rest = id.len & 3u;
    uint32_t slot = (uint32_t)hash & a.id_mask;
    for (uint32_t probes = 0; probes < bccsp::walk::WALK_ID_PROBE_MAX; probes++) {
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
    return found;
}

// walk::gate_sig_fast (block_walk_core.h) by a whole wavefront: the signature's bytes sit one per lane (two coalesced loads), every
// byte the parse looks at travels by a lane permute, lanes 0..31 assemble r and lanes 32..63 assemble s (one byte each), the
// comparison with n/2 is two ballots.  Returns the same code; `field_byte` is this lane's byte of r (lanes 0..31) or s (32..63) when
// the code is GATE_SUBMIT.  Must be called by all 64 lanes; sig / siglen wave-uniform.
__device__ __forceinline__ uint8_t wave_gate_sig(const uint8_t* sig, uint32_t siglen, uint32_t lane, uint32_t& field_byte) {
    using namespace bccsp::walk;
    field_byte = 0;
    if (siglen == 0) return GATE_EMPTY;
    if (siglen < 8 || siglen > 72) return GATE_DECLINED;
    const uint32_t b0 = lane < siglen ? sig[lane] : 0u, b1 = lane + 64 < siglen ? sig[lane + 64] : 0u;
    auto at = [&](uint32_t pos) -> uint32_t {                          // byte `pos` of the signature (0 beyond its end); pos <= 79
        const uint32_t v0 = __shfl(b0, (int)(pos & 63u), 64), v1 = __shfl(b1, (int)(pos & 63u), 64);
        return pos < 64 ? v0 : v1;
    };
    if (at(0) != 0x30 || at(1) != siglen - 2 || at(2) != 0x02) return GATE_DECLINED;
    uint32_t lr = at(3);
    if (lr < 1 || lr > 33 || 4 + lr + 2 > siglen || at(4 + lr) != 0x02) return GATE_DECLINED;
    uint32_t ls = at(5 + lr);
    uint32_t pr = 4, ps = 6 + lr;
    if (ls < 1 || ls > 33 || 6 + lr + ls != siglen) return GATE_DECLINED;
    const uint32_t r0 = at(pr), r1 = at(pr + 1), s0 = at(ps), s1 = at(ps + 1);
    auto minimal_positive = [](uint32_t p0, uint32_t p1, uint32_t l) {
        if (p0 & 0x80) return false;                                    // negative
        if (p0 == 0) return l > 1 && (p1 & 0x80) != 0 && l <= 33;       // a leading zero must be needed (also excludes zero)
        return l <= 32;
    };
    if (!minimal_positive(r0, r1, lr) || !minimal_positive(s0, s1, ls)) return GATE_DECLINED;
    if (r0 == 0) { pr++; lr--; }
    if (s0 == 0) { ps++; ls--; }
    const uint32_t k = lane & 31u;
    const bool is_s = lane >= 32;
    const uint32_t base = is_s ? ps : pr, L = is_s ? ls : lr;
    const bool inside = k + L >= 32;                                    // leading zero bytes of a short integer otherwise
    const uint32_t v = at(inside ? base + k + L - 32 : 0u);
    field_byte = inside ? v : 0u;
    const uint32_t hn = C_HALF_N[k];
    const uint64_t gt = __ballot(is_s && field_byte > hn), lt = __ballot(is_s && field_byte < hn);
    const uint64_t d = gt | lt;
    if (d == 0) return GATE_SUBMIT;                                     // s == n/2 is low
    const uint64_t first = d & (~d + 1);                                // the most significant differing byte sits in the lowest lane
    return (gt & first) ? GATE_HIGH_S : GATE_SUBMIT;
}

// The gate for EVERY signature (block_walk_core.h gate_sig_any): the wavefront form above for the shape every signer produces, and -
// for whatever it declines - the general parser, run identically by all lanes over the signature's bytes (rare, so its cost does not
// matter; what matters is that no encoding takes the block off the device route).  Never returns GATE_DECLINED.
__device__ __forceinline__ uint8_t wave_gate_sig_any(const uint8_t* sig, uint32_t siglen, uint32_t lane, uint32_t& field_byte) {
    using namespace bccsp::walk;
    uint8_t g = wave_gate_sig(sig, siglen, lane, field_byte);
    if (g != GATE_DECLINED) return g;
    uint32_t pr, lr, ps, ls;
    g = gate_sig_general(sig, siglen, pr, lr, ps, ls);
    field_byte = 0;
    if (g == GATE_SUBMIT) {
        const uint32_t k = lane & 31u, base = lane >= 32 ? ps : pr, L = lane >= 32 ? ls : lr;
        field_byte = k + L >= 32 ? sig[base + k + L - 32] : 0u;
    }
    return g;
}

// ---- an identity nobody has met: SerializedIdentity -> PEM -> DER -> P-256 key, by one wavefront -----------------------------------
// What the host's IdentityToP256 + PublicKeyOnCurve do (block_prepass.cpp; the reference: msp/mspimpl.go:408-421 getIdentityFromConf ->
// pem.Decode, x509.ParseCertificate, the ECDSA public key), so that a block naming identities the provider has never seen - a busy
// network has thousands of client certificates, msp/cache/cache.go keeps 100 - stays on the device route:
//   1. msp.SerializedIdentity{1 mspid, 2 id_bytes}: exactly one field 2 (walk::pb_pick, the host's rule);
//   2. the first "-----BEGIN CERTIFICATE-----" of id_bytes; every character behind it up to the next '-' is a base64 digit, skipped
//      ('=', line ends, blanks) or fatal (PemToDer's classes, walk::pem_char_class) - a character per lane, 64-byte coalesced rows,
//      digits compacted into LDS by ballot ranks; "-----END CERTIFICATE-----" must follow;
//   3. 6-bit digits -> bytes, in place (byte j needs digits 4j/3 and 4j/3 + 1, which lie at or above j);
//   4. walk::cert_der_p256_key_offset over those bytes - the host decoder's own lines;
//   5. x, y < p and y^2 = x^3 - 3x + b (on_curve29).
// IDC_P256: key_byte = this lane's byte of X || Y.  IDC_NOT / IDC_NOT_CERT: the host says "not a P-256 certificate identity" too
// (bccsp/sw decides: other curves / no certificate block at all - idemix, garbage).  A certificate of ANY length is decided from its
// first 3 KiB of DER (the digits behind them are classified and counted, not kept); IDC_UNDECIDED is left for a certificate whose
// SubjectPublicKeyInfo starts beyond that window (kilobytes of issuer / subject names): that one tuple is TUPLE_ST_NEEDS_SW - its
// transaction flag 4, no memo entry - and the block stays on the device route (round 3 sent the whole block to the host walk:
// core/common/validation/msgvalidation.go:258-298 treats a creator per transaction, and so must the pass).
// Called by all 64 lanes of a wavefront, with an LDS buffer of its own.
constexpr uint32_t IDFIX_MAX_DIGITS = 4096;                             // 3 KiB of DER
// One wavefront owns its LDS buffer: its LDS instructions execute in order, so all that is needed between "these lanes wrote" and
// "those lanes read" is that the compiler keeps the order (no workgroup barrier: the other wavefronts of the workgroup are elsewhere).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
enum : uint8_t { IDC_P256 = 0, IDC_NOT = 1, IDC_UNDECIDED = 2, IDC_NOT_CERT = 3 };   // NOT_CERT: no PEM certificate block at all (idemix, garbage)
__constant__ char C_PEM_BEGIN[28] = "-----BEGIN CERTIFICATE-----";
__constant__ char C_PEM_END[26] = "-----END CERTIFICATE-----";

__device__ uint8_t wave_identity_to_p256(const uint8_t* ident, uint32_t len, uint32_t lane, uint8_t* lds, uint32_t& key_byte) {
    using namespace bccsp::walk;
    key_byte = 0;
    Pick w(2);
    if (!pb_pick(ident, len, &w, 1) || w.seen != 1) return IDC_NOT_CERT;
    const uint8_t* pem = w.p;
    const uint32_t pl = (uint32_t)w.len;
    // the first BEGIN marker (PemToDer scans for it): candidates are the dashes of a 64-byte row, each checked by 27 lanes at once
    uint32_t start = 0xFFFFFFFFu;
    for (uint32_t base = 0; base + 27 <= pl && start == 0xFFFFFFFFu; base += 64) {
        const uint32_t pos = base + lane;
        uint64_t cand = __ballot(pos + 27 <= pl && pem[pos] == '-');
        while (cand) {
            const uint32_t at = base + (uint32_t)__builtin_ctzll(cand);
            cand &= cand - 1;
            if (__ballot(lane < 27 && pem[at + (lane < 27 ? lane : 0)] != (uint8_t)C_PEM_BEGIN[lane < 27 ? lane : 0]) == 0) {
                start = at + 27;
                break;
            }
        }
    }
    if (start == 0xFFFFFFFFu) return IDC_NOT_CERT;
    // The body, four 64-byte rows per step: their loads are issued together (a row's rank needs the digit count of the rows before
    // it, its load does not), then the rows are classified and compacted one after the other.
    uint32_t total = 0, end_pos = 0xFFFFFFFFu;
    for (uint32_t base = start; base < pl && end_pos == 0xFFFFFFFFu; base += 256) {
        int cls[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t pos = base + 64 * r + lane;
            cls[r] = pos < pl ? pem_char_class(pem[pos]) : (int)PEM_SKIP;
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (end_pos != 0xFFFFFFFFu || base + 64 * r >= pl) break;
            const uint64_t dash = __ballot(cls[r] == PEM_DASH);
            const uint32_t limit = dash ? (uint32_t)__builtin_ctzll(dash) : 64u;
            const bool act = lane < limit;
            if (__ballot(act && cls[r] == PEM_INVALID)) return IDC_NOT;
            const bool digit = act && cls[r] < 64;
            const uint64_t dig = __ballot(digit);
            const uint32_t rank = total + (uint32_t)__builtin_popcountll(dig & ((1ull << lane) - 1ull));
            // (digits beyond the buffer are classified and counted - the certificate's length enters the DER walk - but not kept: what
            //  the walk needs, everything up to SubjectPublicKeyInfo, lies at the front)
            if (digit && rank < IDFIX_MAX_DIGITS) lds[rank] = (uint8_t)cls[r];
            total += (uint32_t)__builtin_popcountll(dig);
            // (a lone wavefront reading on through megabytes of PEM would hold the whole gate kernel up: beyond four windows - 12 KiB of
            //  DER, several times any real certificate - the tuple is left to bccsp/sw)
            if (total > 4 * IDFIX_MAX_DIGITS) return IDC_UNDECIDED;
            if (dash) end_pos = base + 64 * r + limit;
        }
    }
    if (end_pos == 0xFFFFFFFFu || end_pos + 25 > pl) return IDC_NOT;
    if (__ballot(lane < 25 && pem[end_pos + (lane < 25 ? lane : 0)] != (uint8_t)C_PEM_END[lane < 25 ? lane : 0])) return IDC_NOT;
    const uint32_t nder_all = total * 6 / 8;                             // bytes of the whole certificate
    const uint32_t nder = (total < IDFIX_MAX_DIGITS ? total : IDFIX_MAX_DIGITS) * 6 / 8;   // ... of which so many are decoded into LDS
    if (nder_all == 0) return IDC_NOT;
    wave_lds_sync();
    for (uint32_t j0 = 0; j0 < nder; j0 += 64) {
        const uint32_t j = j0 + lane;
        uint32_t byte = 0;
        if (j < nder) {
            const uint32_t q = 4 * j / 3, sh = 4 - 2 * ((4 * j) % 3);       // bit 8j of the digit stream = bit (8j mod 6) of digit q
            byte = ((((uint32_t)lds[q] << 6) | lds[q + 1]) >> sh) & 0xFFu;
        }
        wave_lds_sync();                                                 // (all of this row's digits are read before its bytes land on them)
        if (j < nder) lds[j] = (uint8_t)byte;
    }
    wave_lds_sync();
    const int32_t at = cert_der_p256_key_offset_window(lds, nder, nder_all);
    if (at == -2) return IDC_UNDECIDED;                                  // SubjectPublicKeyInfo beyond the first 3 KiB of DER: bccsp/sw decides THIS tuple
    if (at < 0) return IDC_NOT;
    u256 x, y;
    from_be32(x, lds + at);
    from_be32(y, lds + at + 32);
    const u256 P = FAB_P256_P;
    bool ok = lt256(x, P) & lt256(y, P);
    fe mx, my;
    fe_to_mont(mx, x);
    fe_to_mont(my, y);
    ok = ok & on_curve29(mx, my);
    if (!ok) return IDC_NOT;
    key_byte = lds[at + lane];
    return IDC_P256;
}

// The identity, the gates of one tuple and its row of the submission arrays.  A tuple the device does not decide still gets a
// well-formed row (r = s = 1; its identity's key, or the generator when there is none): there is no compaction and its verdict is
// ignored in favour of gate_st.  An identity the table does not hold is NOT a reason to give the block up: its certificate is decoded
// right here (wave_identity_to_p256: the wavefront's own 4 KiB of LDS - 16 KiB per workgroup, which does not cost an occupancy slot)
// and the identity is offered to the provider's cache; waves that looked their identity up are long gone meanwhile.
__global__ void __launch_bounds__(256) walk_gate_kernel(WalkArrays a) {
    __shared__ uint8_t lds_all[4 * IDFIX_MAX_DIGITS];
    const uint32_t lane = threadIdx.x & 63u;
    // (gate_mode 1: one wavefront per ENVELOPE, its creator tuple only - what the nym launch and the creators' launch wait for; 2: all
    //  tuples but those; 0: everything in one launch)
    uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (a.gate_mode == 1) {
        if (i >= a.n_env || a.counts[i].x == 0) return;
        i = a.bases[i].x;
    }
    if (i >= a.n_tuples) return;
    uint8_t* lds = lds_all + (threadIdx.x >> 6) * IDFIX_MAX_DIGITS;
    const BlockTuple t = a.tuples[i];
    if (a.gate_mode == 2 && i < a.n_dev_tuples && i == a.bases[t.tx].x) return;
    uint32_t idx = wave_identity_lookup(a, t.identity, lane);
    // the row of this tuple (WalkArrays::row_of): creators first when the submission is split
    const bool creator = i < a.n_dev_tuples && i == a.bases[t.tx].x;
    uint32_t row = i;
    if (a.split) {
        // creator tuples at indices <= i: those of the envelopes before this one, plus this envelope's (its first tuple)
        const uint32_t before = i < a.n_dev_tuples ? a.cbase[t.tx] : a.n_creators;
        row = creator ? before : a.n_creators + (i - before - (i < a.n_dev_tuples ? 1u : 0u));
    }
    const DevIdEntry* ent = idx != 0xFFFFFFFFu ? a.id_entries + idx : nullptr;
    // ---- whose key? ----
    const bool unknown = ent == nullptr;
    bool p256 = ent && ent->p256, undecided = false;
    uint32_t key_byte = p256 ? (lane < 32 ? ent->qx[lane & 31u] : ent->qy[lane & 31u]) : C_GXY[lane];
    if (unknown) {
        uint8_t code = IDC_NOT_CERT;
        uint32_t kb = 0;
        if (t.identity.off <= a.arena_len && t.identity.len <= a.arena_len - t.identity.off)
            code = wave_identity_to_p256(a.block + t.identity.off, t.identity.len, lane, lds, kb);
        undecided = code == IDC_UNDECIDED;
        if (code == IDC_P256) {
            p256 = true;
            key_byte = kb;
            idx = 0xFFFFFFFEu;                                           // "decoded here"
        }
        if (code == IDC_P256 || code == IDC_NOT) {
            // Offer it to the provider's cache (the host route caches what it decodes, keys and "this certificate has no P-256 key"
            // alike; identities without a certificate block - a fresh idemix pseudonym per transaction - would only churn it): one
            // slot per table hash; whoever names the same identity after the slot's owner counts as a hit - a comb table is earned by
            // being named often.
            uint64_t term = bccsp::walk::id_lane_term(a.block + t.identity.off, t.identity.len, lane, a.id_seed);
            for (int o = 32; o >= 1; o >>= 1) {
                const uint32_t lo32 = __shfl_xor((uint32_t)term, o, 64), hi32 = __shfl_xor((uint32_t)(term >> 32), o, 64);
                term += ((uint64_t)hi32 << 32) | lo32;
            }
            const uint64_t tag = bccsp::walk::id_hash_finish(term, t.identity.len) | 1ull;
            // Open addressing, up to eight probes: round 4 gave every table hash ONE slot, first come first served - two of a channel's
            // six signers met in a slot in one fresh provider of nine (128 slots, six keys: 11 %), the loser was learned a block later,
            // and that block paid a relaunch on the fresh-key kernels plus 6 ms of table building (lone passes 15.5 / 9.1 / 2.2 ms;
            // DESIGN.md 4.4d "Round 5").  With eight probes a block has to bring dozens of new identities before one waits.
            const uint32_t home = (uint32_t)(tag >> 17) & (WALK_LEARN_SLOTS - 1);
            uint32_t mine = 0, at = home;
            if (lane == 0) {
                for (uint32_t probe = 0; probe < 8u; probe++) {
                    WalkLearn* cand = a.learn + ((home + probe) & (WALK_LEARN_SLOTS - 1));
                    unsigned long long old = __builtin_nontemporal_load((const unsigned long long*)&cand->tag);
                    if (old == 0ull) old = atomicCAS((unsigned long long*)&cand->tag, 0ull, (unsigned long long)tag);
                    if (old == 0ull) {                                         // an empty slot: this wavefront owns it now
                        mine = 1u;
                        at = (home + probe) & (WALK_LEARN_SLOTS - 1);
                        break;
                    }
                    if (old == (unsigned long long)tag) {                      // this identity's slot, owned by an earlier tuple: a hit
                        // (hits only matter up to the registration threshold: a block naming one new identity thousands of times stops counting early)
                        if (__builtin_nontemporal_load(&cand->hits) < 256u) atomicAdd(&cand->hits, 1u);
                        break;
                    }
                }                                                              // (eight other identities in a row: shows up again in the next block)
            }
            at = __shfl(at, 0, 64);
            WalkLearn* slot = a.learn + at;
            mine = __shfl(mine, 0, 64);
            if (mine) {
                (lane < 32 ? slot->qx : slot->qy)[lane & 31u] = (uint8_t)key_byte;
                if (lane == 0) {
                    slot->off = t.identity.off;
                    slot->len = t.identity.len;
                    atomicAdd(&slot->hits, 1u);
                    __threadfence();
                    slot->ready = code == IDC_P256 ? 1u : 2u;
                    atomicAdd(&a.summary->n_learn, 1u);
                }
            }
        }
    }
    // ---- an idemix creator?  (msp/idemixmsp.go:584-599: identity.Verify is NymSignature.Ver under the MSP's issuer key; idemix
    // identities are creators only, docs/source/idemix.rst:171-176.)  The host's rule and order: an identity of the idemix shape that does
    // not ALSO yield a P-256 certificate key, under a registered MSP id, with a NymSignature of four 32-byte fields -> the nym kernel;
    // of the idemix shape but anything else -> bccsp/idemix decides (TUPLE_ST_NEEDS_SW).
    bool nym = false;
    if (a.n_idemix_msps && creator && unknown && !p256 && t.identity.off <= a.arena_len && t.identity.len <= a.arena_len - t.identity.off) {
        bccsp::walk::IdemixNymRef ref;
        if (bccsp::walk::identity_to_idemix_nym(a.block + t.identity.off, t.identity.len, ref)) {
            int32_t issuer = -1;
            for (uint32_t m = 0; m < a.n_idemix_msps; m++) {
                const DevIdemixMsp& ms = a.idemix_msps[m];
                if (ms.len != ref.mspid_len) continue;
                bool same = true;
                for (uint32_t k = 0; k < ms.len; k++) same = same && ms.id[k] == ref.mspid[k];
                if (same) issuer = ms.issuer;
            }
            const uint8_t* sf[4];
            const bool sig_ok = t.sig.len != 0 && t.sig.off <= a.arena_len && t.sig.len <= a.arena_len - t.sig.off &&
                                bccsp::walk::unmarshal_nym_signature32(a.block + t.sig.off, t.sig.len, sf);
            if (issuer >= 0 && sig_ok) {
                nym = true;
                const uint32_t rank = a.cbase[t.tx];                       // this creator's row
                const size_t col = (size_t)32 * a.n_creators;
                const uint8_t* src[6] = {ref.nx, ref.ny, sf[0], sf[1], sf[2], sf[3]};
                if (lane < 32) {
#pragma unroll
                    for (int c = 0; c < 6; c++) a.nym_fields[c * col + 32 * (size_t)rank + lane] = src[c][lane];
                }
                key_byte = lane < 32 ? ref.nx[lane] : ref.ny[lane & 31u];    // the pseudonym in the key slot (what the memo keys on)
                if (lane == 0) {
                    a.nym_issuer[rank] = (uint32_t)issuer;
                    a.nym_issuer_out[rank] = issuer;
                    a.nym_spans[2 * (size_t)rank] = t.suffix.len ? t.suffix.off : 0;
                    a.nym_spans[2 * (size_t)rank + 1] = t.suffix.len ? t.suffix.off + t.suffix.len : 0;
                }
            }
        }
    }
    // ---- the signature ----
    uint8_t gst;
    bool submit = false, general = false;
    uint32_t field_byte = 0;
    // a creator's digest may have been computed from the host's outline of the envelope (WalkArrays::early_creator_hash): the
    // walker's own idea of that message must be the same bytes, or this pass does not answer
    bool outline_differs = false;
    if (a.early_creator_hash && creator) {
        const uint32_t p0 = a.payload_spans[2 * (size_t)t.tx], p1 = a.payload_spans[2 * (size_t)t.tx + 1];
        outline_differs = (t.suffix.len ? t.suffix.off : 0u) != p0 || (t.suffix.len ? t.suffix.off + t.suffix.len : 0u) != p1;
    }
    if (nym) {
        gst = GATE_ST_NYM;
    } else if (!p256) {
        gst = bccsp::TUPLE_ST_NEEDS_SW;
    } else if (t.sig.len == 0) {
        gst = bccsp::TUPLE_ST_EMPTY_SIG;
    } else {
        uint8_t g = bccsp::walk::GATE_BAD_DER;
        if (t.sig.off <= a.arena_len && t.sig.len <= a.arena_len - t.sig.off) {
            g = wave_gate_sig(a.block + t.sig.off, t.sig.len, lane, field_byte);
            if (g == bccsp::walk::GATE_DECLINED) {
                general = true;
                g = wave_gate_sig_any(a.block + t.sig.off, t.sig.len, lane, field_byte);
            }
        }
        if (g == bccsp::walk::GATE_SUBMIT) {
            gst = FABGPU_ST_VALID;
            submit = true;
        } else if (g == bccsp::walk::GATE_HIGH_S) {
            gst = FABGPU_ST_HIGH_S;
        } else if (g == bccsp::walk::GATE_RANGE) {
            gst = FABGPU_ST_RANGE;                                  // r of more than 256 bits: ecdsa.Verify's (false, nil)
        } else {
            gst = bccsp::TUPLE_ST_BAD_DER;
        }
    }
    const bool keyed = submit && ent && ent->key_id >= 0;
    // r | s and qx | qy: one byte per lane, 64-byte coalesced rows
    const uint32_t k = lane & 31u;
    const uint8_t rs = submit ? (uint8_t)field_byte : (uint8_t)(k == 31 ? 1 : 0);
    (lane < 32 ? a.r : a.s)[32 * (size_t)row + k] = rs;
    // (an idemix creator's ECDSA row stays the filler - generator key, r = s = 1 - while its pseudonym travels to the status kernel in the
    // nym columns)
    (lane < 32 ? a.qx : a.qy)[32 * (size_t)row + k] = nym ? C_GXY[lane] : (uint8_t)key_byte;
    if (lane == 0) {
        a.id_idx[i] = idx;
        a.row_of[i] = row;
        a.off2[2 * (size_t)row] = t.suffix.len ? t.suffix.off : 0;
        a.off2[2 * (size_t)row + 1] = t.suffix.len ? t.suffix.off + t.suffix.len : 0;
        a.pre_idx[row] = t.prefix_index >= 0 ? (uint32_t)t.prefix_index : 0xFFFFFFFFu;
        a.key_id[row] = keyed ? (uint32_t)ent->key_id : 0u;
        a.gate_st[i] = gst;
        // What the summary wants to know about this tuple travels as a byte; walk_status_kernel turns the bytes into one atomic per
        // WAVEFRONT.  (An atomic per tuple on the summary's words - ten thousand unknown creators, ten thousand increments of one
        // address - serialises at the memory side: the gate kernel took 1.6 ms instead of 0.15, and the hash kernels beside it 1 ms.)
        a.tflags[i] = (uint8_t)((submit ? TF_SUBMIT : 0) | (keyed ? TF_KEYED : 0) | (unknown ? TF_UNKNOWN : 0) | (general ? TF_GENERAL : 0) |
                                (outline_differs ? TF_OUTLINE : 0) | (undecided ? TF_UNDECIDED : 0) | (nym ? TF_NYM : 0));
    }
}

// TEST HOOK: the identity decoder alone over n identities (spans into arena) -> code, key per identity
__global__ void __launch_bounds__(64) walk_idfix_probe_kernel(uint32_t n, const uint8_t* __restrict__ arena, const uint32_t* __restrict__ spans,
                                                              uint8_t* __restrict__ code, uint8_t* __restrict__ key) {
    __shared__ uint8_t lds[IDFIX_MAX_DIGITS];
    const uint32_t lane = threadIdx.x, i = blockIdx.x;
    if (i >= n) return;
    uint32_t kb = 0;
    const uint8_t c = wave_identity_to_p256(arena + spans[2 * i], spans[2 * i + 1] - spans[2 * i], lane, lds, kb);
    key[64 * (size_t)i + lane] = c == IDC_P256 ? (uint8_t)kb : 0;
    if (lane == 0) code[i] = c == IDC_NOT_CERT ? (uint8_t)IDC_NOT : c;
}

// TEST HOOK: the wavefront gate alone over n signatures (spans into arena) -> code, r, s per signature
__global__ void __launch_bounds__(256) walk_gate_probe_kernel(uint32_t n, const uint8_t* __restrict__ arena, const uint32_t* __restrict__ spans,
                                                               uint8_t* __restrict__ code, uint8_t* __restrict__ r, uint8_t* __restrict__ s) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= n) return;
    uint32_t fb = 0;
    const uint8_t g = wave_gate_sig_any(arena + spans[2 * i], spans[2 * i + 1] - spans[2 * i], lane, fb);
    (lane < 32 ? r : s)[32 * (size_t)i + (lane & 31u)] = g == bccsp::walk::GATE_SUBMIT ? (uint8_t)fb : 0;
    if (lane == 0) code[i] = g;
}

// The creators' payload digests were computed per ENVELOPE (from the host's outline, before anything was walked); the creators' launch
// reads them per ROW: row = the number of tuple-yielding envelopes before this one.
__global__ void __launch_bounds__(256) walk_creator_digest_kernel(WalkArrays a, uint8_t* __restrict__ row_digests) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_env || a.counts[e].x == 0) return;
    const uint4* src = reinterpret_cast<const uint4*>(a.digest_env + 32 * (size_t)e);
    uint4* dst = reinterpret_cast<uint4*>(row_digests + 32 * (size_t)a.cbase[e]);
    dst[0] = src[0];
    dst[1] = src[1];
}

// per-transaction evidence bits
enum : uint32_t { M_BAD_CREATOR = 1, M_BAD_END = 2, M_SW = 4, M_BAD_TXID = 8, M_BAD_PHASH = 16 };

__device__ __forceinline__ void walk_status_part(const WalkArrays& a, uint32_t i, uint32_t* part) {
    const bool live = i < a.n_tuples;
    uint8_t hashed = 0;
    bool creator = false, by_ecdsa = false;
    if (live) {
        const uint8_t gst = a.gate_st[i];
        const BlockTuple t = a.tuples[i];
        uint8_t st = gst;
        const uint32_t row = a.row_of[i];
        creator = i < a.n_dev_tuples && i == a.bases[t.tx].x;
        if (a.row_digests) {
            const uint4* src = reinterpret_cast<const uint4*>(a.row_digests + 32 * (size_t)row);
            uint4* dst = reinterpret_cast<uint4*>(a.tuple_digests + 32 * (size_t)i);
            dst[0] = src[0];
            dst[1] = src[1];
        }
        bool key_from_nym = false;
        if (gst == GATE_ST_NYM) {
            // An idemix creator: the nym kernel's answer for its row (FABGPU_NYM_VALID = 0, BAD_PROOF = 1 = "signature invalid",
            // NEEDS_SW = 6 = TUPLE_ST_NEEDS_SW: exactly the host pass's mapping).  Its memo entry is keyed on the pseudonym and on
            // SHA-256(message) - the creator's payload digest, which this row already holds - so it counts as hashed, and carries its
            // key, only for a caller that asked for digests (as on the host route).
            const uint32_t slot = a.nym_status ? a.nym_slot[a.cbase[t.tx]] : 0xFFFFFFFFu;   // (a row past the launch's capacity: not decided -
            st = slot < a.nym_cap ? a.nym_status[slot] : (uint8_t)bccsp::TUPLE_ST_NEEDS_SW;    //  the host sees n_nym > capacity and launches again)
            if (a.row_digests && st != bccsp::TUPLE_ST_NEEDS_SW) {
                hashed = 1;
                key_from_nym = true;
            }
        }
        if (a.tuple_qxy) {                                           // the key of the tuple's identity (zeros when it has none to report)
            const bool p256 = gst != bccsp::TUPLE_ST_NEEDS_SW && gst != GATE_ST_NYM;
            const uint4 z = make_uint4(0, 0, 0, 0);
            const uint4* sx = reinterpret_cast<const uint4*>(a.qx + 32 * (size_t)row);
            const uint4* sy = reinterpret_cast<const uint4*>(a.qy + 32 * (size_t)row);
            if (key_from_nym) {
                const uint32_t rank = a.cbase[t.tx];
                sx = reinterpret_cast<const uint4*>(a.nym_fields + 32 * (size_t)rank);
                sy = reinterpret_cast<const uint4*>(a.nym_fields + 32 * (size_t)a.n_creators + 32 * (size_t)rank);
            }
            uint4* dst = reinterpret_cast<uint4*>(a.tuple_qxy + 64 * (size_t)i);
            const bool have = p256 || key_from_nym;
            dst[0] = have ? sx[0] : z;
            dst[1] = have ? sx[1] : z;
            dst[2] = have ? sy[0] : z;
            dst[3] = have ? sy[1] : z;
        }
        if (gst == FABGPU_ST_VALID) {                                    // the device decided: exactly PreVerifyParsed's mapping
            const bool in_c = a.split && row < a.n_creators;
            const uint32_t j = in_c ? row : row - (a.split ? a.n_creators : 0u);
            const bool from_all = in_c ? a.all_creators != 0 : a.all_others != 0;
            const bool bit = from_all ? ((a.verdict_bits_all[row >> 6] >> (row & 63)) & 1) : (((in_c ? a.verdict_bits_c : a.verdict_bits)[j >> 6] >> (j & 63)) & 1);
            const uint8_t ds = a.dev_status[row];
            st = (bit && ds == FABGPU_ST_VALID) ? FABGPU_ST_VALID : (ds == FABGPU_ST_VALID ? FABGPU_ST_BAD_MATH : ds);
            hashed = 1;
            by_ecdsa = true;
        }
        a.tuple_status[i] = st;
        a.tuple_hashed[i] = hashed;
        if (st != FABGPU_ST_VALID && t.tx < a.n_env)                     // block-level tuples (orderer signatures) do not flag a transaction
            atomicOr(&a.tx_mask[t.tx], st == bccsp::TUPLE_ST_NEEDS_SW ? M_SW : (t.kind == bccsp::TUPLE_CREATOR ? M_BAD_CREATOR : M_BAD_END));
    }
    // The summary: how many tuples each launch class decided (the provider reports how many went through registered comb tables), and what
    // the gate kernel noted per tuple - added up per WORKGROUP in LDS (`part`, WalkSummary's words) and stored, not added, to the
    // workgroup's row of summary_parts; the finish kernel sums the rows.  (One global atomic per wavefront and word it was, all into one
    // 48-byte line - 625 wavefronts x 3 words for a friendly 10 000-tx block: this kernel 40 -> 23 us.)
    const uint8_t tf = live ? a.tflags[i] : 0;
    const uint64_t hc = __ballot(by_ecdsa && creator), ho = __ballot(by_ecdsa && !creator);
    const uint64_t unk = __ballot((tf & TF_UNKNOWN) != 0), und = __ballot((tf & TF_UNDECIDED) != 0), gen = __ballot((tf & TF_GENERAL) != 0),
                   outl = __ballot((tf & TF_OUTLINE) != 0), sub = __ballot((tf & TF_SUBMIT) != 0), nym = __ballot((tf & TF_NYM) != 0),
                   ukc = __ballot((tf & (TF_SUBMIT | TF_KEYED)) == TF_SUBMIT && creator), uko = __ballot((tf & (TF_SUBMIT | TF_KEYED)) == TF_SUBMIT && !creator);
    if ((threadIdx.x & 63u) == 0) {
        auto add = [&](size_t word, uint64_t m) {
            if (m) atomicAdd(&part[word], (uint32_t)__builtin_popcountll(m));
        };
        add(offsetof(WalkSummary, n_hashed_creator) / 4, hc);
        add(offsetof(WalkSummary, n_hashed_other) / 4, ho);
        add(offsetof(WalkSummary, n_unknown_identity) / 4, unk);
        add(offsetof(WalkSummary, n_undecided) / 4, und);
        add(offsetof(WalkSummary, n_general_der) / 4, gen);
        add(offsetof(WalkSummary, n_outline_differs) / 4, outl);
        add(offsetof(WalkSummary, n_submitted) / 4, sub);
        add(offsetof(WalkSummary, n_nym) / 4, nym);
        add(offsetof(WalkSummary, n_unkeyed_creator) / 4, ukc);
        add(offsetof(WalkSummary, n_unkeyed_other) / 4, uko);
    }
}

// does the digest equal what the block says (block_prepass.cpp HashCheckMatches: lowercase hex for the TxID, raw bytes for the proposal hash)
__device__ __forceinline__ void walk_checks_part(const WalkArrays& a, uint32_t n_checks, uint32_t j) {
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

// statuses (workgroups [0, ceil(n_tuples / 256))) and digest comparisons (the workgroups behind them) as ONE launch: both only feed tx_mask
__global__ void __launch_bounds__(256) walk_status_checks_kernel(WalkArrays a, uint32_t n_checks, uint32_t status_blocks) {
    constexpr uint32_t W = sizeof(WalkSummary) / 4;
    __shared__ uint32_t part[W];
    if (blockIdx.x < status_blocks) {                                     // (wavefront-uniform: a workgroup is one or the other)
        if (threadIdx.x < W) part[threadIdx.x] = 0;
        __syncthreads();
        walk_status_part(a, blockIdx.x * blockDim.x + threadIdx.x, part);
        __syncthreads();
        if (threadIdx.x < W) a.summary_parts[(size_t)blockIdx.x * W + threadIdx.x] = part[threadIdx.x];
    } else {
        walk_checks_part(a, n_checks, (blockIdx.x - status_blocks) * blockDim.x + threadIdx.x);
    }
}

// The last kernel of a pass: the flag of every transaction - in the order ValidateTransaction and then VSCC would reject
// (PreVerifyParsed): not understood > bad creator signature > bad TxID > bad proposal hash > bad endorsement > "ask bccsp/sw" > all
// valid - and everything the host reads, stored straight into host-mapped memory (WalkHostOut: four bytes per lane, coalesced rows
// over PCIe); the workgroup that finishes last raises the flag the host polls.
// rows 4 j .. 4 j + 3 of everything the host reads
__device__ __forceinline__ void walk_finish_rows(const WalkArrays& a, const WalkHostOut& h, uint32_t j) {
    const uint32_t t0 = 4 * j;
    if (t0 < a.n_env) {
        uint32_t packed = 0;
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t t = t0 + k;
            uint8_t f = 0;
            if (t < a.n_env) {
                const uint32_t m = a.tx_mask[t];
                if (!a.tx_understood[t]) f = bccsp::TX_NOT_UNDERSTOOD;
                else if (m & M_BAD_CREATOR) f = bccsp::TX_BAD_CREATOR_SIGNATURE;
                else if (m & M_BAD_TXID) f = bccsp::TX_BAD_TXID;
                else if (m & M_BAD_PHASH) f = bccsp::TX_BAD_PROPOSAL_HASH;
                else if (m & M_BAD_END) f = bccsp::TX_BAD_ENDORSEMENT;
                else if (m & M_SW) f = bccsp::TX_NEEDS_SW;
                else f = bccsp::TX_ALL_SIGNATURES_VALID;
                a.tx_flags[t] = f;
            }
            packed |= (uint32_t)f << (8 * k);
        }
        // (the byte arrays are padded to 256 bytes on both sides: whole dwords may be read and written)
        reinterpret_cast<uint32_t*>(h.tx_flags)[j] = packed;
        reinterpret_cast<uint32_t*>(h.tx_type)[j] = reinterpret_cast<const uint32_t*>(a.tx_type)[j];
        reinterpret_cast<uint32_t*>(h.tx_understood)[j] = reinterpret_cast<const uint32_t*>(a.tx_understood)[j];
    }
    if (t0 < a.n_tuples) {
        reinterpret_cast<uint32_t*>(h.tuple_status)[j] = reinterpret_cast<const uint32_t*>(a.tuple_status)[j];
        reinterpret_cast<uint32_t*>(h.tuple_hashed)[j] = reinterpret_cast<const uint32_t*>(a.tuple_hashed)[j];
        for (uint32_t k = 0; k < 4; k++)
            if (t0 + k < a.n_tuples) h.id_idx[t0 + k] = a.id_idx[t0 + k];
    }
}
// one workgroup's job: the learn records, the summary, the memo's totals.  tot: sizeof(WalkSummary) / 4 words of LDS; at least 252 threads.
__device__ __forceinline__ void walk_finish_summary(const WalkArrays& a, const WalkHostOut& h, uint32_t* tot) {
    const uint32_t* ls = reinterpret_cast<const uint32_t*>(a.learn);
    uint32_t* ld = reinterpret_cast<uint32_t*>(h.learn);
    for (uint32_t w = threadIdx.x; w < sizeof(WalkLearn) * WALK_LEARN_SLOTS / 4; w += blockDim.x) ld[w] = ls[w];
    // the summary = what single kernels added to it (emit: count checks; gate: learn slots) + the status workgroups' rows, summed here by
    // 252 threads (word w, every 21st row) through LDS
    constexpr uint32_t W = sizeof(WalkSummary) / 4;
    if (threadIdx.x < W) tot[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.summary)[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < W * 21) {
        const uint32_t w = threadIdx.x % W, c = threadIdx.x / W, rows = (a.n_tuples + 255) / 256;
        uint32_t v = 0;
        for (uint32_t r = c; r < rows; r += 21) v += a.summary_parts[(size_t)r * W + w];
        if (v) atomicAdd(&tot[w], v);
    }
    __syncthreads();
    if (threadIdx.x < W) reinterpret_cast<uint32_t*>(h.summary)[threadIdx.x] = tot[threadIdx.x];
    if (h.memo_totals && a.memo_totals && threadIdx.x < sizeof(WalkMemoTotals) / 4)
        reinterpret_cast<uint32_t*>(h.memo_totals)[threadIdx.x] = reinterpret_cast<const uint32_t*>(a.memo_totals)[threadIdx.x];
}
__device__ __forceinline__ void walk_finish_raise(const WalkHostOut& h) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = atomicAdd(h.done, 1u);
        if (before == gridDim.x - 1) {
            __threadfence_system();
            __hip_atomic_store(h.flag, h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void __launch_bounds__(256) walk_finish_kernel(WalkArrays a, WalkHostOut h) {
    walk_finish_rows(a, h, blockIdx.x * blockDim.x + threadIdx.x);
    if (blockIdx.x == 0) {
        __shared__ uint32_t tot[sizeof(WalkSummary) / 4];
        walk_finish_summary(a, h, tot);
    }
    walk_finish_raise(h);
}

// A block of a few hundred transactions: statuses, digest comparisons AND the finish in ONE launch - the two launches above are 15 and
// 10 us of work with a launch gap between them that is as long, at the very end of a pass's critical path.  The workgroups do what
// walk_status_checks_kernel's do; the one that is through LAST (a counter: h.done[1]) goes on and does the finish for the whole block -
// at most WALK_SMALL_FINISH_MAX rows, four rounds of its 256 threads.  (No memo: its late half sits between the two.)
constexpr uint32_t WALK_SMALL_FINISH_MAX = 4096;                           // tuples, transactions
__global__ void __launch_bounds__(256) walk_status_finish_small_kernel(WalkArrays a, uint32_t n_checks, uint32_t status_blocks, WalkHostOut h) {
    constexpr uint32_t W = sizeof(WalkSummary) / 4;
    __shared__ uint32_t part[W], tot[W];
    __shared__ uint32_t am_last;
    if (blockIdx.x < status_blocks) {
        if (threadIdx.x < W) part[threadIdx.x] = 0;
        __syncthreads();
        walk_status_part(a, blockIdx.x * blockDim.x + threadIdx.x, part);
        __syncthreads();
        if (threadIdx.x < W) a.summary_parts[(size_t)blockIdx.x * W + threadIdx.x] = part[threadIdx.x];
    } else {
        walk_checks_part(a, n_checks, (blockIdx.x - status_blocks) * blockDim.x + threadIdx.x);
    }
    __threadfence();                                                       // this workgroup's tx_mask bits, statuses, summary row: out
    __syncthreads();
    if (threadIdx.x == 0) am_last = atomicAdd(h.done + 1, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!am_last) return;
    __threadfence();                                                       // ... and everybody else's: in
    const uint32_t most = a.n_env > a.n_tuples ? a.n_env : a.n_tuples;
    for (uint32_t j = threadIdx.x; 4 * j < most; j += blockDim.x) walk_finish_rows(a, h, j);
    walk_finish_summary(a, h, tot);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h.flag, h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace


// The idemix creators, counted off for the nym launch: rank -> row = the number of active ranks below it (one workgroup, the scan of
// walk_scan_kernel), and the launch's list row -> rank.  Creators whose row is past `cap` stay out (the host launches again with the
// capacity the summary names).
__global__ void __launch_bounds__(1024) walk_nym_pack_kernel(WalkArrays a, uint32_t* __restrict__ gather, uint32_t cap) {
    __shared__ uint32_t sk[1024];
    const uint32_t tid = threadIdx.x, n = a.n_creators;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = tid * per < n ? tid * per : n, hi = lo + per < n ? lo + per : n;
    uint32_t k = 0;
    for (uint32_t i = lo; i < hi; i++) k += a.nym_issuer_out[i] >= 0 ? 1u : 0u;
    sk[tid] = k;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {
        uint32_t v = 0;
        if (tid >= o) v = sk[tid - o];
        __syncthreads();
        sk[tid] += v;
        __syncthreads();
    }
    uint32_t slot = sk[tid] - k;
    for (uint32_t i = lo; i < hi; i++) {
        const bool on = a.nym_issuer_out[i] >= 0;
        a.nym_slot[i] = on ? slot : 0xFFFFFFFFu;
        if (on && slot < cap) gather[slot] = i;
        slot += on ? 1u : 0u;
    }
}


// ---- the block's verdict memo, built where the verdicts are (bccsp_host.h BlockMemo; GPUCSP::SeedMemo is the host's version) -------------
// In two halves.  EARLY, beside the verify launches: everything of an entry that is known once the gates are through - which tuples are
// candidates (submitted to the device, or a pseudonym signature with its issuer's hash at hand; a signature of 1 .. 1024 bytes), their
// framed keys WITHOUT the digest ([1 | 2 || issuer hash] || X || Y || u32 len || signature || u32 32: GPUCSP::MemoKeyWrite up to the digest),
// entry indices and offsets - and its copy to the host.  LATE, behind the status kernel: digest, status byte and slot of the candidates that
// were hashed and decided with a status bccsp.Verify decides itself (GPUCSP::SeedMemo's rule); a candidate that was not simply gets no slot.
__device__ __forceinline__ int32_t memo_issuer_slot(const WalkArrays& a, const BlockTuple& t) {
    if (!a.issuer_hashes || !a.nym_issuer_out) return -1;
    const int32_t issuer = a.nym_issuer_out[a.cbase[t.tx]];
    int32_t at = -1;
    for (uint32_t m = 0; m < a.n_idemix_msps; m++)
        if (issuer >= 0 && a.idemix_msps[m].issuer == issuer) at = (int32_t)m;
    return at;
}
__global__ void __launch_bounds__(256) walk_memo_len_kernel(WalkArrays a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_tuples) return;
    const uint8_t gst = a.gate_st[i];
    const BlockTuple t = a.tuples[i];
    const bool nym = gst == GATE_ST_NYM;
    const bool cand = (gst == FABGPU_ST_VALID || (nym && memo_issuer_slot(a, t) >= 0)) && t.sig.len >= 1 && t.sig.len <= 1024 &&
                      t.sig.off <= a.arena_len && t.sig.len <= a.arena_len - t.sig.off;
    a.memo_ent[i] = cand ? 1u + (nym ? 32u : 0u) + 64u + 4u + t.sig.len + 4u : 0u;
}
// key lengths -> entry indices and offsets, and whether the keys fit the caller's room.  Round 6: three small launches over tiles of
// 2 048 tuples - tile sums, one workgroup over the tiles, assignment - instead of ONE workgroup over everything: that one took 145 us for
// a 40 000-tuple block (profiles/r06_timeline_10000tx_memo.txt), sat between the gates and the keys' copy, and with the digest memo's
// index behind it the memo's early half no longer fitted beside the verify launches (it ended 0.2 ms after them).
constexpr uint32_t MEMO_TILE = 2048;                                       // 256 threads x 8 tuples
struct MemoTile {
    uint32_t cnt, bytes;          // candidates and key bytes of the tile (<= 2 048 x 1 165 bytes)
    uint32_t cnt_before, pad;
    uint64_t bytes_before;
};
static_assert(sizeof(MemoTile) == 24, "carved as raw bytes");
// a thread's eight lengths, and their (count, bytes)
__device__ __forceinline__ void memo_tile_load(const WalkArrays& a, uint32_t first, uint32_t (&l)[8], uint32_t& k, uint32_t& b) {
    k = b = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        l[j] = first + j < a.n_tuples ? a.memo_ent[first + j] : 0u;
        k += l[j] ? 1u : 0u;
        b += l[j];
    }
}
// exclusive scan of (k, b) over the 256 threads of a workgroup; tk / tb: the workgroup's totals
__device__ __forceinline__ void memo_tile_scan(uint32_t& k, uint32_t& b, uint32_t& tk, uint32_t& tb) {
    __shared__ uint32_t wk[4], wb[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t ik = k, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t vk = __shfl_up(ik, o, 64), vb = __shfl_up(ib, o, 64);
        if (lane >= (uint32_t)o) { ik += vk; ib += vb; }
    }
    if (lane == 63) { wk[wave] = ik; wb[wave] = ib; }
    __syncthreads();
    tk = tb = 0;
    uint32_t bk = 0, bb = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        if (w < wave) { bk += wk[w]; bb += wb[w]; }
        tk += wk[w];
        tb += wb[w];
    }
    k = bk + ik - k;                                                       // exclusive
    b = bb + ib - b;
}
__global__ void __launch_bounds__(256) walk_memo_tile_sums_kernel(WalkArrays a, MemoTile* tiles) {
    uint32_t l[8], k, b, tk, tb;
    memo_tile_load(a, blockIdx.x * MEMO_TILE + threadIdx.x * 8u, l, k, b);
    memo_tile_scan(k, b, tk, tb);
    if (threadIdx.x == 0) {
        tiles[blockIdx.x].cnt = tk;
        tiles[blockIdx.x].bytes = tb;
    }
}
// one workgroup over the tiles (1 024 at a time): what lies before each, the totals, and the verdict "fits"
__global__ void __launch_bounds__(1024) walk_memo_tile_offsets_kernel(WalkArrays a, MemoTile* tiles, uint32_t n_tiles) {
    __shared__ uint32_t wn[16];
    __shared__ uint64_t wb[16];
    __shared__ uint32_t carry_n;
    __shared__ uint64_t carry_b;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) { carry_n = 0; carry_b = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t t = base + tid;
        const uint32_t k = t < n_tiles ? tiles[t].cnt : 0u;
        const uint64_t b = t < n_tiles ? tiles[t].bytes : 0u;
        uint32_t ik = k;
        uint64_t ib = b;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t vk = __shfl_up(ik, o, 64), blo = __shfl_up((uint32_t)ib, o, 64), bhi = __shfl_up((uint32_t)(ib >> 32), o, 64);
            if (lane >= (uint32_t)o) { ik += vk; ib += ((uint64_t)bhi << 32) | blo; }
        }
        if (lane == 63) { wn[wave] = ik; wb[wave] = ib; }
        __syncthreads();
        uint32_t bk = carry_n, tn = 0;
        uint64_t bb = carry_b, tbv = 0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < wave) { bk += wn[w]; bb += wb[w]; }
            tn += wn[w];
            tbv += wb[w];
        }
        if (t < n_tiles) {
            tiles[t].cnt_before = bk + ik - k;
            tiles[t].bytes_before = bb + ib - b;
        }
        __syncthreads();
        if (tid == 0) { carry_n += tn; carry_b += tbv; }
        __syncthreads();
    }
    if (tid == 0) {
        const uint32_t total_n = carry_n;
        const uint64_t total_b = carry_b;
        const bool fits = total_b <= (uint64_t)a.memo_keys_cap;
        a.memo_key_off[fits ? total_n : 0u] = fits ? (uint32_t)total_b : 0u;
        a.memo_totals->n = fits ? total_n : 0u;
        a.memo_totals->overflow = fits ? 0u : 1u;
        a.memo_totals->bytes = fits ? total_b : 0u;
    }
}
// every tuple's entry index and every entry's key offset (or "no entry" for all of them when the keys do not fit)
__global__ void __launch_bounds__(256) walk_memo_tile_assign_kernel(WalkArrays a, const MemoTile* tiles) {
    uint32_t l[8], k, b, tk, tb;
    const uint32_t first = blockIdx.x * MEMO_TILE + threadIdx.x * 8u;
    memo_tile_load(a, first, l, k, b);
    memo_tile_scan(k, b, tk, tb);
    const bool fits = a.memo_totals->overflow == 0u;
    uint32_t e = tiles[blockIdx.x].cnt_before + k;
    uint64_t off = tiles[blockIdx.x].bytes_before + b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (first + j >= a.n_tuples) break;
        if (l[j] && fits) {
            a.memo_key_off[e] = (uint32_t)off;
            a.memo_ent[first + j] = e;
            e++;
            off += l[j];
        } else {
            a.memo_ent[first + j] = 0xFFFFFFFFu;
        }
    }
}
// one wavefront per candidate: its framed key up to the digest
__global__ void __launch_bounds__(256) walk_memo_write_kernel(WalkArrays a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= a.n_tuples) return;
    const uint32_t e = a.memo_ent[i];
    if (e == 0xFFFFFFFFu) return;
    const BlockTuple t = a.tuples[i];
    const uint32_t off = a.memo_key_off[e];
    const bool nym = a.gate_st[i] == GATE_ST_NYM;
    const uint32_t klen = 1u + (nym ? 32u : 0u) + 64u + 4u + t.sig.len + 4u;
    if ((uint64_t)off + klen > (uint64_t)a.memo_keys_cap) return;            // (the scan said it fits: defensive)
    uint8_t* k = a.memo_keys + off;
    const uint8_t* sig = a.block + t.sig.off;
    uint32_t pos = 1;
    if (lane == 0) k[0] = nym ? 2 : 1;
    const uint8_t *kx, *ky;
    if (nym) {
        const uint32_t rank = a.cbase[t.tx];
        const int32_t m_at = memo_issuer_slot(a, t);
        if (lane < 32) k[1 + lane] = a.issuer_hashes[32 * (size_t)(m_at < 0 ? 0 : m_at) + lane];
        pos = 33;
        kx = a.nym_fields + 32 * (size_t)rank;
        ky = a.nym_fields + 32 * (size_t)a.n_creators + 32 * (size_t)rank;
    } else {
        const uint32_t row = a.row_of[i];
        kx = a.qx + 32 * (size_t)row;
        ky = a.qy + 32 * (size_t)row;
    }
    k[pos + lane] = lane < 32 ? kx[lane] : ky[lane & 31u];
    pos += 64;
    if (lane < 4) k[pos + lane] = (uint8_t)(t.sig.len >> (8 * lane));
    pos += 4;
    for (uint32_t b = lane; b < t.sig.len; b += 64) k[pos + b] = sig[b];
    pos += t.sig.len;
    if (lane < 4) k[pos + lane] = lane == 0 ? 32 : 0;
}
// EARLY too, BESIDE the write kernel (its own stream): the DIGEST memo's index (bccsp_host.h BlockMemo; GPUCSP::HashLookup reads it) - one
// lane per candidate: the two spans of its signed message by entry, and a slot in the second table found from walk::msg_fingerprint of
// the message's bytes as they lie in the block (the lines the host runs over a bccsp.Hash caller's bytes; eight unaligned 8-byte loads).
// Every candidate, decided or not: whether an entry's digest may be handed out is its status byte's business (255 = not decided,
// written by the late half), checked by the lookup.
__global__ void __launch_bounds__(256) walk_memo_index_kernel(WalkArrays a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_tuples) return;
    const uint32_t e = a.memo_ent[i];
    if (e == 0xFFFFFFFFu) return;
    const BlockTuple t = a.tuples[i];
    // (spans the walk emitted lie inside the arena; anything else gets spans no lookup can match and no slot)
    const bool inside = t.prefix.off <= a.arena_len && t.prefix.len <= a.arena_len - t.prefix.off && t.suffix.off <= a.arena_len &&
                        t.suffix.len <= a.arena_len - t.suffix.off && (uint64_t)t.prefix.len + t.suffix.len >= bccsp::walk::HASH_MEMO_MIN_LEN;
    uint4 sp;
    sp.x = inside ? t.prefix.off : 0u;
    sp.y = inside ? t.prefix.len : 0u;
    sp.z = inside ? t.suffix.off : 0u;
    sp.w = inside ? t.suffix.len : 0u;
    reinterpret_cast<uint4*>(a.memo_hspans)[e] = sp;
    if (!inside) return;
    const uint64_t h = bccsp::walk::msg_fingerprint(a.block + t.prefix.off, t.prefix.len, a.block + t.suffix.off, t.suffix.len);
    uint32_t at = (uint32_t)h & a.memo_mask;
    for (uint32_t probe = 0; probe <= a.memo_mask; probe++, at = (at + 1) & a.memo_mask)
        if (atomicCAS(&a.memo_hslots[at], 0u, e + 1) == 0u) break;
}
// LATE: one lane per tuple - a candidate that was hashed and decided gets its digest, its status byte and its place in the slot table
// (GPUCSP::MemoHash, linear probing, entry index + 1); the others get status 255 and no slot.
__global__ void __launch_bounds__(256) walk_memo_late_kernel(WalkArrays a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    if (i < a.n_tuples) {
        const uint32_t e = a.memo_ent[i];
        if (e != 0xFFFFFFFFu) {
            const uint8_t st = a.tuple_status[i];
            live = a.tuple_hashed[i] && st <= FABGPU_ST_RANGE;
            a.memo_status[e] = live ? st : (uint8_t)255;
            if (live) {
                const BlockTuple t = a.tuples[i];
                const uint8_t* sig = a.block + t.sig.off;
                const uint4* src = reinterpret_cast<const uint4*>(a.tuple_digests + 32 * (size_t)i);
                uint4* dst = reinterpret_cast<uint4*>(a.memo_digests + 32 * (size_t)e);
                const uint4 d0 = src[0];
                dst[0] = d0;
                dst[1] = src[1];
                const uint64_t ha = ((uint64_t)d0.y << 32) | d0.x;             // the first eight digest bytes, little-endian
                uint64_t hb = 0;
                const uint32_t nb = t.sig.len < 8 ? t.sig.len : 8u, s0 = t.sig.len > 8 ? t.sig.len - 8 : 0u;
                for (uint32_t q = 0; q < nb; q++) hb |= (uint64_t)sig[s0 + q] << (8 * q);
                uint64_t h = (ha ^ (hb * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
                h ^= h >> 32;
                uint32_t at = (uint32_t)h & a.memo_mask;
                for (uint32_t probe = 0; probe <= a.memo_mask; probe++, at = (at + 1) & a.memo_mask)
                    if (atomicCAS(&a.memo_slots[at], 0u, e + 1) == 0u) break;
            }
        }
    }
    const uint64_t lv = __ballot(live);
    if ((threadIdx.x & 63u) == 0 && lv) atomicAdd(&a.memo_totals->live, (uint32_t)__popcll(lv));
}

hipError_t launch_walk_count(const WalkArrays& a, WalkTotals* host_totals, uint32_t* host_flag, uint32_t seq, hipStream_t st) {
    if (a.n_env && a.stash && a.n_env <= WALK_STAGED_MAX) {
        hipLaunchKernelGGL(walk_count_staged_kernel, dim3((a.n_env + 3) / 4), dim3(256), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    } else if (a.n_env) {
        hipLaunchKernelGGL(walk_count_kernel, dim3((a.n_env + 63) / 64), dim3(64), 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(walk_scan_kernel, dim3(1), dim3(1024), 0, st, a.n_env, (const uint4*)a.counts, a.bases, a.cbase, a.totals, host_totals, host_flag, seq);
    return hipGetLastError();
}
hipError_t launch_walk_emit(const WalkArrays& a, const WalkTotals& t, hipStream_t st) {
    const uint32_t n = a.n_env ? a.n_env : 1;                       // (lane 0 closes gather_off even without envelopes)
    hipLaunchKernelGGL(walk_emit_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a, t.checks, (uint32_t)t.gather_bytes);
    return hipGetLastError();
}
hipError_t launch_walk_gate(const WalkArrays& a, hipStream_t st) {
    const uint32_t n = a.gate_mode == 1 ? a.n_env : a.n_tuples;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(walk_gate_kernel, dim3((n + 3) / 4), dim3(256), 0, st, a);   // four wavefronts = four tuples per workgroup
    return hipGetLastError();
}
hipError_t launch_walk_idfix_probe(uint32_t n, const void* arena, const void* spans, void* code, void* key, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(walk_idfix_probe_kernel, dim3(n), dim3(64), 0, st, n, (const uint8_t*)arena, (const uint32_t*)spans, (uint8_t*)code, (uint8_t*)key);
    return hipGetLastError();
}
hipError_t launch_walk_creator_digests(const WalkArrays& a, void* row_digests, hipStream_t st) {
    if (a.n_env == 0) return hipSuccess;
    hipLaunchKernelGGL(walk_creator_digest_kernel, dim3((a.n_env + 255) / 256), dim3(256), 0, st, a, (uint8_t*)row_digests);
    return hipGetLastError();
}
hipError_t launch_walk_gate_probe(uint32_t n, const void* arena, const void* spans, void* code, void* r, void* s, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(walk_gate_probe_kernel, dim3((n + 3) / 4), dim3(256), 0, st, n, (const uint8_t*)arena, (const uint32_t*)spans, (uint8_t*)code, (uint8_t*)r,
                       (uint8_t*)s);
    return hipGetLastError();
}
hipError_t launch_walk_nym_pack(const WalkArrays& a, uint32_t* gather, uint32_t cap, hipStream_t st) {
    if (a.n_creators == 0) return hipSuccess;
    hipLaunchKernelGGL(walk_nym_pack_kernel, dim3(1), dim3(1024), 0, st, a, gather, cap);
    return hipGetLastError();
}
hipError_t launch_walk_memo_early(const WalkArrays& a, hipStream_t st, hipEvent_t scanned) {
    if (a.n_tuples == 0 || !a.memo_ent || !a.memo_tiles) return hipSuccess;
    hipLaunchKernelGGL(walk_memo_len_kernel, dim3((a.n_tuples + 255) / 256), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t n_tiles = (a.n_tuples + MEMO_TILE - 1) / MEMO_TILE;
    MemoTile* tiles = (MemoTile*)a.memo_tiles;
    hipLaunchKernelGGL(walk_memo_tile_sums_kernel, dim3(n_tiles), dim3(256), 0, st, a, tiles);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(walk_memo_tile_offsets_kernel, dim3(1), dim3(1024), 0, st, a, tiles, n_tiles);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(walk_memo_tile_assign_kernel, dim3(n_tiles), dim3(256), 0, st, a, (const MemoTile*)tiles);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (scanned && (e = hipEventRecord(scanned, st)) != hipSuccess) return e;     // entry indices and offsets are assigned: the index kernel may start
    hipLaunchKernelGGL(walk_memo_write_kernel, dim3((a.n_tuples + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError();
}
// the digest memo's index; needs what launch_walk_memo_early's scan assigned (the caller orders it behind that, on a stream of its own
// beside the write kernel; memo_hslots zeroed by the caller)
hipError_t launch_walk_memo_index(const WalkArrays& a, hipStream_t st) {
    if (a.n_tuples == 0 || !a.memo_ent || !a.memo_hspans || !a.memo_hslots) return hipSuccess;
    hipLaunchKernelGGL(walk_memo_index_kernel, dim3((a.n_tuples + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_walk_memo_late(const WalkArrays& a, hipStream_t st) {
    if (a.n_tuples == 0 || !a.memo_ent) return hipSuccess;
    hipLaunchKernelGGL(walk_memo_late_kernel, dim3((a.n_tuples + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_walk_status_checks(const WalkArrays& a, uint32_t n_checks, hipStream_t st) {
    const uint32_t sb = (a.n_tuples + 255) / 256, cb = (n_checks + 255) / 256;
    if (sb + cb == 0) return hipSuccess;
    hipLaunchKernelGGL(walk_status_checks_kernel, dim3(sb + cb), dim3(256), 0, st, a, n_checks, sb);
    return hipGetLastError();
}
bool walk_small_finish_fits(const WalkArrays& a, uint32_t n_checks) {
    (void)n_checks;
    return a.n_tuples != 0 && a.n_tuples <= WALK_SMALL_FINISH_MAX && a.n_env <= WALK_SMALL_FINISH_MAX;
}
hipError_t launch_walk_status_finish_small(const WalkArrays& a, uint32_t n_checks, const WalkHostOut& h, hipStream_t st) {
    const uint32_t sb = (a.n_tuples + 255) / 256, cb = (n_checks + 255) / 256;
    hipLaunchKernelGGL(walk_status_finish_small_kernel, dim3(sb + cb), dim3(256), 0, st, a, n_checks, sb, h);
    return hipGetLastError();
}
hipError_t launch_walk_finish(const WalkArrays& a, const WalkHostOut& h, hipStream_t st) {
    const uint32_t most = a.n_env > a.n_tuples ? a.n_env : a.n_tuples;
    hipLaunchKernelGGL(walk_finish_kernel, dim3(most ? (most + 1023) / 1024 : 1), dim3(256), 0, st, a, h);
    return hipGetLastError();
}

// see warm_kernel_functions_kernels (kernels.hip): the walk's kernels, all of which a provider's FIRST block launches for the first time
int warm_kernel_functions_walk() {
    int ok = 0;
    hipFuncAttributes a;
    const void* fns[] = {(const void*)walk_status_finish_small_kernel, (const void*)walk_status_checks_kernel, (const void*)walk_scan_kernel, (const void*)walk_nym_pack_kernel, (const void*)walk_memo_write_kernel, (const void*)walk_memo_index_kernel, (const void*)walk_memo_tile_sums_kernel, (const void*)walk_memo_tile_offsets_kernel, (const void*)walk_memo_tile_assign_kernel, (const void*)walk_memo_len_kernel, (const void*)walk_memo_late_kernel, (const void*)walk_idfix_probe_kernel, (const void*)walk_gate_probe_kernel, (const void*)walk_gate_kernel, (const void*)walk_finish_kernel, (const void*)walk_emit_kernel, (const void*)walk_creator_digest_kernel, (const void*)walk_count_staged_kernel, (const void*)walk_count_kernel};
    for (const void* f : fns) ok += hipFuncGetAttributes(&a, f) == hipSuccess ? 1 : 0;
    return ok;
}

}  // namespace fab
