// Device-side building blocks shared by the kernel translation units (kernels.hip: SHA-256 / ECDSA P-256; idemix_kernels.hip:
// idemix pseudonym signatures): SHA-256 on one message per lane, SoA field loads, verdict packing, the per-lane point-table
// workspace.  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ec29.h"

namespace fab {

// ------------------------------------------------------------------------------------------------
// SHA-256 (FIPS 180-4), one message per lane, ragged arena
// ------------------------------------------------------------------------------------------------
static __device__ __constant__ uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
// a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96): left alone, hipcc emits two v_xor_b32 for every Sigma / sigma
// (476 of the ~1700 instructions of a compressed block)
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
// Ch and Maj likewise (table bit (x << 2 | y << 1 | z) = f(x, y, z)): hipcc finds the one instruction for Ch by itself but spends
// v_xor + v_and + v_bitop3 on Maj - 128 of a block's 1 512 instructions
__device__ __forceinline__ uint32_t sha_ch(uint32_t e, uint32_t f, uint32_t g) { return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
__device__ __forceinline__ uint32_t sha_maj(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }

__device__ __forceinline__ void sha256_compress(uint32_t h[8], uint32_t w[16]) {
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
            uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        uint32_t t1 = hh + xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25)) + sha_ch(e, f, g) + K256[i] + w[i & 15];
        uint32_t t2 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22)) + sha_maj(a, b, c);
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

// SHA-256 of a lane's byte stream  A || B  continuing from state h (the IV, or the mid-state of a shared prefix):
//   A = arena[sa, sa + a), a < 64   (the tail of a shared prefix that did not fill a block; a = 0 without prefix)
//   B = arena[sb, sb + b)           (the lane's own bytes)
// base = bytes already absorbed into h (a multiple of 64); the length field is base + a + b.
// arena_words = readable dwords of the arena allocation: every load index is clamped to [0, arena_words), so no out-of-bounds
// read whatever the offsets say.  The block loop bound is wave-uniform (max over the wave); lanes past their count idle.
// Where the partial first block A comes from: the arena (a shared prefix's tail) or registers (a header the lane built itself).
struct ShaTailArena {
    const uint32_t* __restrict__ arena32;
    int32_t last_word;
    uint32_t sa;
    __device__ __forceinline__ void words(uint32_t va[16]) const {
        const uint32_t shiftA = sa & 3u;
        const int32_t wa = (int32_t)(sa >> 2);
        uint32_t rawA[17];
#pragma unroll
        for (int k = 0; k < 17; k++) {
            int32_t idx = wa + k;
            idx = idx < last_word ? idx : last_word;
            rawA[k] = arena32[idx];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = __builtin_bswap32(__builtin_amdgcn_alignbyte(rawA[k + 1], rawA[k], shiftA));
    }
};
struct ShaTailRegs {
    const uint32_t (&w)[16];   // big-endian words of A, zero beyond its a bytes
    __device__ __forceinline__ void words(uint32_t va[16]) const {
#pragma unroll
        for (int k = 0; k < 16; k++) va[k] = w[k];
    }
};
// PREFETCH: the 17 dwords of block k + 1 are requested before block k is compressed (17 more VGPRs).  A lane reads its own message,
// so one load instruction of a wave touches 64 different cache lines and a block's loads take 2-3 us to come back - as long as the
// ~2 600 instructions of the compression itself when nothing hides them.  The pure hash kernels turn it on (a 2.6 KB message per
// lane: a few per cent on the 2.6 KB creator payloads of a block, 15 % on the short mid-state and hash-check messages); the fused verify kernels are at their register budget and keep the plain loop.
template <class Tail, bool PREFETCH = false>
__device__ __forceinline__ void sha256_stream_t(const uint32_t* __restrict__ arena32, uint32_t arena_words, uint32_t h[8], const Tail& tail,
                                                uint32_t a, uint32_t sb, uint32_t b, uint32_t base, bool active, bool any_prefix) {
    const uint32_t len = a + b;                        // stream bytes still to absorb
    uint32_t nblk = active ? ((len + 9 + 63) >> 6) : 0;
    uint32_t maxblk = nblk;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t other = __shfl_xor(maxblk, o, 64);
        maxblk = other > maxblk ? other : maxblk;
    }
    maxblk = __builtin_amdgcn_readfirstlane(maxblk);
    const int32_t vstart = (int32_t)sb - (int32_t)a;   // B's bytes sit at stream position a: virtual start of the B stream
    const uint32_t shift = (uint32_t)vstart & 3u;      // byte misalignment of the B stream
    const int32_t last_word = arena_words ? (int32_t)arena_words - 1 : 0;
    uint32_t nxt[17];
    auto fetch = [&](uint32_t blk_, uint32_t (&dst)[17]) {
        const int32_t wi_ = (vstart + (int32_t)(blk_ << 6)) >> 2;     // first aligned dword (may be negative in block 0 of a prefixed lane)
#pragma unroll
        for (int k = 0; k < 17; k++) {
            int32_t idx = wi_ + k;
            idx = idx < last_word ? idx : last_word;
            idx = idx > 0 ? idx : 0;
            dst[k] = arena32[idx];
        }
    };
    if (PREFETCH && maxblk) fetch(0, nxt);
    for (uint32_t blk = 0; blk < maxblk; blk++) {
        uint32_t w[16];
        uint32_t pos = blk << 6;                       // byte position of this block inside the stream
        uint32_t raw[17];
        if (PREFETCH) {
#pragma unroll
            for (int k = 0; k < 17; k++) raw[k] = nxt[k];
            if (blk + 1 < maxblk) fetch(blk + 1, nxt);
        } else {
            fetch(blk, raw);
        }
        bool full = pos + 64 <= len;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            // little-endian funnel shift to the stream's byte phase, then to big-endian
            uint32_t v = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], shift);
            w[k] = __builtin_bswap32(v);
        }
        if (any_prefix && blk == 0) {                  // wave-uniform: merge the prefix tail A into the first block
            uint32_t va[16];
            tail.words(va);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                int32_t na = (int32_t)a - 4 * k;        // bytes of this word that belong to A
                uint32_t keepA = na >= 4 ? 0xFFFFFFFFu : (na <= 0 ? 0u : ~(0xFFFFFFFFu >> (8 * na)));
                w[k] = (va[k] & keepA) | (w[k] & ~keepA);
            }
        }
        if (!full) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                uint32_t p = pos + 4 * k;              // byte position of this word
                int32_t rem = (int32_t)len - (int32_t)p;  // stream bytes left at this word
                uint32_t keep = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ~(0xFFFFFFFFu >> (8 * rem)));
                uint32_t v = w[k] & keep;
                if (rem >= 0 && rem < 4) v |= 0x80u << (24 - 8 * rem);
                w[k] = v;
            }
        }
        if (blk + 1 == nblk) {                          // the lane's final block carries the bit length of the whole message
            uint32_t total = base + len;
            w[14] = total >> 29;
            w[15] = total << 3;
        }
        if (blk < nblk) sha256_compress(h, w);
    }
}

template <bool PREFETCH = false>
__device__ __forceinline__ void sha256_stream(const uint32_t* __restrict__ arena32, uint32_t arena_words, uint32_t h[8], uint32_t sa, uint32_t a,
                                              uint32_t sb, uint32_t b, uint32_t base, bool active, bool any_prefix) {
    ShaTailArena tail{arena32, arena_words ? (int32_t)arena_words - 1 : 0, sa};
    sha256_stream_t<ShaTailArena, PREFETCH>(arena32, arena_words, h, tail, a, sb, b, base, active, any_prefix);
}

__device__ __forceinline__ void sha256_iv(uint32_t h[8]) {
    h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
    h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
}
// Hash message [start, start+len) of the arena.
template <bool PREFETCH = false>
__device__ __forceinline__ void sha256_lane(const uint32_t* __restrict__ arena32, uint32_t arena_words, uint32_t start,
                                            uint32_t len, bool active, uint32_t h[8]) {
    sha256_iv(h);
    sha256_stream<PREFETCH>(arena32, arena_words, h, 0, 0, start, len, 0, active, false);
}

// Shared prefixes (the endorsements of one transaction all sign  prp || endorser_i,
// core/common/validation/statebased/validator_keylevel.go:246-258): prefix p = arena[pre_off[p], pre_off[p+1]).
// The mid-state kernel absorbs the whole 64-byte blocks of every prefix once; a message that names prefix p continues from
// mid[p] with the prefix's last (len mod 64) bytes followed by its own suffix.
struct sha_prefixes {
    const uint32_t* pre_idx;   // per message: prefix index, or 0xFFFFFFFF for none; nullptr = the batch has no prefixes
    const uint32_t* pre_off;   // m + 1 offsets into the arena (spans: m (start, end) pairs)
    const uint32_t* mid;       // m x 8 words, written by sha256_midstate_kernel
    uint32_t m;
    uint32_t spans;            // 1: off / pre_off hold (start, end) pairs instead of n + 1 consecutive offsets
    uint32_t* digests;         // optional out: n x 32 bytes, the SHA-256 of every message as the fused kernel computed it (nullptr = none)
};
// The digest a fused kernel verified against leaves the chip only when the caller asks for it (the block pass keys its verdict
// memo on it: bccsp.Verify(k, sig, digest) is what the Go validators will ask, msp/identities.go:188).
__device__ __forceinline__ void emit_digest(const sha_prefixes& pre, uint32_t i, bool active, const uint32_t h[8]) {
    if (pre.digests == nullptr || !active) return;
    uint4* out = reinterpret_cast<uint4*>(pre.digests + 8 * (size_t)i);
    out[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
    out[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
}
// The digest of message i of a (possibly prefixed) batch, in h.
__device__ __forceinline__ void sha256_message(const uint32_t* __restrict__ arena32, uint32_t arena_words, const uint32_t* __restrict__ off,
                                               const sha_prefixes& pre, uint32_t ic, bool active, uint32_t h[8]) {
    uint32_t start = off[pre.spans ? 2 * ic : ic], len = off[pre.spans ? 2 * ic + 1 : ic + 1] - start;
    if (pre.pre_idx == nullptr) {                       // wave-uniform
        sha256_lane(arena32, arena_words, start, len, active, h);
        return;
    }
    uint32_t pi = pre.pre_idx[ic];
    bool has = pi < pre.m;
    uint32_t ps = has ? pre.pre_off[pre.spans ? 2 * pi : pi] : 0, pl = has ? pre.pre_off[pre.spans ? 2 * pi + 1 : pi + 1] - ps : 0;
    uint32_t base = pl & ~63u, tail = pl & 63u;
    sha256_iv(h);
    if (has && base) {
        const uint4* mp = reinterpret_cast<const uint4*>(pre.mid + 8 * (size_t)pi);
        uint4 m0 = mp[0], m1 = mp[1];
        h[0] = m0.x; h[1] = m0.y; h[2] = m0.z; h[3] = m0.w;
        h[4] = m1.x; h[5] = m1.y; h[6] = m1.z; h[7] = m1.w;
    }
    sha256_stream(arena32, arena_words, h, ps + base, tail, start, len, base, active, true);
}

// 32-byte big-endian field i of an SoA array -> limbs. Two coalesced 16-byte loads per lane.
__device__ __forceinline__ void load_be_field(u256& v, const uint8_t* __restrict__ base, uint32_t i) {
    const uint4* p = reinterpret_cast<const uint4*>(base + 32 * (size_t)i);
    uint4 hi = p[0], lo = p[1];
    v.w[7] = __builtin_bswap32(hi.x); v.w[6] = __builtin_bswap32(hi.y);
    v.w[5] = __builtin_bswap32(hi.z); v.w[4] = __builtin_bswap32(hi.w);
    v.w[3] = __builtin_bswap32(lo.x); v.w[2] = __builtin_bswap32(lo.y);
    v.w[1] = __builtin_bswap32(lo.z); v.w[0] = __builtin_bswap32(lo.w);
}

__device__ __forceinline__ void emit_verdict(uint32_t i, bool active, uint32_t st, uint64_t* __restrict__ verdict_bits,
                                             uint8_t* __restrict__ status) {
    uint64_t ballot = __ballot(active && st == 0u);   // 0 = valid in every status vocabulary (ST_VALID, NYM_VALID)
    if ((threadIdx.x & 63) == 0 && active) verdict_bits[i >> 6] = ballot;   // i is a multiple of 64 here
    if (status != nullptr && active) status[i] = (uint8_t)st;
}

// Per-lane table j*Q in a global-memory workspace (not private/scratch memory: the runtime caps a dispatch's scratch
// at ~140 MiB, which at 1.8 KB per lane admitted only ~1270 wavefronts and held the kernel at one wave per SIMD).
// Layout (FABGPU_QTAB_LANE_MAJOR, the default): per LANE contiguous - entry j of lane t is the 128-byte line
// slot + (t * 16 + j - 1) * 128 (seven 16-byte cells = 27 limbs + pad, the eighth cell unused): a gather by digit is ONE cache line
// per lane, fully used.  The round-1 layout [entry][cell][lane] kept a wave's stores contiguous but scattered every gather over seven
// 4 KiB rows in which a lane wanted 16 bytes each (see p256_pair29.h PairQTab for the measured effect on the pair kernel: FETCH_SIZE / 3.8).
#ifndef FABGPU_QTAB_LANE_MAJOR
#define FABGPU_QTAB_LANE_MAJOR 1
#endif
template <int BLOCK>
struct GlobalQTab29 {
    uint4* lane;   // first 16-byte cell of this lane's table
#if FABGPU_QTAB_LANE_MAJOR
    static constexpr size_t STRIDE = 1;          // cell q of an entry at +q
    static constexpr size_t ENTRY = 8;           // cells per entry
    static __device__ __forceinline__ GlobalQTab29 of(uint4* slot, uint32_t t) { return GlobalQTab29{slot + (size_t)t * (16 * 8)}; }
#else
    static constexpr size_t STRIDE = BLOCK;
    static constexpr size_t ENTRY = 7 * (size_t)BLOCK;
    static __device__ __forceinline__ GlobalQTab29 of(uint4* slot, uint32_t t) { return GlobalQTab29{slot + t}; }
#endif
    template <class J>
    __device__ __forceinline__ void store(int j, const J& p) {
        uint4* e = lane + (size_t)(j - 1) * ENTRY;
        e[0 * STRIDE] = make_uint4(p.X.v[0], p.X.v[1], p.X.v[2], p.X.v[3]);
        e[1 * STRIDE] = make_uint4(p.X.v[4], p.X.v[5], p.X.v[6], p.X.v[7]);
        e[2 * STRIDE] = make_uint4(p.X.v[8], p.Y.v[0], p.Y.v[1], p.Y.v[2]);
        e[3 * STRIDE] = make_uint4(p.Y.v[3], p.Y.v[4], p.Y.v[5], p.Y.v[6]);
        e[4 * STRIDE] = make_uint4(p.Y.v[7], p.Y.v[8], p.Z.v[0], p.Z.v[1]);
        e[5 * STRIDE] = make_uint4(p.Z.v[2], p.Z.v[3], p.Z.v[4], p.Z.v[5]);
        e[6 * STRIDE] = make_uint4(p.Z.v[6], p.Z.v[7], p.Z.v[8], 0);
    }
    template <class J>
    __device__ __forceinline__ void load(uint32_t d, J& p) const {
        const uint4* e = lane + (size_t)(d - 1) * ENTRY;
        uint4 a = e[0 * STRIDE], b = e[1 * STRIDE], c = e[2 * STRIDE], dd = e[3 * STRIDE];
        uint4 f = e[4 * STRIDE], g = e[5 * STRIDE], h = e[6 * STRIDE];
        p.X.v[0] = a.x; p.X.v[1] = a.y; p.X.v[2] = a.z; p.X.v[3] = a.w;
        p.X.v[4] = b.x; p.X.v[5] = b.y; p.X.v[6] = b.z; p.X.v[7] = b.w;
        p.X.v[8] = c.x; p.Y.v[0] = c.y; p.Y.v[1] = c.z; p.Y.v[2] = c.w;
        p.Y.v[3] = dd.x; p.Y.v[4] = dd.y; p.Y.v[5] = dd.z; p.Y.v[6] = dd.w;
        p.Y.v[7] = f.x; p.Y.v[8] = f.y; p.Z.v[0] = f.z; p.Z.v[1] = f.w;
        p.Z.v[2] = g.x; p.Z.v[3] = g.y; p.Z.v[4] = g.z; p.Z.v[5] = g.w;
        p.Z.v[6] = h.x; p.Z.v[7] = h.y; p.Z.v[8] = h.z;
    }
};

// 32 verdicts per wave sit on the even lanes: squeeze the even bits of the ballot into one u32 half-word
__device__ __forceinline__ void pair_emit_verdict(uint32_t i, uint32_t n, bool active, bool odd, uint32_t st, uint32_t* __restrict__ verdict32,
                                                  uint8_t* __restrict__ status) {
    uint64_t x = __ballot(active && !odd && st == 0u) & 0x5555555555555555ull;   // 0 = valid (ST_VALID, NYM_VALID)
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
    x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
    x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
    x = (x | (x >> 16)) & 0x00000000ffffffffull;
    if ((threadIdx.x & 63) == 0 && active) {       // i is a multiple of 32 here
        verdict32[i >> 5] = (uint32_t)x;
        if (i + 32 >= n && ((i >> 5) & 1u) == 0) verdict32[(i >> 5) + 1] = 0;   // no wave owns the upper half of the last word
    }
    if (status != nullptr && active && !odd) status[i] = (uint8_t)st;
}

// partner lane of a two-lanes-per-signature kernel (quad_perm:[1,0,3,2])
__device__ __forceinline__ int32_t lane_pair_swap(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false); }

}  // namespace fab
