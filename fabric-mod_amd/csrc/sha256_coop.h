// SHA-256 with EIGHT lanes on a message, for launches that cannot fill the chip (a few thousand messages at most: the creators' payloads
// and the endorsements' messages of a block of a reference network's size - 10 to 500 transactions).
//
// Such a launch is latency-bound: its time is one lane's instruction stream, 1 512 VALU instructions per 64-byte block at 4.2 cycles
// each with one wavefront on a SIMD (device_common.h sha256_compress; counted in the ISA), a 3.4 KB creator payload 170 us.  The chain
// of 64 rounds per block cannot be cut - round t needs round t - 1 - but a third of those instructions are not in it: the message
// schedule (W[16..63]: 480 instructions) depends on the block's own 64 bytes only.  So the eight lanes of a group take eight
// CONSECUTIVE blocks of their message, each lane expands the schedule of one of them and leaves W[t] + K[t] in LDS (phase 1, once per
// eight blocks), and then all eight walk the rounds of those blocks in turn, reading one word per round (phase 2: the same value in
// all eight lanes - the redundancy costs nothing, a wavefront's instruction takes its four cycles whatever its lanes hold).  Per block:
// ~970 instructions of rounds + 1/8 of ~700 of loading and expanding = ~1 060 against 1 512.
//
// The message is  arena[ps, ps + pl) || arena[sb, sb + b)  - two spans, so that an endorsement's  prp || endorser  (msp/identities.go:178
// via core/common/validation/statebased/validator_keylevel.go:246-258) is hashed whole by its own group, without the mid-state kernel
// in front: three endorsements hash their transaction's prp three times, which on idle SIMDs is free, and one launch (and one
// cross-stream wait) leaves the chain.  pl = 0: a plain message.
//
// Everything is per WAVEFRONT (loop bounds, the LDS buffer - 17 536 bytes -, the ordering of its LDS accesses: a wavefront's LDS
// instructions execute in order, so between "these lanes wrote" and "those lanes read" only the compiler has to be held: no workgroup
// barrier, the other wavefronts of a workgroup are elsewhere in their own messages).
#pragma once
#include "device_common.h"

namespace fab {

constexpr int SHAC_LANES = 8;
constexpr int SHAC_PER_WAVE = 64 / SHAC_LANES;                       // messages per wavefront
// LDS of a group: [block of the chunk][t], 64 + 4 words per block; + 4 per group: the eight groups' 128-bit reads of phase 2 (one address per
// group, broadcast to its eight lanes) start at bank offsets 0, 36, 8, 44, 16, 52, 24, 60 of 64 - disjoint four-bank windows.
// (Rounds 2-4 kept it [t][block] and read ONE word per round: 64 dependent-looking LDS round trips per block that a lone wavefront cannot
// hide - SQ_WAIT_ANY was 32 % of the wave's cycles in profiles/r05_idemix_two_phase_first_pmc_sq.txt, 2.96 ns per instruction instead of
// 1.9.  Now a block's 64 words come in as sixteen 128-bit reads issued a whole block ahead of their use.)
constexpr int SHAC_BLOCK_WORDS = 64 + 4;
constexpr int SHAC_GROUP_WORDS = SHAC_BLOCK_WORDS * SHAC_LANES + 4;
constexpr int SHAC_LDS_WORDS = SHAC_PER_WAVE * SHAC_GROUP_WORDS;

__device__ __forceinline__ void shac_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// h: the digest's eight words in every lane of the group.  lane = the lane in its wavefront, lds = that wavefront's SHAC_LDS_WORDS.
// first32 / first_words: the buffer the FIRST span's offsets (ps) refer to - the arena itself for a block's messages (sha256_coop below), a
// buffer of headers for the idemix challenge (idemix_kernels.hip: "sign" || t || Nym || ipk.Hash, written by the point-arithmetic phase).
__device__ __forceinline__ void sha256_coop_ex(const uint32_t* __restrict__ first32, uint32_t first_words, const uint32_t* __restrict__ arena32,
                                               uint32_t arena_words, uint32_t ps, uint32_t pl, uint32_t sb, uint32_t b, bool active,
                                               uint32_t* __restrict__ lds, uint32_t lane, uint32_t h[8]) {
    const uint32_t sub = lane & (SHAC_LANES - 1);
    uint32_t* __restrict__ wk = lds + (lane >> 3) * SHAC_GROUP_WORDS;
    const uint32_t len = pl + b;
    const uint32_t nblk = active ? ((len + 9 + 63) >> 6) : 0;
    uint32_t maxblk = nblk;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        uint32_t other = __shfl_xor(maxblk, o, 64);
        maxblk = other > maxblk ? other : maxblk;
    }
    maxblk = __builtin_amdgcn_readfirstlane(maxblk);
    const int32_t last_word = arena_words ? (int32_t)arena_words - 1 : 0;
    const int32_t last_first = first_words ? (int32_t)first_words - 1 : 0;
    auto fetch = [&](bool from_first, int32_t wi, uint32_t (&dst)[17]) {
        const uint32_t* __restrict__ base = from_first ? first32 : arena32;
        const int32_t last = from_first ? last_first : last_word;
#pragma unroll
        for (int k = 0; k < 17; k++) {
            int32_t idx = wi + k;
            idx = idx < last ? idx : last;
            idx = idx > 0 ? idx : 0;
            dst[k] = base[idx];
        }
    };
    // the 17 words under this lane's block of chunk c0 (and which span they come from)
    auto chunk_fetch = [&](uint32_t c0, uint32_t (&raw)[17]) {
        const uint32_t pos = (c0 + sub) << 6;
        const bool in_first = pos + 64 <= pl;
        const int32_t vstart = in_first ? (int32_t)ps : (int32_t)sb - (int32_t)pl;
        fetch(in_first, (vstart + (int32_t)pos) >> 2, raw);
    };
    auto rounds = [&](const uint4 (&wk4)[16], uint32_t m) {
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int t = 0; t < 64; t++) {
            const uint4& q = wk4[t >> 2];
            const uint32_t wkt = (t & 3) == 0 ? q.x : ((t & 3) == 1 ? q.y : ((t & 3) == 2 ? q.z : q.w));
            const uint32_t t1 = hh + xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25)) + sha_ch(e, f, g) + wkt;
            const uint32_t t2 = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22)) + sha_maj(a, bb, c);
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a & m; h[1] += bb & m; h[2] += c & m; h[3] += d & m; h[4] += e & m; h[5] += f & m; h[6] += g & m; h[7] += hh & m;
    };
    auto load_block = [&](uint32_t j, uint4 (&wk4)[16]) {
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(wk + j * SHAC_BLOCK_WORDS);
#pragma unroll
        for (int q = 0; q < 16; q++) wk4[q] = src[q];
    };
    sha256_iv(h);
    uint32_t raw_next[17];
    chunk_fetch(0, raw_next);
    for (uint32_t c0 = 0; c0 < maxblk; c0 += SHAC_LANES) {
        {
            // ---- phase 1: this lane's block of the chunk -> W[t] + K[t], t = 0 .. 63 ----
            const uint32_t pos = (c0 + sub) << 6;                          // the block's byte position in the message
            const bool in_first = pos + 64 <= pl;                          // wholly inside the first span
            const int32_t vstart = in_first ? (int32_t)ps : (int32_t)sb - (int32_t)pl;   // arena address of message byte 0 under the block's span
            const uint32_t shift = (uint32_t)vstart & 3u;
            uint32_t w[16], raw[17];
#pragma unroll
            for (int k = 0; k < 17; k++) raw[k] = raw_next[k];             // fetched while the previous chunk's rounds ran
#pragma unroll
            for (int k = 0; k < 16; k++) w[k] = __builtin_bswap32(__builtin_amdgcn_alignbyte(raw[k + 1], raw[k], shift));
            const bool straddles = !in_first && pos < pl;                  // the first span ends inside this block
            if (__ballot(straddles) != 0ull) {
                fetch(true, (int32_t)((ps + pos) >> 2), raw);
                const uint32_t shift_a = ps & 3u;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t va = __builtin_bswap32(__builtin_amdgcn_alignbyte(raw[k + 1], raw[k], shift_a));
                    const int32_t na = straddles ? (int32_t)pl - (int32_t)(pos + 4 * k) : 0;    // bytes of this word that are the first span's
                    const uint32_t keep_a = na >= 4 ? 0xFFFFFFFFu : (na <= 0 ? 0u : ~(0xFFFFFFFFu >> (8 * na)));
                    w[k] = (va & keep_a) | (w[k] & ~keep_a);
                }
            }
            if (__ballot(pos + 64 > len) != 0ull) {                        // a last or padding block somewhere in the wavefront
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int32_t rem = (int32_t)len - (int32_t)(pos + 4 * k);               // message bytes left at this word
                    const uint32_t keep = rem >= 4 ? 0xFFFFFFFFu : (rem <= 0 ? 0u : ~(0xFFFFFFFFu >> (8 * rem)));
                    uint32_t v = w[k] & keep;
                    if (rem >= 0 && rem < 4) v |= 0x80u << (24 - 8 * rem);
                    w[k] = v;
                }
            }
            if (c0 + sub + 1 == nblk) {                                    // the final block carries the bit length
                w[14] = len >> 29;
                w[15] = len << 3;
            }
            uint4* __restrict__ dst = reinterpret_cast<uint4*>(wk + sub * SHAC_BLOCK_WORDS);
#pragma unroll
            for (int t = 0; t < 64; t++) {
                if (t >= 16) {
                    const uint32_t w15 = w[(t - 15) & 15], w2 = w[(t - 2) & 15];
                    const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
                    const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
                    w[t & 15] = w[t & 15] + s0 + w[(t - 7) & 15] + s1;
                }
                if ((t & 3) == 3)
                    dst[t >> 2] = make_uint4(w[(t - 3) & 15] + K256[t - 3], w[(t - 2) & 15] + K256[t - 2], w[(t - 1) & 15] + K256[t - 1], w[t & 15] + K256[t]);
            }
        }
        shac_wave_sync();
        // the next chunk's bytes travel while this chunk's rounds run
        if (c0 + SHAC_LANES < maxblk) chunk_fetch(c0 + SHAC_LANES, raw_next);
        // ---- phase 2: the rounds of the chunk's blocks, one after the other, in all eight lanes alike; block j + 1's words are
        //      requested before block j's rounds start (two register sets, the loop unrolled by two) ----
        const uint32_t cnt = maxblk - c0 < (uint32_t)SHAC_LANES ? maxblk - c0 : (uint32_t)SHAC_LANES;
        uint4 wa[16], wb[16];
        load_block(0, wa);
        for (uint32_t j = 0; j < cnt; j += 2) {
            if (j + 1 < cnt) load_block(j + 1, wb);
            rounds(wa, c0 + j < nblk ? 0xFFFFFFFFu : 0u);                  // (a shorter message of the wavefront is through already)
            if (j + 1 < cnt) {
                if (j + 2 < cnt) load_block(j + 2, wa);
                rounds(wb, c0 + j + 1 < nblk ? 0xFFFFFFFFu : 0u);
            }
        }
        shac_wave_sync();
    }
}

__device__ __forceinline__ void sha256_coop(const uint32_t* __restrict__ arena32, uint32_t arena_words, uint32_t ps, uint32_t pl, uint32_t sb, uint32_t b,
                                            bool active, uint32_t* __restrict__ lds, uint32_t lane, uint32_t h[8]) {
    sha256_coop_ex(arena32, arena_words, arena32, arena_words, ps, pl, sb, b, active, lds, lane, h);
}

__device__ __forceinline__ void sha256_coop_store(uint32_t* __restrict__ digests, uint32_t i, const uint32_t h[8]) {
    uint4* out = reinterpret_cast<uint4*>(digests + 8 * (size_t)i);
    out[0] = make_uint4(__builtin_bswap32(h[0]), __builtin_bswap32(h[1]), __builtin_bswap32(h[2]), __builtin_bswap32(h[3]));
    out[1] = make_uint4(__builtin_bswap32(h[4]), __builtin_bswap32(h[5]), __builtin_bswap32(h[6]), __builtin_bswap32(h[7]));
}

}  // namespace fab
