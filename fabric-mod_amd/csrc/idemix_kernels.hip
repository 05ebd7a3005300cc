// HIP kernel for gfx950 (MI355X / CDNA4): batched idemix pseudonym-signature verification on FP256BN, one signature per
// lane, the two SHA-256 of the Fiat-Shamir challenge fused behind the point arithmetic (the commitment t never leaves the
// registers).  Integer VALU work, no MFMA.
//
// Replaces (reference, all CPU): idemix/nymsignature.go:74-109, called per creator signature from msp/idemixmsp.go:584-599
// through bccsp/idemix/handlers/nymsigner.go:62-95.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bn_nym29.h"
#include "bn_quad29.h"
#include "device_common.h"
#include "kernels.h"
#include "sha256_coop.h"

namespace fab {

// 32-byte big-endian field (a u256's words, most significant first) into the header words at byte offset OFF
template <int OFF>
__device__ __forceinline__ void put_be32(uint32_t* hw, const u256& v) {
    constexpr int W = OFF / 4, S = OFF % 4;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint32_t be = v.w[7 - j];
        if (S == 0) {
            hw[W + j] = be;
        } else {
            hw[W + j] |= be >> (8 * S);
            hw[W + j + 1] |= be << (32 - 8 * S);
        }
    }
}

// idemix/nymsignature.go:89-107 for one lane.  proofData = "sign" || 04 t.x t.y || 04 nym.x nym.y || ipk.Hash || msg :
// a 166-byte header the lane assembles in registers (two whole SHA blocks + 38 bytes that share the third block with the
// start of the message), then the message from the arena; c = digest mod r; ProofC' = SHA-256(c || nonce) mod r.
// The first 166 bytes of proofData (idemix/nymsignature.go:89-99): "sign" || 04 t.x t.y || 04 nym.x nym.y || ipk.Hash, as 48 big-endian words
// (the last 26 bytes zero).
constexpr int NYM_HDR_BYTES = 166;
constexpr int NYM_HDR_WORDS = 48;        // 192 bytes per row in the header buffer of the two-phase form
__device__ __forceinline__ void nym_header_words(uint32_t hw[NYM_HDR_WORDS], const u256& tx, const u256& ty, const u256& nx, const u256& ny,
                                                 const uint32_t ipk_hash[8]) {
#pragma unroll
    for (int k = 0; k < 48; k++) hw[k] = 0;
    hw[0] = 0x7369676eu;                 // "sign" (idemix/signature.go:19)
    hw[1] = 0x04u << 24;                 // byte 4: uncompressed-point tag of t
    put_be32<5>(hw, tx);
    put_be32<37>(hw, ty);
    hw[17] |= 0x04u << 16;               // byte 69: tag of nym
    put_be32<70>(hw, nx);
    put_be32<102>(hw, ny);
#pragma unroll
    for (int j = 0; j < 8; j++) {        // ipk.Hash at byte 134
        hw[33 + j] |= ipk_hash[j] >> 16;
        hw[34 + j] |= ipk_hash[j] << 16;
    }
}

// idemix/nymsignature.go:100-107 behind the first hash: c = digest mod r; ProofC' = SHA-256(c || nonce) mod r; equal to ProofC?
__device__ __forceinline__ bool nym_second_hash_matches(const uint32_t h1[8], const u256& nonce, const u256& proof_c) {
    uint32_t h[8], w[16];
    u256 d, c;
#pragma unroll
    for (int k = 0; k < 8; k++) d.w[k] = h1[7 - k];
    bn_mod_order(c, d);
    // c (32 bytes) || nonce (32 bytes) = one block, then the padding block of a 64-byte message
    sha256_iv(h);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        w[j] = c.w[7 - j];
        w[8 + j] = nonce.w[7 - j];
    }
    sha256_compress(h, w);
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = 0;
    w[0] = 0x80000000u;
    w[15] = 512u;
    sha256_compress(h, w);
#pragma unroll
    for (int k = 0; k < 8; k++) d.w[k] = h[7 - k];
    bn_mod_order(c, d);
    uint32_t diff = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) diff |= c.w[k] ^ proof_c.w[k];
    return diff == 0;
}

__device__ __forceinline__ bool nym_challenge_matches(const uint32_t* __restrict__ arena32, uint32_t arena_words, uint32_t start, uint32_t len,
                                                      bool active, const u256& tx, const u256& ty, const u256& nx, const u256& ny,
                                                      const uint32_t ipk_hash[8], const u256& nonce, const u256& proof_c) {
    uint32_t hw[NYM_HDR_WORDS];
    nym_header_words(hw, tx, ty, nx, ny, ipk_hash);
    uint32_t h[8], w[16];
    sha256_iv(h);
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = hw[k];
    sha256_compress(h, w);
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = hw[16 + k];
    sha256_compress(h, w);
    uint32_t tailw[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tailw[k] = hw[32 + k];
    ShaTailRegs tail{tailw};
    sha256_stream_t(arena32, arena_words, h, tail, 38u, start, len, 128u, active, true);

    return nym_second_hash_matches(h, nonce, proof_c);
}

// One registered issuer on the device: comb tables of HSk and HRand, ipk.Hash as big-endian words.
struct IssuerDev {
    const int32_t* hsk;
    const int32_t* hrand;
    uint32_t hash[8];
};

// Register budget: bounded at two workgroups per CU (256 registers per wave) although the launcher asks for one per CU.  Measured
// (r01, MI355X): with the whole file (bound 1, no spills) the kernel alone is 2 % faster, but a wave then owns its SIMD and the
// ECDSA kernel of a mixed batch (BASELINE config 5, the other stream) can no longer share the CU: 300 000 mixed signatures went
// from 6.2 ms to 7.4 ms.  The spills (258 VGPRs, 13 k scratch accesses per wave against 850 k VALU instructions) are the cheaper evil.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2)
    idemix_nym_verify_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words, const uint32_t* __restrict__ off,
                             uint32_t spans, const uint32_t* __restrict__ issuer_id, const IssuerDev* __restrict__ issuers, uint32_t n_issuers,
                             const uint8_t* __restrict__ nym_x, const uint8_t* __restrict__ nym_y, const uint8_t* __restrict__ proof_c,
                             const uint8_t* __restrict__ s_sk, const uint8_t* __restrict__ s_rnym, const uint8_t* __restrict__ nonce,
                             uint4* __restrict__ qws, uint64_t* __restrict__ verdict_bits, uint8_t* __restrict__ status,
                             const uint32_t* __restrict__ gather) {
    GlobalQTab29<BLOCK> qtab = GlobalQTab29<BLOCK>::of(qws + (size_t)blockIdx.x * (QWS_UINT4_PER_LANE * BLOCK), threadIdx.x);
    const uint32_t ntiles = (n + BLOCK - 1) / BLOCK;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * BLOCK + threadIdx.x;
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        if (gather != nullptr) {                                           // row i reads the inputs of row gather[i] (~0: nobody's)
            const uint32_t g = gather[ic];
            active = active && g != 0xFFFFFFFFu;
            ic = g != 0xFFFFFFFFu ? g : 0u;
        }
        uint32_t iss = issuer_id != nullptr ? issuer_id[ic] : 0u;
        bool iss_ok = iss < n_issuers;
        const IssuerDev* id = issuers + (iss_ok ? iss : 0u);
        KeyTab8 hsk{id->hsk}, hrand{id->hrand};
        u256 nx, ny, c, ssk, srn, nn, tx, ty;
        load_be_field(nx, nym_x, ic);
        load_be_field(ny, nym_y, ic);
        load_be_field(c, proof_c, ic);
        load_be_field(ssk, s_sk, ic);
        load_be_field(srn, s_rnym, ic);
        load_be_field(nn, nonce, ic);
        uint32_t st = bn_nym_commitment29(tx, ty, nx, ny, c, ssk, srn, hsk, hrand, qtab);
        uint32_t ih[8];
#pragma unroll
        for (int k = 0; k < 8; k++) ih[k] = id->hash[k];
        uint32_t start = spans ? off[2 * ic] : off[ic], len = (spans ? off[2 * ic + 1] : off[ic + 1]) - start;   // spans: (start, end) pairs
        bool match = nym_challenge_matches(arena32, arena_words, start, len, active, tx, ty, nx, ny, ih, nn, c);
        if (st == NYM_VALID) st = match ? NYM_VALID : NYM_BAD_PROOF;
        if (!iss_ok) st = NYM_NEEDS_SW;
        emit_verdict(i, active, st, verdict_bits, status);
    }
}

// Two lanes per signature (bn_nym29.h "two lanes per signature"): 128 signatures per 256-thread workgroup, for batches that
// cannot fill the chip with one signature per lane - the idemix creators of one block.  Lane 2k / 2k+1 of a wave own signature k;
// each computes one half of the verification equation, the halves are exchanged by DPP, the even lane carries the verdict.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2)
    idemix_nym_verify_split_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words, const uint32_t* __restrict__ off,
                                   uint32_t spans, const uint32_t* __restrict__ issuer_id, const IssuerDev* __restrict__ issuers, uint32_t n_issuers,
                                   const uint8_t* __restrict__ nym_x, const uint8_t* __restrict__ nym_y, const uint8_t* __restrict__ proof_c,
                                   const uint8_t* __restrict__ s_sk, const uint8_t* __restrict__ s_rnym, const uint8_t* __restrict__ nonce,
                                   uint4* __restrict__ qws, uint64_t* __restrict__ verdict_bits, uint8_t* __restrict__ status,
                             const uint32_t* __restrict__ gather) {
    GlobalQTab29<BLOCK> qtab = GlobalQTab29<BLOCK>::of(qws + (size_t)blockIdx.x * (QWS_UINT4_PER_LANE * BLOCK), threadIdx.x);
    constexpr uint32_t PER_WG = BLOCK / 2;
    const uint32_t ntiles = (n + PER_WG - 1) / PER_WG;
    const bool odd = (threadIdx.x & 1u) != 0;
    uint32_t* verdict32 = reinterpret_cast<uint32_t*>(verdict_bits);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * PER_WG + (threadIdx.x >> 1);
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        if (gather != nullptr) {                                           // row i reads the inputs of row gather[i] (~0: nobody's)
            const uint32_t g = gather[ic];
            active = active && g != 0xFFFFFFFFu;
            ic = g != 0xFFFFFFFFu ? g : 0u;
        }
        uint32_t iss = issuer_id != nullptr ? issuer_id[ic] : 0u;
        bool iss_ok = iss < n_issuers;
        const IssuerDev* id = issuers + (iss_ok ? iss : 0u);
        KeyTab8 hsk{id->hsk}, hrand{id->hrand};
        u256 nx, ny, c, ssk, srn, nn, tx, ty;
        load_be_field(nx, nym_x, ic);
        load_be_field(ny, nym_y, ic);
        load_be_field(c, proof_c, ic);
        load_be_field(ssk, s_sk, ic);
        load_be_field(srn, s_rnym, ic);
        load_be_field(nn, nonce, ic);
        bn_nym_half mine;
        bn_nym_split_part1(mine, odd, nx, ny, c, ssk, srn, hsk, hrand, qtab);
        jacbn theirs;
#pragma unroll
        for (int l = 0; l < 9; l++) {
            theirs.X.v[l] = lane_pair_swap(mine.P.X.v[l]);
            theirs.Y.v[l] = lane_pair_swap(mine.P.Y.v[l]);
            theirs.Z.v[l] = lane_pair_swap(mine.P.Z.v[l]);
        }
        bool theirs_inf = lane_pair_swap(mine.inf ? 1 : 0) != 0;
        uint32_t st = bn_nym_split_part2(tx, ty, mine, theirs, theirs_inf);
        uint32_t ih[8];
#pragma unroll
        for (int k = 0; k < 8; k++) ih[k] = id->hash[k];
        uint32_t start = spans ? off[2 * ic] : off[ic], len = (spans ? off[2 * ic + 1] : off[ic + 1]) - start;   // spans: (start, end) pairs
        bool match = nym_challenge_matches(arena32, arena_words, start, len, active, tx, ty, nx, ny, ih, nn, c);
        if (st == NYM_VALID) st = match ? NYM_VALID : NYM_BAD_PROOF;
        if (!iss_ok) st = NYM_NEEDS_SW;
        pair_emit_verdict(i, n, active, odd, st, verdict32, status);
    }
}

// Four lanes per signature (bn_quad29.h): 64 signatures per 256-thread workgroup = one verdict word per tile, for the batches the
// idemix creators of one block make (a few thousand).  Lanes 4k .. 4k+3 of a wave own signature k: pair 0 computes
// HSk s_sk - k1 Nym, pair 1 HRand s_rnym - k2 phi(Nym), every point operation on two lanes; lane 4k hashes and reports.
//
// FUSED = false (the default since round 5, "two phases"): the kernel stops at the commitment - lane 4k writes the 166-byte header
// "sign" || t || Nym || ipk.Hash of its signature (192 bytes per launch row) and its status so far into the workspace - and
// idemix_nym_challenge_coop_kernel hashes behind it with EIGHT lanes on a message (sha256_coop.h).  In the fused form one lane of four
// walks the 73 blocks of a 4.6 KB creator message alone - 1 512 instructions per block, a quarter of the kernel's stream with three
// lanes of four idle - and a wave holds sixteen messages, twice what the cooperative form takes; splitting the phases lets each use
// the lane count its arithmetic wants.
template <int BLOCK, bool FUSED>
__global__ void __launch_bounds__(BLOCK, FUSED ? 2 : 1)
    idemix_nym_verify_quad_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words, const uint32_t* __restrict__ off,
                                  uint32_t spans, const uint32_t* __restrict__ issuer_id, const IssuerDev* __restrict__ issuers, uint32_t n_issuers,
                                  const uint8_t* __restrict__ nym_x, const uint8_t* __restrict__ nym_y, const uint8_t* __restrict__ proof_c,
                                  const uint8_t* __restrict__ s_sk, const uint8_t* __restrict__ s_rnym, const uint8_t* __restrict__ nonce,
                                  uint4* __restrict__ qws, uint64_t* __restrict__ verdict_bits, uint8_t* __restrict__ status,
                             const uint32_t* __restrict__ gather, uint32_t* __restrict__ hdr_out, uint8_t* __restrict__ st_out,
                             uint32_t* __restrict__ comb_flags, const uint4* __restrict__ comb_rec) {
    // one table per PAIR: BLOCK / 2 of them in this workgroup's slot
    PairBNQTab qtab = PairBNQTab::of(qws + (size_t)blockIdx.x * (QWS_UINT4_PER_LANE * (BLOCK / 2)), threadIdx.x >> 1);
    constexpr uint32_t PER_WG = BLOCK / 4;
    const uint32_t ntiles = (n + PER_WG - 1) / PER_WG;
    const uint32_t nwords = (n + 63) / 64;
    const bool odd = (threadIdx.x & 1u) != 0;
    const bool half = (threadIdx.x & 2u) != 0;
    uint16_t* verdict16 = reinterpret_cast<uint16_t*>(verdict_bits);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t i = tile * PER_WG + (threadIdx.x >> 2);
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        if (gather != nullptr) {                                           // row i reads the inputs of row gather[i] (~0: nobody's)
            const uint32_t g = gather[ic];
            active = active && g != 0xFFFFFFFFu;
            ic = g != 0xFFFFFFFFu ? g : 0u;
        }
        uint32_t iss = issuer_id != nullptr ? issuer_id[ic] : 0u;
        bool iss_ok = iss < n_issuers;
        const IssuerDev* id = issuers + (iss_ok ? iss : 0u);
        u256 nx, ny, c, ssk, srn, nn, tx, ty;
        load_be_field(nx, nym_x, ic);
        load_be_field(ny, nym_y, ic);
        load_be_field(c, proof_c, ic);
        load_be_field(ssk, s_sk, ic);
        load_be_field(srn, s_rnym, ic);
        load_be_field(nn, nonce, ic);
        bn_nym_quad_half mine;
        // (comb_flags: the fixed-base terms come from idemix_nym_comb_quad_kernel, launched beside this one - one flag per wavefront of
        //  a tile, one 80-byte record per lane; nullptr: computed here)
        bn_nym_quad_part1(mine, odd, half, nx, ny, c, ssk, srn, id->hsk, id->hrand, qtab,
                          comb_flags != nullptr ? comb_flags + (size_t)tile * (BLOCK / 64) + (threadIdx.x >> 6) : nullptr,
                          comb_rec != nullptr ? comb_rec + ((size_t)tile * BLOCK + threadIdx.x) * NYM_COMB_UINT4_PER_LANE : nullptr);
        uint32_t st = bn_nym_quad_part2(tx, ty, mine, odd);
        uint32_t ih[8];
#pragma unroll
        for (int k = 0; k < 8; k++) ih[k] = id->hash[k];
        const bool lead = (threadIdx.x & 3u) == 0;
        if (!FUSED) {
            // phase 1 ends here: the header and the status so far, by LAUNCH row (rows of the tile's tail and idle rows too: phase 2 reads them)
            if (!iss_ok) st = NYM_NEEDS_SW;
            if (lead) {
                uint32_t hw[NYM_HDR_WORDS];
                nym_header_words(hw, tx, ty, nx, ny, ih);
                uint4* dst = reinterpret_cast<uint4*>(hdr_out + (size_t)i * NYM_HDR_WORDS);
#pragma unroll
                for (int q = 0; q < 11; q++)         // 44 words cover the 166 bytes; stored as the BYTES they are (the hash phase reads an arena)
                    dst[q] = make_uint4(__builtin_bswap32(hw[4 * q]), __builtin_bswap32(hw[4 * q + 1]), __builtin_bswap32(hw[4 * q + 2]), __builtin_bswap32(hw[4 * q + 3]));
                st_out[i] = (uint8_t)(active ? st : 0xFFu);
            }
            continue;
        }
        uint32_t start = spans ? off[2 * ic] : off[ic], len = (spans ? off[2 * ic + 1] : off[ic + 1]) - start;   // spans: (start, end) pairs
        bool match = nym_challenge_matches(arena32, arena_words, start, len, active, tx, ty, nx, ny, ih, nn, c);
        if (st == NYM_VALID) st = match ? NYM_VALID : NYM_BAD_PROOF;
        if (!iss_ok) st = NYM_NEEDS_SW;
        // 16 verdicts per wave sit on lanes 0, 4, 8, ...: squeeze every fourth bit of the ballot into 16 bits
        uint64_t x = __ballot(active && lead && st == 0u) & 0x1111111111111111ull;
        x = (x | (x >> 3)) & 0x0303030303030303ull;
        x = (x | (x >> 6)) & 0x000f000f000f000full;
        x = (x | (x >> 12)) & 0x000000ff000000ffull;
        x = (x | (x >> 24)) & 0x000000000000ffffull;
        const uint32_t i0 = tile * PER_WG + ((threadIdx.x & ~63u) >> 2);       // first signature of this wave: a multiple of 16
        if ((threadIdx.x & 63u) == 0 && (i0 >> 6) < nwords) verdict16[i0 >> 4] = (uint16_t)x;   // every 16-bit part of every word has an owner
        if (status != nullptr && active && lead) status[i] = (uint8_t)st;
    }
}

// The fixed-base half of the commitments on its own (round 5, "three launches"): HSk * s_sk and HRand * s_rnym need nothing of the
// pseudonym, are a quarter of the commitment kernel's stream (32 mixed additions of 135 doublings + 27 additions + a table), and 6 000
// signatures leave 650 of 1 024 SIMDs idle - so this kernel runs BESIDE idemix_nym_verify_quad_kernel<.., false> on a second stream,
// same geometry (tile, lane) -> same rows, and leaves per lane the pair state of its term (80 bytes) and per wavefront a flag, written
// behind the wavefront's records with release semantics.  The commitment kernel picks the records up after its variable-base half if
// the flag is there, and computes the term itself if it is not (a chip busy with other launches): same bits either way.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2)
    idemix_nym_comb_quad_kernel(uint32_t n, const uint32_t* __restrict__ issuer_id, const IssuerDev* __restrict__ issuers, uint32_t n_issuers,
                                const uint8_t* __restrict__ s_sk, const uint8_t* __restrict__ s_rnym, const uint32_t* __restrict__ gather,
                                uint32_t* __restrict__ comb_flags, uint4* __restrict__ comb_rec) {
    constexpr uint32_t PER_WG = BLOCK / 4;
    const uint32_t ntiles = (n + PER_WG - 1) / PER_WG;
    const bool odd = (threadIdx.x & 1u) != 0;
    const bool half = (threadIdx.x & 2u) != 0;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t* flag = comb_flags + (size_t)tile * (BLOCK / 64) + (threadIdx.x >> 6);
        // the commitment kernel has already given up on this wavefront's rows (it computed them itself): nothing to do
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 2u) continue;
        uint32_t i = tile * PER_WG + (threadIdx.x >> 2);
        bool active = i < n;
        uint32_t ic = active ? i : (n - 1);
        if (gather != nullptr) {
            const uint32_t g = gather[ic];
            ic = g != 0xFFFFFFFFu ? g : 0u;
        }
        const uint32_t iss = issuer_id != nullptr ? issuer_id[ic] : 0u;
        const IssuerDev* id = issuers + (iss < n_issuers ? iss : 0u);
        u256 ssk, srn;
        load_be_field(ssk, s_sk, ic);
        load_be_field(srn, s_rnym, ic);
        pairbn_pt S;
        bool s_inf;
        bn_nym_quad_comb(S, s_inf, odd, half, ssk, srn, id->hsk, id->hrand);
        bn_nym_comb_store(comb_rec + ((size_t)tile * BLOCK + threadIdx.x) * NYM_COMB_UINT4_PER_LANE, S, s_inf);
        // the wavefront's 64 records first, then its flag: 0 -> 1 with release semantics; a 2 that arrived meanwhile stays (the records are
        // simply not read).  EVERY lane fences its own record stores at agent scope before lane 0 publishes the flag: lane 0's release
        // orders lane 0's stores only, as far as the memory model is concerned - that the hardware's s_waitcnt covers the whole
        // wavefront is how it happened to work before, not what the code may rely on.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if ((threadIdx.x & 63u) == 0) {
            uint32_t expect = 0u;
            __hip_atomic_compare_exchange_strong(flag, &expect, 1u, __ATOMIC_RELEASE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Phase 2 of the two-phase form: the Fiat-Shamir challenge of every launch row with EIGHT lanes on its message (sha256_coop.h: eight
// consecutive blocks' schedules in parallel, then the rounds; ~1 060 instructions per block against 1 512).  The message of row i is
// header[i] (166 bytes, phase 1 wrote them) || arena[start, start + len); c = digest mod r, ProofC' = SHA-256(c || nonce) mod r on every
// lane of the group alike.  A wavefront holds eight rows = one byte of the verdict bitmap; W wavefronts per workgroup, no grid stride:
// the launch covers ceil(n / 64) * 64 rows, so every byte of every verdict word has an owner.
template <int W>
__global__ void __launch_bounds__(64 * W)
    idemix_nym_challenge_coop_kernel(uint32_t n, const uint32_t* __restrict__ arena32, uint32_t arena_words, const uint32_t* __restrict__ off, uint32_t spans,
                                     const uint8_t* __restrict__ proof_c, const uint8_t* __restrict__ nonce, const uint32_t* __restrict__ hdr,
                                     const uint8_t* __restrict__ st_in, uint64_t* __restrict__ verdict_bits, uint8_t* __restrict__ status,
                                     const uint32_t* __restrict__ gather) {
    extern __shared__ uint32_t shac_lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* lds = shac_lds + wave * SHAC_LDS_WORDS;
    const uint32_t row0 = (blockIdx.x * W + wave) * SHAC_PER_WAVE;             // this wavefront's first row: a multiple of 8
    const uint32_t i = row0 + (lane >> 3);
    bool active = i < n;
    uint32_t ic = active ? i : (n - 1);
    if (gather != nullptr) {
        const uint32_t g = gather[ic];
        active = active && g != 0xFFFFFFFFu;
        ic = g != 0xFFFFFFFFu ? g : 0u;
    }
    const uint32_t start = spans ? off[2 * ic] : off[ic], len = (spans ? off[2 * ic + 1] : off[ic + 1]) - start;
    const uint32_t rows = ((n + 63u) / 64u) * 64u;
    uint32_t h[8];
    sha256_coop_ex(hdr, rows * NYM_HDR_WORDS, arena32, arena_words, (active ? i : 0u) * (NYM_HDR_WORDS * 4), NYM_HDR_BYTES, start, len, active, lds, lane, h);
    u256 nn, pc;
    load_be_field(nn, nonce, ic);
    load_be_field(pc, proof_c, ic);
    const bool match = nym_second_hash_matches(h, nn, pc);
    uint32_t st = active ? st_in[i] : 0xFFu;
    if (st == NYM_VALID) st = match ? NYM_VALID : NYM_BAD_PROOF;
    const bool lead = (lane & 7u) == 0;
    // eight verdicts per wave sit on lanes 0, 8, 16, ...: gather every eighth bit of the ballot into one byte
    const uint64_t x = __ballot(active && lead && st == 0u) & 0x0101010101010101ull;
    const uint8_t byte = (uint8_t)((x * 0x0102040810204080ull) >> 56);
    if (lane == 0 && row0 < rows) reinterpret_cast<uint8_t*>(verdict_bits)[row0 >> 3] = byte;
    if (status != nullptr && active && lead) status[i] = (uint8_t)st;
}

size_t idemix_issuer_dev_bytes() { return sizeof(IssuerDev); }
void idemix_issuer_dev_fill(void* host_slot, const void* d_hsk, const void* d_hrand, const uint8_t hash32[32]) {
    IssuerDev* s = (IssuerDev*)host_slot;
    s->hsk = (const int32_t*)d_hsk;
    s->hrand = (const int32_t*)d_hrand;
    for (int k = 0; k < 8; k++)
        s->hash[k] = ((uint32_t)hash32[4 * k] << 24) | ((uint32_t)hash32[4 * k + 1] << 16) | ((uint32_t)hash32[4 * k + 2] << 8) | hash32[4 * k + 3];
}

static bool idemix_quad(uint32_t n, bool allow_split, bool allow_quad) { return allow_split && allow_quad && n <= (uint32_t)IDEMIX_QUAD_MAX; }
static uint32_t idemix_quad_wgs(uint32_t n) {
    const uint32_t tiles = (n + VERIFY_BLOCK / 4 - 1) / (VERIFY_BLOCK / 4);
    return tiles < (uint32_t)VERIFY_MAX_WGS ? tiles : (uint32_t)VERIFY_MAX_WGS;
}
static size_t idemix_quad_tables_bytes(uint32_t n) { return (size_t)idemix_quad_wgs(n) * (VERIFY_BLOCK / 2) * QWS_UINT4_PER_LANE * 16; }
static size_t idemix_rows(uint32_t n) { return ((size_t)n + 63) / 64 * 64; }
size_t idemix_workspace_bytes(uint32_t n, bool allow_split, bool allow_quad) {
    // four-lane form: the pairs' tables | two-phase form: + 192 bytes of header and one status byte per launch row
    // (+ the comb launch's records, 80 bytes per lane = 320 per row, and one flag per wavefront)
    if (idemix_quad(n, allow_split, allow_quad))
        return idemix_quad_tables_bytes(n) + idemix_rows(n) * (NYM_HDR_WORDS * 4 + 1 + 4 * NYM_COMB_UINT4_PER_LANE * 16) + idemix_rows(n) / 16 * 4 + 1024;
    VerifyGeom g = verify_geom(n, allow_split);
    return (size_t)g.wgs * g.block * QWS_UINT4_PER_LANE * 16;
}

hipError_t launch_idemix_nym_verify(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* issuer_id, const void* issuers,
                                    uint32_t n_issuers, const void* nym_x, const void* nym_y, const void* proof_c, const void* s_sk,
                                    const void* s_rnym, const void* nonce, void* qws, void* verdict_bits, void* status, bool allow_split,
                                    bool allow_quad, bool spans, hipStream_t st, const void* gather, uint32_t lds_reserve, bool two_phase, const NymSide* side) {
    if (n == 0) return hipSuccess;
    if (idemix_quad(n, allow_split, allow_quad)) {                             // four lanes per signature: 64 signatures per workgroup
        dim3 qgrid(idemix_quad_wgs(n)), qblock(VERIFY_BLOCK);
        if (!two_phase) {
            hipLaunchKernelGGL((idemix_nym_verify_quad_kernel<VERIFY_BLOCK, true>), qgrid, qblock, lds_reserve, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                               (const uint32_t*)off, spans ? 1u : 0u, (const uint32_t*)issuer_id, (const IssuerDev*)issuers, n_issuers, (const uint8_t*)nym_x,
                               (const uint8_t*)nym_y, (const uint8_t*)proof_c, (const uint8_t*)s_sk, (const uint8_t*)s_rnym, (const uint8_t*)nonce,
                               (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status, (const uint32_t*)gather, (uint32_t*)nullptr, (uint8_t*)nullptr,
                               (uint32_t*)nullptr, (const uint4*)nullptr);
            return hipGetLastError();
        }
        // three launches: the fixed-base terms on the side stream BESIDE the commitments (four lanes per signature each), then the
        // challenges (eight lanes per message) - the last two on the caller's stream
        uint8_t* ws = (uint8_t*)qws;
        uint32_t* hdr = (uint32_t*)(ws + ((idemix_quad_tables_bytes(n) + 255) & ~(size_t)255));
        uint8_t* st_tmp = (uint8_t*)hdr + idemix_rows(n) * (NYM_HDR_WORDS * 4);
        uint4* comb_rec = nullptr;
        uint32_t* comb_flags = nullptr;
        bool joined = true, side_launched = false, commit_launched = false;
        hipError_t commit_err = hipSuccess;
        auto launch_commitments = [&]() {
            hipLaunchKernelGGL((idemix_nym_verify_quad_kernel<VERIFY_BLOCK, false>), qgrid, qblock, lds_reserve, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                               (const uint32_t*)off, spans ? 1u : 0u, (const uint32_t*)issuer_id, (const IssuerDev*)issuers, n_issuers, (const uint8_t*)nym_x,
                               (const uint8_t*)nym_y, (const uint8_t*)proof_c, (const uint8_t*)s_sk, (const uint8_t*)s_rnym, (const uint8_t*)nonce,
                               (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status, (const uint32_t*)gather, hdr, st_tmp, comb_flags, (const uint4*)comb_rec);
            commit_err = hipGetLastError();
            commit_launched = true;
        };
        if (side != nullptr && side->stream != nullptr) {
            uint4* rec = (uint4*)(((uintptr_t)(st_tmp + idemix_rows(n)) + 255) & ~(uintptr_t)255);
            uint32_t* flags = (uint32_t*)(rec + idemix_rows(n) * 4 * NYM_COMB_UINT4_PER_LANE);
            // TEST HOOK NymSide::test_side_after (tests/test_idemix_gpu.py): the side launch is ordered BEHIND the commitment launch, so
            // that every wavefront of the commitment kernel finds no records, gives up (flag 0 -> 2) and computes its terms itself, and
            // every wavefront of the side launch finds the 2 and skips - both halves of the fallback, deterministically.
            const bool side_after = side->test_side_after;       // (set through the test-hook library only: fabgpu_test_nym_side_after)
            // flags to zero in stream order, the side stream behind them; any failure on the way: the commitment kernel simply does it all
            if (hipMemsetAsync(flags, 0, idemix_rows(n) / 16 * 4, st) == hipSuccess) {
                comb_rec = rec;
                comb_flags = flags;
                if (side_after) launch_commitments();
                if (hipEventRecord(side->fork, st) == hipSuccess && hipStreamWaitEvent(side->stream, side->fork, 0) == hipSuccess) {
                    hipLaunchKernelGGL(idemix_nym_comb_quad_kernel<VERIFY_BLOCK>, qgrid, qblock, 0, side->stream, n, (const uint32_t*)issuer_id, (const IssuerDev*)issuers,
                                       n_issuers, (const uint8_t*)s_sk, (const uint8_t*)s_rnym, (const uint32_t*)gather, flags, rec);
                    (void)hipGetLastError();   // (a side launch that did not happen leaves the flags at zero: the commitment kernel computes the terms itself)
                    // the caller's stream must not run past this call's kernels before the side launch is through with the workspace
                    side_launched = true;
                    joined = hipEventRecord(side->join, side->stream) == hipSuccess;
                    if (!joined) hipStreamSynchronize(side->stream);
                }
            }
        }
        if (!commit_launched) launch_commitments();
        if (side_launched && joined && hipStreamWaitEvent(st, side->join, 0) != hipSuccess) hipStreamSynchronize(side->stream);
        if (commit_err != hipSuccess) return commit_err;
        constexpr int W = 4;
        dim3 cgrid((uint32_t)(idemix_rows(n) / (W * SHAC_PER_WAVE))), cblock(64 * W);
        hipLaunchKernelGGL(idemix_nym_challenge_coop_kernel<W>, cgrid, cblock, (size_t)W * SHAC_LDS_WORDS * 4, st, n, (const uint32_t*)arena,
                           (uint32_t)((arena_bytes + 3) / 4), (const uint32_t*)off, spans ? 1u : 0u, (const uint8_t*)proof_c, (const uint8_t*)nonce,
                           (const uint32_t*)hdr, (const uint8_t*)st_tmp, (uint64_t*)verdict_bits, (uint8_t*)status, (const uint32_t*)gather);
        return hipGetLastError();
    }
    VerifyGeom g = verify_geom(n, allow_split);
    dim3 grid(g.wgs), block(g.block);
    if (g.pair) {
        hipLaunchKernelGGL(idemix_nym_verify_split_kernel<VERIFY_BLOCK>, grid, block, lds_reserve, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                           (const uint32_t*)off, spans ? 1u : 0u, (const uint32_t*)issuer_id, (const IssuerDev*)issuers, n_issuers, (const uint8_t*)nym_x,
                           (const uint8_t*)nym_y, (const uint8_t*)proof_c, (const uint8_t*)s_sk, (const uint8_t*)s_rnym, (const uint8_t*)nonce,
                           (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status, (const uint32_t*)gather);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(idemix_nym_verify_kernel<VERIFY_BLOCK>, grid, block, lds_reserve, st, n, (const uint32_t*)arena, (uint32_t)((arena_bytes + 3) / 4),
                       (const uint32_t*)off, spans ? 1u : 0u, (const uint32_t*)issuer_id, (const IssuerDev*)issuers, n_issuers, (const uint8_t*)nym_x,
                       (const uint8_t*)nym_y, (const uint8_t*)proof_c, (const uint8_t*)s_sk, (const uint8_t*)s_rnym, (const uint8_t*)nonce,
                       (uint4*)qws, (uint64_t*)verdict_bits, (uint8_t*)status, (const uint32_t*)gather);
    return hipGetLastError();
}

// see warm_kernel_functions_kernels (kernels.hip)
int warm_kernel_functions_idemix() {
    int ok = 0;
    hipFuncAttributes a;
    const void* fns[] = {(const void*)idemix_nym_verify_kernel<VERIFY_BLOCK>, (const void*)idemix_nym_verify_split_kernel<VERIFY_BLOCK>,
                         (const void*)idemix_nym_verify_quad_kernel<VERIFY_BLOCK, true>, (const void*)idemix_nym_verify_quad_kernel<VERIFY_BLOCK, false>,
                         (const void*)idemix_nym_challenge_coop_kernel<4>, (const void*)idemix_nym_comb_quad_kernel<VERIFY_BLOCK>};
    for (const void* f : fns) ok += hipFuncGetAttributes(&a, f) == hipSuccess ? 1 : 0;
    return ok;
}

}  // namespace fab
