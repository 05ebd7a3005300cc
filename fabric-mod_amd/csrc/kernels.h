// Launch entry points of kernels.hip (host side).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace fab {
constexpr int KTAB_STRIDE = 2;       // the device's array of key tables: [2 k] the 8-bit comb of key k (CombTab<8>, 640 KiB), [2 k + 1] its 16-bit comb (80 MiB) or nullptr
constexpr int VERIFY_BLOCK = 256;         // 4 wavefronts per workgroup, one per SIMD
constexpr int VERIFY_MAX_WGS = 256;       // persistent workgroup slots: one per CU (see kernels.hip for why not two)
constexpr int VERIFY_PAIR_MAX = 32768;    // up to here two lanes per signature (one round of 256 workgroups x 128 signatures)
// workspace of the per-lane j*Q tables: 16 entries x 8 uint4 (one 128-byte line each; seven cells used) per lane or per signature
constexpr size_t QWS_UINT4_PER_LANE = (size_t)16 * 8;
constexpr size_t QWS_PAIR_UINT4_PER_SIG = (size_t)16 * 8;

// Shared message prefixes of a fused batch (device pointers): prefix p = arena[pre_off[p], pre_off[p+1]), message i continues
// prefix pre_idx[i] (0xFFFFFFFF = none); mid_scratch: m x 32 bytes for the mid-states.  m = 0 / pre_idx = nullptr: no prefixes.
struct ShaPrefixArgs {
    uint32_t m = 0;
    const void* pre_off = nullptr;
    const void* pre_idx = nullptr;
    void* mid_scratch = nullptr;
    bool spans = false;   // off / pre_off hold (start, end) pairs
    void* digests = nullptr;   // optional out (device): n x 32 bytes, the digest of every message
    bool mid_ready = false;    // the mid-states are already in mid_scratch (launch_sha256_midstates ran, e.g. on another stream)
    // Bytes of (unused) dynamic LDS to request with the launch.  More than half a CU's 160 KB keeps a second workgroup - of this or of
    // any other kernel - off the CU: two launches that run side by side on two streams then land on DISJOINT CUs instead of sharing
    // SIMDs (two waves on a SIMD take 1.5x the time each; the dispatcher fills a CU that still has room before it moves on).
    uint32_t lds_reserve = 0;
};
struct VerifyGeom {
    uint32_t block;   // threads per workgroup
    uint32_t wgs;     // workgroups launched (= workspace slots)
    bool pair;        // two lanes per signature
};
VerifyGeom verify_geom(uint32_t n, bool allow_pair);
size_t verify_workspace_bytes(uint32_t n, bool allow_pair);
int pair_table_default();
size_t pair_table_lds_bytes();   // dynamic LDS one workgroup of the LDS-table pair kernel asks for
constexpr int PAIR_TABLE_LDS_FROM = 16384;   // launches of more tuples than this keep the pair kernel's per-signature table in LDS
// the mid-state kernel of a prefixed batch alone (the fused launchers run it themselves unless pa.mid_ready)
hipError_t launch_sha256_midstates(const void* arena, size_t arena_bytes, const ShaPrefixArgs& pa, hipStream_t st);   // honours pa.lds_reserve
hipError_t launch_sha256_batch(uint32_t n, const void* arena, size_t arena_bytes, const void* off, void* digests, hipStream_t st, uint32_t lds_spread = 0);
// n messages given as (start, end) pairs -> n x 32 digest bytes
hipError_t launch_sha256_spans(uint32_t n, const void* arena, size_t arena_bytes, const void* spans, void* digests, hipStream_t st,
                               uint32_t lds_reserve = 0, uint32_t lds_spread = 0);   // lds_spread: for the eight-lane road (n <= SHA_COOP_MAX, no lds_reserve)
// gathered messages (pieces of the arena stitched into `scratch` at out_off[j] .. out_off[j+1]) -> n x 32 digest bytes
hipError_t launch_gather_sha256(uint32_t n, const void* arena, size_t arena_bytes, const void* spans, const void* out_off, void* scratch,
                                size_t scratch_bytes, void* digests, hipStream_t st, uint32_t lds_reserve = 0, uint32_t lds_spread = 0);
hipError_t launch_p256_verify(uint32_t n, const void* qx, const void* qy, const void* e, const void* r, const void* s,
                              const void* gtab, void* qws, void* verdict_bits, void* status, bool allow_pair, hipStream_t st,
                              uint32_t lds_reserve = 0, int table_lds = 0);   // lds_reserve: see ShaPrefixArgs; table_lds: 1 PairQTabLds, 0 global, -1 by size
hipError_t launch_sha256_p256_verify(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* qx,
                                     const void* qy, const void* r, const void* s, const void* gtab, void* qws,
                                     void* verdict_bits, void* status, bool allow_pair, const ShaPrefixArgs& pa, hipStream_t st);
hipError_t launch_p256_verify_keyed(uint32_t n, const void* key_id, uint32_t nkeys, const void* ktabs, const void* e, const void* r, const void* s,
                                    const void* gtab, void* verdict_bits, void* status, bool allow_pair, hipStream_t st, uint32_t lds_reserve = 0);
hipError_t launch_sha256_p256_verify_keyed(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* key_id, uint32_t nkeys,
                                           const void* ktabs, const void* r, const void* s, const void* gtab, void* verdict_bits, void* status,
                                           bool allow_pair, const ShaPrefixArgs& pa, hipStream_t st);

// ---- wide_kernels.hip: a registered key's verification on eight lanes per signature, in two phases (p256_wide29.h) ----
// For launches of at most WIDE_LAUNCH_MAX signatures (they cannot fill the chip: their time is one wavefront's instruction stream).
// pre: everything that does not need the digest (w = s^-1, u2 = r w, T = u2 Q) -> 144 bytes of scratch per signature; it can run while the
// messages are still being hashed.  post: e = the 32-byte digests by row -> verdict bits (ceil(n / 64) words, written bytewise) and status.
constexpr int WIDE_LAUNCH_MAX = 8192;
constexpr size_t WIDE_SCRATCH_BYTES = 144;
// lds_spread: how many wavefronts of the small launches a CU should take at most - spread_waves_per_cu(the wavefronts that run at the same
// time, over all launches that do) = 1, 2, 4, 8 or SPREAD_NONE; a launch turns it into unused dynamic LDS per workgroup
// (wide_kernels.hip "PLACEMENT").  0 = from this launch's own wavefront count.
constexpr uint32_t SPREAD_NONE = 0xFFFFFFFFu;
uint32_t spread_waves_per_cu(uint32_t wavefronts);
hipError_t launch_p256_wide_pre(uint32_t n, const void* key_id, uint32_t nkeys, const void* ktabs, const void* r, const void* s, const void* gtab,
                                void* scratch, hipStream_t st, uint32_t lds_spread = 0);
hipError_t launch_p256_wide_post(uint32_t n, const void* e, const void* r, const void* gtab, const void* scratch, void* verdict_bits, void* status,
                                 hipStream_t st, uint32_t lds_spread = 0);
// SHA-256 of n (possibly prefixed: pa.mid_scratch must hold the mid-states) messages -> pa.digests (n x 32 bytes), one message per lane
hipError_t launch_sha256_messages(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const ShaPrefixArgs& pa, hipStream_t st);
// The same digests with eight lanes on a message (sha256_coop.h) and NO mid-states: a prefixed message is hashed whole.  For launches that
// cannot fill the chip - SHA_COOP_MAX messages are 256 wavefronts; launch_sha256_batch / _spans (without an LDS reservation) take this road
// by themselves up to that size.
constexpr uint32_t SHA_COOP_MAX = 2048;
// Larger launches: one lane per message, except that messages well above the launch's average length (a quarter above arena_bytes / n,
// at least 2 KiB) are hashed on eight lanes each - at most eight per 64 messages - by wavefronts of their own inside the same launch
// (wide_kernels.hip sha256_mixed_kernel).  pairs: off holds (start, end) pairs instead of n + 1 consecutive offsets.  lds_reserve: see
// ShaPrefixArgs.
hipError_t launch_sha256_mixed(uint32_t n, const void* arena, size_t arena_bytes, const void* off, bool pairs, void* digests, hipStream_t st,
                               uint32_t lds_reserve = 0);
hipError_t launch_sha256_messages_coop(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const ShaPrefixArgs& pa, hipStream_t st,
                                       uint32_t lds_spread = 0);

// ---- idemix_kernels.hip: idemix pseudonym signatures on FP256BN ----
// A registered issuer occupies one slot of idemix_issuer_dev_bytes() bytes in a device array; fill a host copy of the slot
// with idemix_issuer_dev_fill (device pointers of the two comb tables + ipk.Hash) and copy it up.
size_t idemix_issuer_dev_bytes();
void idemix_issuer_dev_fill(void* host_slot, const void* d_hsk, const void* d_hrand, const uint8_t hash32[32]);
// one signature per lane; (allow_split and n <= VERIFY_PAIR_MAX) two lanes per signature; (allow_split, allow_quad and
// n <= IDEMIX_QUAD_MAX) four lanes per signature, every point operation on a lane pair; workspace: idemix_workspace_bytes
constexpr int IDEMIX_QUAD_MAX = 16384;     // 256 workgroups x 64 signatures: one round of the chip
// a second stream for the fixed-base terms of the four-lane form (idemix_nym_comb_quad_kernel runs beside the commitment kernel) and the
// two events that fork it off the caller's stream and join it again; owned by the caller (fabgpu_ctx keeps one per workspace slot)
// The fixed-base terms of the idemix four-lane form run on a side stream BESIDE the commitment launch only while that launch leaves
// SIMDs free: at 12 288 signatures it occupies 768 of the chip's 1 024; beyond, the side launch cannot be co-resident, the commitment
// wavefronts spin for records that are not coming and then compute the terms themselves.  Measured (round 6,
// tools/bench_cfg5_mixed.py, idemix alone): 10 000 signatures 0.745 ms with the side stream / 0.810 without; 16 000: 1.010 / 0.815.
constexpr uint32_t NYM_SIDE_STREAM_MAX = 12288;
struct NymSide {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool test_side_after = false;   // TEST HOOK: order the side launch behind the commitment launch (both halves of the fallback, deterministically)
};
hipError_t launch_idemix_nym_verify(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* issuer_id, const void* issuers,
                                    uint32_t n_issuers, const void* nym_x, const void* nym_y, const void* proof_c, const void* s_sk,
                                    const void* s_rnym, const void* nonce, void* qws, void* verdict_bits, void* status, bool allow_split,
                                    bool allow_quad, bool spans, hipStream_t st,    // spans: off = n x (start, end) instead of n + 1 running offsets
                                    const void* gather = nullptr,                 // gather: row i takes its inputs from row gather[i] (uint32; ~0 = an idle row)
                                    uint32_t lds_reserve = 0,                     // unused LDS asked for: keeps other reserving kernels off this one's CUs
                                    bool two_phase = true,                        // four-lane form: commitments, then the challenges with eight lanes on a message
                                    const NymSide* side = nullptr);               // ... and the fixed-base terms beside the commitments on this stream
// every LANE owns a 16-entry table in the one- and two-lane geometries, every lane PAIR in the four-lane one
size_t idemix_workspace_bytes(uint32_t n, bool allow_split, bool allow_quad);
// every kernel function of a translation unit resolved now instead of at its first launch (GPUCSP::Preallocate); returns how many
int warm_kernel_functions_kernels();
int warm_kernel_functions_wide();
int warm_kernel_functions_idemix();
int warm_kernel_functions_walk();
int warm_kernel_functions_keytab();
// keytab_kernels.hip: comb tables of registered P-256 keys built on the device (qxy, tabs: device memory; see the unit's header)
size_t keytab_scratch_bytes(uint32_t n_keys);
hipError_t launch_keytab_build(uint32_t n_keys, const void* qxy, void* const* tabs, void* scratch, hipStream_t st);
// ... and the generator's 16-bit comb (80 MiB) at fabgpu_init
size_t gtab_scratch_bytes();
hipError_t launch_gtab_build(void* d_tab, void* scratch, hipStream_t st);
// ... and a registered key's 16-bit comb in the same format (FABGPU_FLAG_KEY_TABLES_16BIT); scratch: keytab16_scratch_bytes() per key
size_t keytab16_scratch_bytes();
hipError_t launch_keytab16_build(uint32_t n_keys, const void* qxy, void* const* tabs, void* scratch, hipStream_t st);
}  // namespace fab
