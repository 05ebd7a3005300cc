// Launch entry points of kernels.hip (host side).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace fab {
constexpr int VERIFY_BLOCK = 256;  // 4 wavefronts share one LDS copy of the generator comb table

hipError_t launch_sha256_batch(uint32_t n, const void* arena, size_t arena_bytes, const void* off, void* digests, hipStream_t st);
hipError_t launch_p256_verify(uint32_t n, const void* qx, const void* qy, const void* e, const void* r, const void* s,
                              const void* gtab, void* verdict_bits, void* status, hipStream_t st);
hipError_t launch_sha256_p256_verify(uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* qx,
                                     const void* qy, const void* r, const void* s, const void* gtab, void* verdict_bits,
                                     void* status, hipStream_t st);
}  // namespace fab
