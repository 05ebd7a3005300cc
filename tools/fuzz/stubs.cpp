// Link-time stand-ins for the device entry points: the fuzzers exercise host-side parsers only (no GPU, no libfabgpu.so).
#include "../../include/fabgpu.h"
extern "C" {
int fabgpu_idemix_issuer_register(fabgpu_ctx*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint32_t*) { return -1; }
int fabgpu_idemix_nym_verify_batch(fabgpu_ctx*, size_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint8_t*, const uint8_t*, const uint8_t*,
                                   const uint8_t*, const uint8_t*, const uint8_t*, uint64_t*, uint8_t*) { return -1; }
const char* fabgpu_strerror(int) { return "stub"; }
int fabgpu_init(const fabgpu_cfg*, fabgpu_ctx**) { return -1; }
void fabgpu_shutdown(fabgpu_ctx*) {}
int fabgpu_p256_verify_batch(fabgpu_ctx*, size_t, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint64_t*, uint8_t*) { return -1; }
int fabgpu_p256_verify_batch_keyed(fabgpu_ctx*, size_t, const uint32_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint64_t*, uint8_t*) { return -1; }
int fabgpu_sha256_batch(fabgpu_ctx*, size_t, const uint8_t*, const uint32_t*, uint8_t*) { return -1; }
int fabgpu_sha256_p256_verify_batch(fabgpu_ctx*, size_t, const uint8_t*, const uint32_t*, const uint8_t*, const uint8_t*, const uint8_t*, const uint8_t*, uint64_t*, uint8_t*) { return -1; }
int fabgpu_sha256_p256_verify_batch_keyed(fabgpu_ctx*, size_t, const uint8_t*, const uint32_t*, const uint32_t*, const uint8_t*, const uint8_t*, uint64_t*, uint8_t*) { return -1; }
int fabgpu_p256_key_register(fabgpu_ctx*, const uint8_t*, const uint8_t*, uint32_t*) { return -1; }
int fabgpu_p256_key_lookup(fabgpu_ctx*, const uint8_t*, const uint8_t*, uint32_t*) { return 1; }
int fabgpu_identity_verify_batch(fabgpu_ctx*, const fabgpu_identity_batch*) { return -1; }
int fabgpu_arena_stage(fabgpu_ctx*, const void*, size_t, uint64_t*) { return -1; }
int fabgpu_device_count(fabgpu_ctx*) { return 0; }
int fabgpu_p256_key_register_many(fabgpu_ctx* const*, int, const uint8_t*, const uint8_t*, uint32_t*) { return -1; }
}
#include "block_walk_dev.h"
namespace fab {
int walk_idtab_set(fabgpu_ctx*, uint32_t, const DevIdEntry*, const uint8_t*, size_t, uint64_t) { return -1; }
int walk_block_pass(fabgpu_ctx*, WalkRequest&) { return -1; }
void* walk_pinned_alloc(fabgpu_ctx*, size_t) { return nullptr; }
void walk_pinned_free(fabgpu_ctx*, void*) {}
int walk_preallocate(fabgpu_ctx*, size_t, uint32_t, uint32_t, int) { return -1; }
double walk_warm_copies(fabgpu_ctx*, void* const*, const size_t*, int) { return -1; }
int walk_gate_probe(fabgpu_ctx*, uint32_t, const uint8_t*, size_t, const uint32_t*, uint8_t*, uint8_t*, uint8_t*) { return -1; }
size_t key_table_words() { return 1; }
bool key_table_build(const uint8_t*, const uint8_t*, int32_t*) { return false; }
int key_register_many_prebuilt(fabgpu_ctx* const*, int, const uint8_t*, const uint8_t*, const int32_t*, uint32_t*) { return -1; }
int key_register_batch(fabgpu_ctx*, int, const uint8_t*, uint32_t*) { return -1; }
int arena_stage_keep(fabgpu_ctx*, const void*, size_t, uint64_t*, HostCopy*) { return -1; }
void host_copy_release(HostCopy* c) { if (c) *c = HostCopy(); }
void host_copy_limit(fabgpu_ctx*, uint32_t) {}
void host_copy_preallocate(fabgpu_ctx*, size_t, uint32_t) {}
void host_copy_stats(fabgpu_ctx*, uint64_t* a, uint64_t* b, uint64_t* c) { if (a) *a = 0; if (b) *b = 0; if (c) *c = 0; }
}
