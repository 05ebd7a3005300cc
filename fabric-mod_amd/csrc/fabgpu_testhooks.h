/*
 * fabgpu_testhooks.h - what libfabgpu_testhooks.so exports: probes, walker-against-walker comparisons, the synthetic block generator and
 * the kernel timer.  TEST / BENCH INFRASTRUCTURE, not part of the drop-in boundary: nothing here is declared in the public headers (include/) or exported
 * by libfabgpu.so (tests/test_host_logic.py checks both), and nothing in the product path loads this library.  The functions keep the
 * names they had while they lived in the product library (rounds 1-5).
 */
#ifndef FABGPU_TESTHOOKS_H
#define FABGPU_TESTHOOKS_H
#include "../../include/fabgpu.h"
#include "../../include/fabgpu_bccsp.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Duration in milliseconds of the most recent kernel launched through ctx, measured with HIP events on the
 * launch stream.  Only for contexts created with FABGPU_FLAG_TIME_KERNELS; <0 otherwise / if nothing was launched. */
float fabgpu_last_kernel_ms(fabgpu_ctx* ctx);
/* TEST HOOKS: the comb table of registered key `key_id` as it lies on the device (built there since round 6: keytab_kernels.hip), and the
 * table the host builder (p256_tables29.h) makes for a key - 32 windows x 256 entries x 20 words; the two must be byte-identical. */
int fabgpu_test_key_table(fabgpu_ctx* ctx, uint32_t key_id, int32_t* out_words, size_t cap_words);
int fabgpu_test_key_table_host(const uint8_t* qx32, const uint8_t* qy32, int32_t* out_words, size_t cap_words);
/* TEST HOOK: the context's generator comb (80 MiB, built on the device at fabgpu_init since round 6) against the host builder's: the
 * index of the first differing 32-bit word, -1 identical, -2 error */
long long fabgpu_test_gtab_compare_with_host(fabgpu_ctx* ctx);
/* FABGPU_FLAG_KEY_TABLES_16BIT: waits for the queued builds; the number of 16-bit key tables, after key_id's was checked entry for entry against
 * the key's 8-bit table where the two overlap (-1: the key has none, -2: error, -(1000 + w): window w disagrees) */
long long fabgpu_test_key_tables16(fabgpu_ctx* ctx, uint32_t key_id);
/* TEST HOOK: while on, the idemix four-lane form queues its side launch (the fixed-base terms) BEHIND the commitment launch, so that
 * every commitment wavefront gives up on its records and computes the terms itself, and every side wavefront skips its rows. */
void fabgpu_test_nym_side_after(fabgpu_ctx* ctx, int on);


/* Synthetic block generator: n tuples, fresh P-256 keypair per signature, low-S, `invalid_permille`
 * of them mutated (equal parts 1: flipped digest bit, 2: wrong key, 3: s -> n-s, 4: r+1); kind[i] in 0..4 records the
 * mutation.  e_in (n x 32) gives the digests to sign (e.g. SHA-256 of synthetic messages computed by
 * fabgpu_sha256_batch); NULL draws random digests.  e_out receives the digest the verifier should be given (for kind 1
 * it differs from the signed one in one bit; callers in hash mode flip a message bit instead).  Pure host code,
 * deterministic in (seed, n, e_in). */
int fabgpu_synth_batch(size_t n, uint64_t seed, uint32_t invalid_permille, const uint8_t* e_in, uint8_t* qx, uint8_t* qy,
                       uint8_t* e_out, uint8_t* r, uint8_t* s, uint8_t* kind, int threads);


/* TEST HOOK: the device walker against the host walker on one block, record for record.  0 identical (*declined = 1: the device walk
 * declined the block, `diff` says why), 1 they differ (`diff` says where). */
int fabgpu_csp_block_walk_compare(fabgpu_csp* csp, const uint8_t* block, size_t len, int* declined, char* diff, size_t cap);
/* TEST HOOKS (pure host, no device): the device walk's two-run procedure (count, prefix sum, write) carried out serially on the host
 * and compared with the host walker (0 identical, 1 different, FABGPU_EINVAL framing refused); the device's signature gate (0 submit,
 * 1 high-S, 2 empty, 3 declined: the general parser decides); the identity-table hash. */
int fabgpu_block_walk_twopass_compare(const uint8_t* block, size_t len, char* diff, size_t cap);
int fabgpu_gate_sig_fast(const uint8_t* sig, size_t len, uint8_t* r32, uint8_t* s32);
/* ... and the gate the device route applies to every signature (the fast gate, then the general parser for what that declines):
 * 0 submit (r32 / s32 set), 1 high-S, 2 empty, 4 does not unmarshal or r, s <= 0, 5 r of more than 256 bits ((false, nil)) */
int fabgpu_gate_sig_any(const uint8_t* sig, size_t len, uint8_t* r32, uint8_t* s32);
/* TEST HOOK (device): the device route's identity decoder (one wavefront per identity) over n identities = arena[spans[2i], spans[2i+1]):
 * code 0 P-256 key (key[64 i ..] = X || Y), 1 not such an identity, 2 undecided (left to the host) */
int fabgpu_csp_idfix_probe(fabgpu_csp* csp, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* key);
/* TEST HOOK (device): the same gate in the wavefront form the kernels run, over n signatures = arena[spans[2i], spans[2i+1]) */
int fabgpu_csp_gate_probe(fabgpu_csp* csp, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* r, uint8_t* s);
uint64_t fabgpu_identity_table_hash(const uint8_t* p, size_t len);


#ifdef __cplusplus
}
#endif
#endif
