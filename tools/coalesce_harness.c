/* Native driver of the coalescer through the C ABI only (plain C99 + pthreads, like a cgo caller would behave): T threads, each calling
 * fabgpu_csp_verify_coalesced (bccsp.Verify) in a loop on its own tuples; prints calls per second, launches and the mean batch.
 *   gcc -O2 -std=gnu99 -Iinclude -Ifabric-mod_amd/csrc tools/coalesce_harness.c -Lfabric-mod_amd/lib -lfabgpu_testhooks -lfabgpu -lpthread -o /tmp/coalesce
 *   (the synthetic signatures come from the test-hook library: fabgpu_testhooks.h)
 *   LD_LIBRARY_PATH=fabric-mod_amd/lib /tmp/coalesce <threads> <calls per thread> [window_us]                                        */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fabgpu_bccsp.h"
#include "fabgpu_testhooks.h"

#define N 4096
static uint8_t qx[N * 32], qy[N * 32], e[N * 32], r[N * 32], s[N * 32], kind[N];
static uint8_t der[N][80];
static size_t derlen[N];
static fabgpu_csp* csp;
static int calls_per_thread, nthreads;
static volatile int wrong = 0, infra = 0;

static size_t der_int(uint8_t* out, const uint8_t* be32) {
    int i = 0;
    while (i < 31 && be32[i] == 0) i++;
    size_t n = 32 - i, pad = (be32[i] & 0x80) ? 1 : 0;
    out[0] = 0x02;
    out[1] = (uint8_t)(n + pad);
    if (pad) out[2] = 0;
    memcpy(out + 2 + pad, be32 + i, n);
    return 2 + pad + n;
}

static void* worker(void* arg) {
    long w = (long)arg;
    char err[256];
    for (int c = 0; c < calls_per_thread; c++) {
        size_t j = ((size_t)w * 131 + (size_t)c * 7) % N;
        int valid = -1, flags = 0;
        int rc = fabgpu_csp_verify_coalesced(csp, qx + 32 * j, qy + 32 * j, der[j], derlen[j], e + 32 * j, 32, &valid, &flags, err, sizeof err);
        if (rc != 0) __sync_fetch_and_add(&infra, 1);
        else if ((valid == 1) != (kind[j] == 0)) __sync_fetch_and_add(&wrong, 1);
    }
    return NULL;
}

int main(int argc, char** argv) {
    nthreads = argc > 1 ? atoi(argv[1]) : 256;
    calls_per_thread = argc > 2 ? atoi(argv[2]) : 200;
    uint32_t window = argc > 3 ? (uint32_t)atoi(argv[3]) : 50;
    char err[256];
    if (fabgpu_csp_new(NULL, &csp, err, sizeof err) != 0) { fprintf(stderr, "csp: %s\n", err); return 2; }
    if (fabgpu_synth_batch(N, 77, 100, NULL, qx, qy, e, r, s, kind, 8) != 0) return 3;
    for (int i = 0; i < N; i++) {
        uint8_t* d = der[i];
        size_t lr = der_int(d + 2, r + 32 * i), ls = der_int(d + 2 + lr, s + 32 * i);
        d[0] = 0x30;
        d[1] = (uint8_t)(lr + ls);
        derlen[i] = 2 + lr + ls;
    }
    fabgpu_csp_coalescer_configure(csp, window, 32768);
    pthread_t* th = malloc(sizeof(pthread_t) * nthreads);
    struct timespec t0, t1;
    uint64_t c0, l0, g0, c1, l1, g1;
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, 256 << 10);
    /* warm-up */
    { int valid, flags; fabgpu_csp_verify_coalesced(csp, qx, qy, der[0], derlen[0], e, 32, &valid, &flags, err, sizeof err); }
    fabgpu_csp_coalescer_stats(csp, &c0, &l0, &g0);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (long w = 0; w < nthreads; w++) pthread_create(&th[w], &at, worker, (void*)w);
    for (long w = 0; w < nthreads; w++) pthread_join(th[w], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fabgpu_csp_coalescer_stats(csp, &c1, &l1, &g1);
    double dt = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    /* kind 3 (s -> n - s) is decided by the host's low-S gate and never reaches the coalescer */
    printf("{\"threads\": %d, \"calls\": %ld, \"calls_per_s\": %.0f, \"coalesced_calls\": %llu, \"launches\": %llu, \"mean_batch\": %.1f, \"largest_batch\": %llu, "
           "\"wrong\": %d, \"infrastructure_errors\": %d, \"window_us\": %u}\n",
           nthreads, (long)nthreads * calls_per_thread, (double)nthreads * calls_per_thread / dt, (unsigned long long)(c1 - c0),
           (unsigned long long)(l1 - l0), (double)(c1 - c0) / (double)(l1 - l0 ? l1 - l0 : 1), (unsigned long long)g1, wrong, infra, window);
    fabgpu_csp_free(csp);
    return wrong || infra ? 1 : 0;
}
