/*
 * go_call_replay.c - what the reference-side Go binding does to libfabgpu.so, call for call, as a C program (the build image has
 * no Go toolchain: VERDICT r3 item 2 asked that the bench take the binding's EXACT call sequence instead of the ctypes one).
 *
 * The sequence (fabric-mod_amd/go/...):
 *   bccsp/gpu/gpu.go New                          fabgpu_csp_new2 (Devices, ConcurrentPasses, ExpectBlockBytes, ExpectTuples)
 *   extensions/gossip/state/preverify_on_arrival.go  a block ARRIVES (gossip/state/state.go:328,785-787): on a goroutine of its own
 *       pre.HasBlock(seq)                         fabgpu_csp_memo_has_block
 *       pre.PreVerifyBlock(data, seq)             fabgpu_csp_block_preverify2, FABGPU_PASS_SEED_MEMO, tx_flags only, cap_tx as remembered
 *                                                 by the provider (atomic max; FABGPU_ETOOBIG -> grow, retry: the library kept the upload)
 *   extensions/validation/preverify.go Validate   the committer reaches the block:
 *       pre.HasBlock(seq)                         fabgpu_csp_memo_has_block            (true: nothing to marshal, nothing to submit)
 *       the unchanged validator, validatorPoolSize goroutines (core/peer/config.go:255-257: NumCPU), one transaction each
 *       (core/committer/txvalidator/v20/validator.go:194-210); per signature identity.Verify (msp/identities.go:169-196):
 *           digest = bccsp.Hash(msg)              gpu.go Hash: fabgpu_csp_hash_lookup - the digest the pass computed, handed out only when every
 *                                                 byte of msg equals the block's; a miss -> bccsp/sw's SHA-256 on the CPU (here OpenSSL's)
 *                                                 (argument 5 = 0: the round-5 provider, whose Hash always went to bccsp/sw)
 *           bccsp.Verify(k, sig, digest)          fabgpu_csp_memo_lookup  ((true, nil) on a valid hit; else bccsp/sw - counted, not run)
 *       pre.EvictBlock(seq)                       fabgpu_csp_memo_evict_block
 * Block k + 1 arrives (its pass runs) while block k is validated: the steady state of one channel.
 *
 * What it prints (one JSON line): the pass as the binding calls it (first block of a fresh provider, i.e. over its caps; warm blocks),
 * the validators' CPU residue per block (hashing + memo lookups on T threads), and end-to-end validated tx/s of the two-stage pipeline.
 *
 * usage: go_call_replay <block file> [blocks=12] [validator threads=16] [devices=1] [hash memo=1] [arrivals in flight=1] [blocks pre-verified ahead=arrivals]
 * Test / bench infrastructure: links the product library through its C ABI only (include/fabgpu*.h) + libcrypto for the CPU SHA-256.
 */
#define _GNU_SOURCE
#include <openssl/sha.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fabgpu.h"
#include "fabgpu_bccsp.h"

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

typedef struct {
    uint32_t tx;
    uint8_t kind;
    uint32_t ident_off, ident_len, pre_off, pre_len, suf_off, suf_len, sig_off, sig_len;
    uint8_t xy[64];
    int has_key;
} tuple_t;

static fabgpu_csp* g_csp;
static uint32_t g_cap_tx = 1024; /* gpu.go Provider.capTx */
static int g_hash_memo = 1;      /* gpu.go Provider.Hash asks the digest memo first (0: round 5's pass-through to bccsp/sw) */

/* gpu.go PreVerifyBlock */
static int pre_verify_block(const uint8_t* blk, size_t len, uint64_t seq, uint32_t* n_tx, uint32_t* seeded, int* retries) {
    *retries = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        uint32_t cap = __atomic_load_n(&g_cap_tx, __ATOMIC_RELAXED);
        uint8_t* flags = (uint8_t*)calloc(cap ? cap : 1, 1); /* make([]uint8, capTx) */
        fabgpu_block_pass ps;
        memset(&ps, 0, sizeof(ps));
        ps.block = blk;
        ps.len = len;
        ps.block_seq = seq;
        ps.flags = FABGPU_PASS_SEED_MEMO;
        ps.cap_tx = cap;
        ps.cap_tuples = 0;
        ps.tx_flags = flags;
        int rc = fabgpu_csp_block_preverify2(g_csp, &ps);
        free(flags);
        if (rc == FABGPU_ETOOBIG) {
            uint32_t want = ps.n_tx + ps.n_tx / 8 + 16, cur;
            do {
                cur = __atomic_load_n(&g_cap_tx, __ATOMIC_RELAXED);
            } while (cur < want && !__atomic_compare_exchange_n(&g_cap_tx, &cur, want, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
            (*retries)++;
            continue;
        }
        if (rc != 0) return rc;
        *n_tx = ps.n_tx;
        *seeded = ps.memo_seeded;
        return 0;
    }
    return FABGPU_EINVAL;
}

/* ---- the validator pool: validatorPoolSize workers, a transaction at a time ---- */
typedef struct {
    const uint8_t* blk;
    const tuple_t* tuples;
    const uint32_t* tx_first; /* n_tx + 1: first tuple of each transaction */
    uint32_t n_tx;
    uint32_t next;            /* atomic */
    uint64_t hits, misses, hashed_bytes, hash_hits, hash_bytes;
    int check;                /* also hash every message on the CPU and compare (block 0 only: outside the medians) */
    uint64_t digest_mismatches;
    double t_cat, t_hash, t_memo, t_thread; /* GO_REPLAY_PROFILE: thread-milliseconds in the copy, in Hash, in Verify, in the worker */
} validate_job;
static int g_profile;
static pthread_mutex_t g_prof_mu = PTHREAD_MUTEX_INITIALIZER;

/* bccsp/sw's Hash: SHA-256 on the CPU.
 * (SHA256_Init / Update / Final, not the one-shot SHA256(): OpenSSL 3 routes that through an EVP fetch per call, whose locks
 *  sixteen threads fight over - 42 ms per block instead of 6) */
static void sw_hash(const uint8_t* msg, size_t n, uint8_t* digest) {
    SHA256_CTX c;
    SHA256_Init(&c);
    SHA256_Update(&c, msg, n);
    SHA256_Final(digest, &c);
}
/* gpu.go Provider.Hash(msg, &bccsp.SHA256Opts{}); returns 1 if the digest memo answered */
static int provider_hash(const uint8_t* msg, size_t n, uint8_t* digest) {
    if (g_hash_memo && n >= 64 && fabgpu_csp_hash_lookup(g_csp, msg, n, digest) == 0) return 1;
    sw_hash(msg, n, digest);
    return 0;
}

static void validate_some(validate_job* j);
/* The pool is the peer's: validatorPoolSize workers that live as long as the channel does (a goroutine per transaction behind a semaphore of
 * that size, core/committer/txvalidator/v20/validator.go:194-210, on GOMAXPROCS long-lived OS threads) - not threads made per block.  Two
 * barriers per block: "here is a block", "the block is done". */
static pthread_barrier_t g_go, g_done;
static validate_job* volatile g_job;
static volatile int g_quit;
static void* validator(void* arg) {
    (void)arg;
    for (;;) {
        pthread_barrier_wait(&g_go);
        if (g_quit) return NULL;
        validate_some(g_job);
        pthread_barrier_wait(&g_done);
    }
}
static void validate_some(validate_job* j) {
    uint64_t hits = 0, misses = 0, bytes = 0, hh = 0, hb = 0;
    uint8_t* cat = NULL;
    size_t cat_cap = 0;
    double p_cat = 0, p_hash = 0, p_memo = 0, p0 = g_profile ? now_ms() : 0, pa = 0, pb = 0;
    for (;;) {
        uint32_t t = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (t >= j->n_tx) break;
        for (uint32_t i = j->tx_first[t]; i < j->tx_first[t + 1]; i++) {
            const tuple_t* tp = &j->tuples[i];
            uint8_t digest[32];
            /* msp/identities.go:173-181: digest = bccsp.Hash(msg).  A creator signs Envelope.payload; an endorser prp || endorser
             * (validator_keylevel.go:246-258 builds that concatenation: append(prp, endorser...) - a copy, then one hash) */
            const uint8_t* msg;
            size_t n;
            if (g_profile) pa = now_ms();
            if (tp->pre_len == 0) {
                msg = j->blk + tp->suf_off;
                n = tp->suf_len;
            } else {
                n = (size_t)tp->pre_len + tp->suf_len;
                if (cat_cap < n) {
                    free(cat);
                    cat = (uint8_t*)malloc(n + n / 2);
                    cat_cap = n + n / 2;
                }
                memcpy(cat, j->blk + tp->pre_off, tp->pre_len);
                memcpy(cat + tp->pre_len, j->blk + tp->suf_off, tp->suf_len);
                msg = cat;
            }
            if (g_profile) { pb = now_ms(); p_cat += pb - pa; }
            if (provider_hash(msg, n, digest)) {
                hh++;
                hb += n;
                if (j->check) {
                    uint8_t want[32];
                    sw_hash(msg, n, want);
                    if (memcmp(want, digest, 32) != 0) __atomic_fetch_add(&j->digest_mismatches, 1, __ATOMIC_RELAXED);
                }
            }
            bytes += (uint64_t)tp->pre_len + tp->suf_len;
            if (g_profile) { pa = now_ms(); p_hash += pa - pb; }
            uint8_t st = 255;
            /* gpu.go Verify: xyOf(k) -> fabgpu_csp_memo_lookup; (true, nil) only on a valid hit */
            if (tp->has_key && fabgpu_csp_memo_lookup(g_csp, tp->xy, tp->xy + 32, j->blk + tp->sig_off, tp->sig_len, digest, 32, &st) == 0 && st == FABGPU_ST_VALID) hits++;
            else misses++; /* -> bccsp/sw (not run here: the replay is about the path's own cost) */
            if (g_profile) p_memo += now_ms() - pa;
        }
    }
    free(cat);
    if (g_profile) {
        pthread_mutex_lock(&g_prof_mu);
        j->t_cat += p_cat; j->t_hash += p_hash; j->t_memo += p_memo; j->t_thread += now_ms() - p0;
        pthread_mutex_unlock(&g_prof_mu);
    }
    __atomic_fetch_add(&j->hits, hits, __ATOMIC_RELAXED);
    __atomic_fetch_add(&j->misses, misses, __ATOMIC_RELAXED);
    __atomic_fetch_add(&j->hashed_bytes, bytes, __ATOMIC_RELAXED);
    __atomic_fetch_add(&j->hash_hits, hh, __ATOMIC_RELAXED);
    __atomic_fetch_add(&j->hash_bytes, hb, __ATOMIC_RELAXED);
}

/* ---- the arrival goroutine of one block ---- */
typedef struct {
    const uint8_t* blk;
    size_t len;
    uint64_t seq;
    double ms;
    int rc, retries;
    uint32_t n_tx, seeded;
} arrival_job;

static void* arrival(void* arg) {
    arrival_job* a = (arrival_job*)arg;
    double t0 = now_ms();
    uint64_t have = 0;
    fabgpu_csp_memo_has_block(g_csp, a->seq, &have); /* preverify_on_arrival.go: a duplicate? */
    a->rc = have ? 0 : pre_verify_block(a->blk, a->len, a->seq, &a->n_tx, &a->seeded, &a->retries);
    a->ms = now_ms() - t0;
    return NULL;
}

/* ---- the arrival goroutines of the channel (see main) ---- */
static arrival_job* g_arr;
static int g_n_blocks, g_next_arrival, g_validation_started, g_arrivals = 1, g_window = 1;
static unsigned char g_arr_done[4096];
static pthread_mutex_t g_arr_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_arr_cv = PTHREAD_COND_INITIALIZER;
static void* arrival_worker(void* arg) {
    (void)arg;
    for (;;) {
        pthread_mutex_lock(&g_arr_mu);
        const int i = g_next_arrival;
        if (i >= g_n_blocks) {
            pthread_mutex_unlock(&g_arr_mu);
            return NULL;
        }
        g_next_arrival = i + 1;
        while (g_validation_started < i - g_window) pthread_cond_wait(&g_arr_cv, &g_arr_mu);
        pthread_mutex_unlock(&g_arr_mu);
        arrival(&g_arr[i]);
        pthread_mutex_lock(&g_arr_mu);
        g_arr_done[i] = 1;
        pthread_cond_broadcast(&g_arr_cv);
        pthread_mutex_unlock(&g_arr_mu);
    }
}

static int cmp_d(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }
static double median(double* v, int n) {
    qsort(v, (size_t)n, sizeof(double), cmp_d);
    return n ? (n & 1 ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2])) : 0;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s <block file> [blocks=12] [validator threads=16] [devices=1] [hash memo=1] [arrivals in flight=1] [window=arrivals]\n", argv[0]);
        return 2;
    }
    int n_blocks = argc > 2 ? atoi(argv[2]) : 12, n_thr = argc > 3 ? atoi(argv[3]) : 16, n_dev = argc > 4 ? atoi(argv[4]) : 1;
    g_hash_memo = argc > 5 ? atoi(argv[5]) != 0 : 1;
    g_arrivals = argc > 6 ? atoi(argv[6]) : 1;
    if (g_arrivals < 1) g_arrivals = 1;
    if (g_arrivals > 4) g_arrivals = 4;                /* maxArrivalPasses */
    g_window = argc > 7 ? atoi(argv[7]) : g_arrivals;  /* blocks that may be pre-verified ahead of the one being validated (gossip's payload buffer) */
    if (g_window < g_arrivals) g_window = g_arrivals;
    if (g_window > 5) g_window = 5;                    /* (the library's default memo holds six 40 000-signature blocks) */
    g_profile = getenv("GO_REPLAY_PROFILE") != NULL; /* (probes only: a clock read around every step of every signature) */
    if (n_blocks < 3) n_blocks = 3;
    if (n_blocks > 4096) n_blocks = 4096;
    if (n_thr < 1) n_thr = 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    size_t len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* blk = (uint8_t*)malloc(len);
    if (fread(blk, 1, len, f) != len) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);

    /* what the validators hold when they call identity.Verify: the tuples of the block (the Go side has them as unmarshalled
     * structures; the pure-host walker of the library lists the same byte ranges) and each identity's key (msp deserialization +
     * bccsp.KeyImport happened when the identity was first seen: gpu.go keeps X || Y under the key's SKI) - preparation, untimed */
    uint32_t n_tuples = 0, tail_len = 0, tail_base = 0;
    fabgpu_block_tuples(blk, len, 0, &n_tuples, NULL, NULL, NULL, NULL, 0, &tail_len, &tail_base);
    uint32_t* tx = (uint32_t*)malloc(4 * (size_t)n_tuples + 4);
    uint8_t* kind = (uint8_t*)malloc(n_tuples + 1);
    uint32_t* sp = (uint32_t*)malloc(32 * (size_t)n_tuples + 32);
    if (fabgpu_block_tuples(blk, len, n_tuples, &n_tuples, tx, kind, sp, NULL, 0, &tail_len, &tail_base) != 0) { fprintf(stderr, "block does not parse\n"); return 2; }
    tuple_t* tuples = (tuple_t*)calloc(n_tuples + 1, sizeof(tuple_t));
    uint32_t n_env_tuples = 0, n_tx = 0;
    for (uint32_t i = 0; i < n_tuples; i++) {
        if (kind[i] == 2) continue; /* orderer block signatures: MCS.VerifyBlock ran before AddPayload (internal/peer/gossip/mcs.go) */
        tuple_t* t = &tuples[n_env_tuples++];
        t->tx = tx[i];
        t->kind = kind[i];
        t->ident_off = sp[8 * i]; t->ident_len = sp[8 * i + 1];
        t->pre_off = sp[8 * i + 2]; t->pre_len = sp[8 * i + 3];
        t->suf_off = sp[8 * i + 4]; t->suf_len = sp[8 * i + 5];
        t->sig_off = sp[8 * i + 6]; t->sig_len = sp[8 * i + 7];
        if (tx[i] + 1 > n_tx) n_tx = tx[i] + 1;
        /* (few distinct identities: remember the last few decoded) */
        static struct { uint32_t off, len; uint8_t xy[64]; int ok; } memo[8];
        static int memo_n = 0;
        int found = -1;
        for (int k = 0; k < memo_n; k++)
            if (memo[k].len == t->ident_len && memcmp(blk + memo[k].off, blk + t->ident_off, t->ident_len) == 0) { found = k; break; }
        if (found < 0) {
            int k = memo_n < 8 ? memo_n++ : (int)(i % 8);
            memo[k].off = t->ident_off; memo[k].len = t->ident_len;
            memo[k].ok = fabgpu_identity_to_p256(blk + t->ident_off, t->ident_len, memo[k].xy) == 0;
            found = k;
        }
        t->has_key = memo[found].ok;
        memcpy(t->xy, memo[found].xy, 64);
    }
    uint32_t* tx_first = (uint32_t*)calloc(n_tx + 2, 4);
    for (uint32_t i = 0; i < n_env_tuples; i++) tx_first[tuples[i].tx + 1]++;
    for (uint32_t t = 0; t < n_tx; t++) tx_first[t + 1] += tx_first[t];

    /* gpu.go New */
    fabgpu_csp_opts o;
    memset(&o, 0, sizeof(o));
    o.size = sizeof(o);
    o.n_devices = n_dev;
    int32_t devs[64];
    for (int i = 0; i < 64; i++) devs[i] = 0; /* (n_dev contexts; ordinals 0..n-1 when the node has them) */
    int visible = fabgpu_device_count(NULL);
    for (int i = 0; i < n_dev && i < 64; i++) devs[i] = visible > 0 ? i % visible : 0;
    o.devices = devs;
    o.concurrent_passes = g_arrivals > 2 ? (uint32_t)g_arrivals : 2; /* GPUOpts.ConcurrentPasses */
    o.pass_hash_memo = g_hash_memo ? 0 : -1;   /* (GPUOpts.HashMemo: on unless the operator switched it off) */
    o.pass_timing = getenv("GO_REPLAY_PASS_TIMING") ? 1 : 0; /* stage breakdown of every pass on stderr (probes only) */
    o.expect_block_bytes = len + 4096;
    o.expect_tuples = n_tuples + 64;
    if (o.expect_tuples > g_cap_tx) g_cap_tx = o.expect_tuples; /* gpu.go New: capTx starts at max(1024, Options.ExpectTuples) */
    const uint32_t cap_tx_start = g_cap_tx;
    char err[256];
    double t_new = now_ms();
    int rc = fabgpu_csp_new2(&o, &g_csp, err, sizeof(err));
    if (rc != 0) {
        printf("{\"error\": \"fabgpu_csp_new2: %s (%s)\"}\n", fabgpu_strerror(rc), err);
        return rc == FABGPU_ENODEV ? 0 : 1;
    }
    t_new = now_ms() - t_new;

    /* every block a fresh copy: a peer's blocks arrive in memory the runtime has never seen */
    uint8_t** copies = (uint8_t**)malloc(sizeof(uint8_t*) * (size_t)n_blocks);
    for (int k = 0; k < n_blocks; k++) {
        copies[k] = (uint8_t*)malloc(len);
        memcpy(copies[k], blk, len);
    }
    arrival_job* arr = (arrival_job*)calloc((size_t)n_blocks, sizeof(arrival_job));
    double* val_ms = (double*)calloc((size_t)n_blocks, sizeof(double));
    double* has_ms = (double*)calloc((size_t)n_blocks, sizeof(double));
    uint64_t hits = 0, misses = 0, hashed = 0, hash_hits = 0, hash_bytes = 0, mismatches = 0;
    pthread_t* pool = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_thr);
    pthread_barrier_init(&g_go, NULL, (unsigned)n_thr + 1);
    pthread_barrier_init(&g_done, NULL, (unsigned)n_thr + 1);
    for (int t = 0; t < n_thr; t++) pthread_create(&pool[t], NULL, validator, NULL);

    /* The first blocks of a fresh process, one after the other.  Block 0 pays what a process pays once - code objects loaded at their
     * first launch, the signers' certificates decoded and their comb tables built - AND finds the provider's caps at the initial 1 024
     * transactions (FABGPU_ETOOBIG, grow, retry).  To price the over-caps retry ALONE the caps are put back to 1 024 once the provider
     * is warm (what a provider whose earlier blocks were all small looks like) and one more block is sent: `over_caps_on_a_warm_provider`. */
    for (int k = 0; k < n_blocks; k++) { arr[k].blk = copies[k]; arr[k].len = len; arr[k].seq = 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1); }
    arrival_job warmup[6];
    memset(warmup, 0, sizeof(warmup));
    double warm_ms[6];
    for (int k = 0; k < 6; k++) {
        warmup[k].blk = (uint8_t*)malloc(len);
        memcpy((void*)warmup[k].blk, blk, len);
        warmup[k].len = len;
        warmup[k].seq = 0xD6E8FEB86659FD93ull * (uint64_t)(k + 1);
        if (k == 5) __atomic_store_n(&g_cap_tx, 1024, __ATOMIC_RELAXED);
        arrival(&warmup[k]);
        warm_ms[k] = warmup[k].ms;
        uint64_t ev = 0;
        fabgpu_csp_memo_evict_block(g_csp, warmup[k].seq, &ev);
        if (warmup[k].rc != 0) { printf("{\"error\": \"warm-up pass %d: %s\"}\n", k, fabgpu_strerror(warmup[k].rc)); return 1; }
    }
    double warm4 = warm_ms[3] < warm_ms[4] ? warm_ms[3] : warm_ms[4];
    arrival(&arr[0]);
    /* The arrival goroutines (preverify_on_arrival.go: at most maxArrivalPasses = 4 passes in flight per channel; here g_arrivals of them):
     * block i's pass may start once the validation of block i - g_window has started.  Window 1, one arrival: block k + 1 arrives while
     * block k is validated (a channel that receives a block per validated block).  A wider window with two arrivals in flight: a channel
     * whose blocks arrive faster than it validates them (catch-up, a saturated orderer) - passes overlap on the device and the payload
     * buffer (gossip/state: blockBufferSize) holds pre-verified blocks for the committer. */
    g_arr = arr;
    g_arr_done[0] = 1;
    g_n_blocks = n_blocks;
    g_next_arrival = 1;
    g_validation_started = -1;
    pthread_t arr_th[8];
    for (int a = 0; a < g_arrivals; a++) pthread_create(&arr_th[a], NULL, arrival_worker, NULL);
    double wall0 = now_ms();
    for (int k = 0; k < n_blocks; k++) {
        pthread_mutex_lock(&g_arr_mu);
        g_validation_started = k;                           /* blocks up to k + g_arrivals may travel now */
        pthread_cond_broadcast(&g_arr_cv);
        while (!g_arr_done[k]) pthread_cond_wait(&g_arr_cv, &g_arr_mu);
        pthread_mutex_unlock(&g_arr_mu);
        if (arr[k].rc != 0) { printf("{\"error\": \"pass of block %d: %s\"}\n", k, fabgpu_strerror(arr[k].rc)); return 1; }
        double v0 = now_ms();
        uint64_t have = 0;
        fabgpu_csp_memo_has_block(g_csp, arr[k].seq, &have); /* preverify.go Validate: HasBlock */
        has_ms[k] = now_ms() - v0;
        if (!have) { printf("{\"error\": \"block %d: no memo entries waiting\"}\n", k); return 1; }
        validate_job job;
        memset(&job, 0, sizeof(job));
        job.blk = copies[k];
        job.tuples = tuples;
        job.tx_first = tx_first;
        job.n_tx = n_tx;
        job.check = k == 0;
        g_job = &job;
        pthread_barrier_wait(&g_go);
        pthread_barrier_wait(&g_done);
        uint64_t ev = 0;
        fabgpu_csp_memo_evict_block(g_csp, arr[k].seq, &ev); /* preverify.go: defer EvictBlock */
        val_ms[k] = now_ms() - v0;
        if (g_profile)
            fprintf(stderr, "block %d: validators %.2f ms wall; thread-ms: copy %.2f hash %.2f verify-memo %.2f in-worker %.2f (of %d threads)\n", k, val_ms[k], job.t_cat, job.t_hash,
                    job.t_memo, job.t_thread, n_thr);
        hits += job.hits; misses += job.misses; hashed += job.hashed_bytes;
        hash_hits += job.hash_hits; hash_bytes += job.hash_bytes; mismatches += job.digest_mismatches;
    }
    double wall = now_ms() - wall0;
    for (int a = 0; a < g_arrivals; a++) pthread_join(arr_th[a], NULL);
    g_quit = 1;
    pthread_barrier_wait(&g_go);
    for (int t = 0; t < n_thr; t++) pthread_join(pool[t], NULL);

    /* protoutil.BlockDataHash (protoutil/blockutils.go:65-68, called by MCS VerifyBlock at internal/peer/gossip/mcs.go:156): ONE serial
     * SHA-256 over the concatenated envelopes = the block's bytes less a few bytes of framing per envelope.  It stays on the CPU
     * (a serial hash has no lanes to spread over); priced here so that the line shows what stands in front of the pass. */
    double bdh[3];
    for (int k = 0; k < 3; k++) {
        unsigned char dg[32];
        double h0 = now_ms();
        SHA256_CTX c;
        SHA256_Init(&c);
        SHA256_Update(&c, copies[k % n_blocks], len);
        SHA256_Final(dg, &c);
        bdh[k] = now_ms() - h0;
    }
    double bdh_med = median(bdh, 3);

    double warm[4096], vals[4096];
    int nw = 0;
    for (int k = 2; k < n_blocks && nw < 4096; k++) warm[nw++] = arr[k].ms;
    int nv = 0;
    for (int k = 1; k < n_blocks && nv < 4096; k++) vals[nv++] = val_ms[k];
    double warm_med = median(warm, nw), val_med = median(vals, nv);
    uint64_t per_dev[64];
    int nd = fabgpu_csp_passes_per_device(g_csp, per_dev, 64);
    uint64_t dw = 0, hw = 0;
    char why[128];
    fabgpu_csp_pass_routes(g_csp, &dw, &hw, why, sizeof(why));
    printf("{\"block_bytes\": %zu, \"n_tx\": %u, \"signatures_per_block\": %u, \"blocks\": %d, \"validator_threads\": %d, \"device_contexts\": %d, "
           "\"provider_new_ms\": %.3f, \"first_block_of_a_fresh_process\": {\"ms\": %.3f, \"etoobig_retries\": %d, \"cap_tx_before\": %u, \"cap_tx_after\": %u, "
           "\"what\": \"the first block: six certificates decoded on the device, their comb tables built there (0.8 ms), the block on the fresh-key kernels; once per process\"}, "
           "\"lone_passes_ms\": [%.3f, %.3f, %.3f, %.3f, %.3f], "
           "\"over_caps_on_a_warm_provider\": {\"ms\": %.3f, \"etoobig_retries\": %d, \"warm_lone_pass_ms\": %.3f, \"over_warm\": %.3f}, "
           "\"pipelined_pass_ms_median\": %.3f, "
           "\"validators_ms_per_block_median\": %.3f, \"block_data_hash_ms\": %.3f, \"has_block_ms\": %.4f, \"arrivals_in_flight\": %d, \"arrival_window\": %d, \"hash_memo\": %d, \"cpu_sha256_MB_per_block\": %.2f, \"hash_memo_MB_per_block\": %.2f, \"hash_memo_hits\": %llu, \"hash_memo_digest_mismatches\": %llu, \"memo_hits\": %llu, \"memo_misses\": %llu, "
           "\"pipeline_wall_ms\": %.3f, \"ms_per_block_end_to_end\": %.3f, \"validated_tx_per_s_end_to_end\": %.1f, "
           "\"passes_on_device_route\": %llu, \"passes_on_host_route\": %llu, \"passes_per_context\": [",
           len, n_tx, n_env_tuples, n_blocks, n_thr, n_dev, t_new, warm_ms[0], warmup[0].retries, cap_tx_start, g_cap_tx, warm_ms[0], warm_ms[1], warm_ms[2], warm_ms[3], warm_ms[4],
           warm_ms[5], warmup[5].retries, warm4, warm4 > 0 ? warm_ms[5] / warm4 : 0, warm_med,
           val_med, bdh_med, has_ms[n_blocks / 2], g_arrivals, g_window, g_hash_memo, (hashed - hash_bytes) / 1e6 / n_blocks, hash_bytes / 1e6 / n_blocks, (unsigned long long)hash_hits,
           (unsigned long long)mismatches, (unsigned long long)hits, (unsigned long long)misses, wall, wall / n_blocks,
           (double)n_tx * n_blocks / (wall * 1e-3), (unsigned long long)dw, (unsigned long long)hw);
    for (int d = 0; d < nd; d++) printf("%s%llu", d ? ", " : "", (unsigned long long)per_dev[d]);
    printf("]}\n");
    fabgpu_csp_free(g_csp);
    return 0;
}
