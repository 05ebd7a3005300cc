// C-ABI shim (include/fabgpu.h) over the HIP kernels: context, staging, launches.
// One context = one GPU + one HIP stream; host entry points stage through pinned buffers the context
// owns (nothing of the caller's memory is retained after return - cgo pointer rules).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <sys/mman.h>
#include <set>
#include <mutex>
#include <new>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fabgpu.h"
#include "kernels.h"
#include "block_walk_dev.h"
#include "worker_pool.h"
#include "bn_tables29.h"
#include "p256_tables29.h"

using namespace fab;

namespace {

struct Buf {  // growable pinned-host + device pair
    void* h = nullptr;
    void* d = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return FABGPU_OK;
        release();
        size_t want = bytes + bytes / 4 + 256;
        if (hipHostMalloc(&h, want, hipHostMallocDefault) != hipSuccess) { h = nullptr; return FABGPU_ENOMEM; }
        if (hipMalloc(&d, want) != hipSuccess) { hipHostFree(h); h = nullptr; d = nullptr; return FABGPU_ENOMEM; }
        cap = want;
        return FABGPU_OK;
    }
    void release() {
        if (h) hipHostFree(h);
        if (d) hipFree(d);
        h = d = nullptr;
        cap = 0;
    }
};

struct DevBuf {  // growable device-only buffer
    void* d = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return FABGPU_OK;
        release();
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&d, want) != hipSuccess) { d = nullptr; return FABGPU_ENOMEM; }
        cap = want;
        return FABGPU_OK;
    }
    void release() {
        if (d) hipFree(d);
        d = nullptr;
        cap = 0;
    }
};
// Pinned host memory on 2 MiB pages, for tables and block copies that the HOST reads at random (the verdict / digest memo: every
// validator thread, every signature): transparent huge pages where the kernel grants them on request (THP "madvise" or "always"),
// registered with the runtime so that DMA treats it like hipHostMalloc'd memory.  A memo lookup touches a handful of places in a few
// tens of MB that DMA wrote - cold in every cache AND, on 4 KiB pages, in every TLB: each touch then pays for a page walk whose
// entries are cold as well.  nullptr: not available here (the caller falls back to hipHostMalloc).
static void* pinned_huge_alloc(size_t bytes) {
    constexpr size_t huge = (size_t)2 << 20;
    const size_t want = (bytes + huge - 1) & ~(huge - 1);
    void* p = nullptr;
    if (posix_memalign(&p, huge, want) != 0 || !p) return nullptr;
    (void)madvise(p, want, MADV_HUGEPAGE);
    for (size_t o = 0; o < want; o += 4096) ((volatile uint8_t*)p)[o] = 0;      // fault the pages in now (huge ones where granted), not inside a pass
    if (hipHostRegister(p, want, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        free(p);
        return nullptr;
    }
    return p;
}
static void pinned_huge_free(void* p) {
    if (!p) return;
    (void)hipHostUnregister(p);
    free(p);
}

struct PinBuf {  // growable pinned host buffer
    void* h = nullptr;
    size_t cap = 0;
    unsigned flags = hipHostMallocDefault;
    bool want_huge = false;   // try 2 MiB pages first (pinned_huge_alloc)
    bool is_huge = false;
    int ensure(size_t bytes) {
        if (bytes <= cap) return FABGPU_OK;
        release();
        size_t want = bytes + bytes / 4 + 256;
        if (want_huge && (h = pinned_huge_alloc(want)) != nullptr) {
            is_huge = true;
            cap = want;
            return FABGPU_OK;
        }
        if (hipHostMalloc(&h, want, flags) != hipSuccess) { h = nullptr; return FABGPU_ENOMEM; }
        cap = want;
        return FABGPU_OK;
    }
    void release() {
        if (h && is_huge) pinned_huge_free(h);
        else if (h) hipHostFree(h);
        h = nullptr;
        cap = 0;
        is_huge = false;
    }
};

}  // namespace

struct fabgpu_ctx {
    int device = 0;
    bool allow_pair = true;   // !FABGPU_FLAG_ONE_LANE_ONLY
    bool allow_quad = true;   // !FABGPU_FLAG_NO_QUAD (idemix: four lanes per signature for batches <= IDEMIX_QUAD_MAX)
    bool nym_side_stream = true; // !FABGPU_FLAG_NYM_NO_SIDE_STREAM (idemix four-lane form: the fixed-base terms on a second stream beside the commitments)
    std::atomic<bool> test_nym_side_after{false};   // TEST HOOK (fab::ctx_test_nym_side_after; read once per nym launch, handed to the launcher as a parameter)
    bool nym_two_phase = true;   // !FABGPU_FLAG_NYM_FUSED_HASH (idemix four-lane form: commitments, then challenges with eight lanes on a message)
    bool allow_wide = true;   // !FABGPU_FLAG_NO_WIDE (registered keys: eight lanes per signature in two phases for launches <= WIDE_LAUNCH_MAX)
    int pair_table_lds = -1;       // the verify-only pair kernel's per-signature table: 1 in LDS, 0 in the global workspace, -1 by batch size (kernels.h)
    hipStream_t stream = nullptr;
    // FABGPU_FAULT_INJECT (tests of the failure contract only): "launch" makes every kernel submission report hipErrorLaunchFailure,
    // "oom" makes every workspace / staging allocation fail.  A non-zero return must then reach the caller and no verdict may be written.
    int fault = 0;
    int32_t* d_gtab = nullptr;
    std::mutex mu;
    Buf fields;   // qx|qy|e|r|s
    Buf arena;    // message bytes
    Buf offs;     // u32 offsets
    Buf out;      // verdict words | status bytes | digests
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool time_kernels = false;   // FABGPU_FLAG_TIME_KERNELS: bracket every launch with two timing events (fabgpu_last_kernel_ms); off by
                                 // default - two more packets per launch between back-to-back kernels are not free
    // Workspaces for the verify kernels' per-lane j*Q tables.  A launch borrows one: `reserved` from acquire until release has
    // recorded an event behind the kernel, `armed` from then until that event completes - two separate flags, because between
    // acquire and release the event is stale (or was never recorded) and hipEventQuery on it would answer "done".
    // acquire hands the POINTER out under the lock: the vector may grow while the caller launches.
    struct QWs {
        void* p = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        bool reserved = false;   // handed to a launch that has not recorded `done` yet
        bool armed = false;      // `done` was recorded behind the launch that used it
        NymSide nym_side;             // idemix four-lane form: the side stream of the fixed-base launch and its fork / join events (made on first use)
        hipStream_t last = nullptr;   // the stream of that launch: a later launch on the SAME stream runs behind it anyway and may
                                      // take the workspace without waiting (20 queued steps of a bench loop used to mean 20 workspaces
                                      // of 61 MB each, allocated inside the timed region)
    };
    std::vector<QWs> qws;
    // Registered public keys: one 640 KiB comb table each, resident on the device; d_ktabs mirrors the pointer array.
    std::mutex kmu;
    std::vector<int32_t*> ktabs;
    std::map<std::string, uint32_t> key_ids;   // qx||qy -> id
    const int32_t** d_ktabs = nullptr;      // KTAB_STRIDE pointers per key: [2 k] the 8-bit comb, [2 k + 1] the 16-bit comb or nullptr
    size_t d_ktabs_cap = 0;                 // (in keys)
    std::vector<void*> retired;   // outgrown d_ktabs arrays, freed at shutdown
    // Round 6, FABGPU_FLAG_KEY_TABLES_16BIT: a registered key also gets a 16-bit comb (CombTab<16>, 80 MiB - the generator's own format),
    // built on stream_keytab BEHIND the registration (nobody waits for it; its pointer reaches d_ktabs[2 k + 1] by a copy queued after
    // the build, so a launch sees nullptr or a finished table).  Up to KTAB16_MAX keys: 5 GiB of a 288 GB device.  All under kmu.
    bool key_tables_16 = false;
    hipStream_t stream_keytab16 = nullptr;  // (its own: a registration waits on stream_keytab for the 8-bit table and must not find an 80 MiB build queued there)
    static constexpr size_t KTAB16_MAX = 64;
    std::vector<void*> ktab16;              // per key id: the table, or nullptr
    std::vector<void*> ktab16_slabs;        // tables come from slabs of KTAB16_SLAB (the first one at fabgpu_init: a registration must not pay for an 80 MiB hipMalloc)
    static constexpr size_t KTAB16_SLAB = 8;
    size_t ktab16_carved = 0;               // tables handed out of the slabs so far
    void* ktab16_rooms = nullptr;           // KTAB16_MAX builders' rooms (key + pointer + scratch), one allocation
    uint8_t* ktab16_heads = nullptr;        // pinned host memory, KTAB16_MAX x 128 bytes: what each build's upload reads (never reused: the copies are asynchronous)
    size_t ktab16_room_bytes = 0;
    size_t ktab16_count = 0;
    // Where the tables live: slabs of KTAB_SLAB tables (a table is never freed before shutdown, and hipMalloc / hipFree per table were
    // most of what a registration cost once the tables were built on the device: 4.7 ms of runtime calls around 0.8 ms of kernels for a
    // channel's six signers).  ktab_free: tables whose installation failed.  All under kmu.
    static constexpr size_t KTAB_SLAB = 32;
    std::vector<void*> ktab_slabs;
    size_t ktab_slab_used = KTAB_SLAB;
    std::vector<int32_t*> ktab_free;
    // the device builder's room (keytab_kernels.hip): keys + table pointers in, the chain's and the affine bases' scratch; its own stream
    std::mutex ktab_build_mu;
    void* d_ktab_in = nullptr;
    void* d_ktab_scr = nullptr;
    size_t ktab_in_cap = 0, ktab_scr_cap = 0;
    hipStream_t stream_keytab = nullptr;
    // Registered idemix issuers: a fixed-capacity device array of slots (idemix_kernels.hip IssuerDev); a slot is written
    // before n_issuers is raised, so launches in flight never read a half-written one.
    std::mutex imu;
    void* d_issuers = nullptr;
    std::vector<int32_t*> itabs;                  // two comb tables per issuer
    std::map<std::string, uint32_t> issuer_ids;   // hsk || hrand || hash -> id
    Buf nym;          // staging of the nym host-pointer entry point: issuer_id | six fields (| spans, when they ride in an identity batch)
    Buf nymout;       // results of the pseudonym signatures that ride in an identity batch: verdict words | status bytes
    hipStream_t stream2 = nullptr;   // the pseudonym signatures of an identity batch run here, next to the ECDSA kernels on `stream`
    hipEvent_t ev_up = nullptr;      // "the arena is on the device" (recorded on stream, awaited by stream2)
    // the device block pass runs on four streams: walk / gates / endorsements on `stream`, the creators' hashes and launch on
    // stream2, the mid-states on stream3, the TxID / proposal-hash digests on stream4
    hipStream_t stream3 = nullptr, stream4 = nullptr;
    hipEvent_t ev_w[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool pred_has_nym = false;       // the previous block had idemix creators: queue the nym launch without waiting for the gates
    uint32_t pred_nym_rows = 0;      // ... and how many: the launch runs over that many packed rows plus a margin
    Buf gath;         // gathered hashes of an identity batch: spans | running offsets | digests
    void* d_gscr = nullptr;   // device scratch the gather kernel stitches the messages into
    size_t gscr_cap = 0;
    // fabgpu_arena_stage: arenas uploaded ahead of the batches that refer to them.  A few slots, so that the channels of a peer
    // validating at once do not throw each other's block out between "staged" and "submitted" (with one slot, two callers cost
    // more per block than one: the loser uploaded its 50 MB a second time inside its device call).  A slot's mutex is held while
    // it is being filled and while a batch reads it; smu only guards the choice of a slot.  Lock order: smu, slot, mu.
    struct Staged {
        std::mutex m;
        void* d = nullptr;
        size_t cap = 0, len = 0;
        std::atomic<uint64_t> token{0};
    };
    static constexpr int N_STAGED = 3;
    std::mutex smu;
    Staged staged_slots[N_STAGED];
    uint64_t stage_seq = 0;
    Buf keyed;        // staging of the keyed host-pointer entry point: key_id | e | r | s
    Buf pre;          // staging of prefixed batches: pre_off | pre_idx | mid-states
    Buf tailbuf;      // staging of an identity batch's tail when the arena itself bypasses the pinned buffer
    // the block pass on the device (block_walk_dev.h): per-envelope arrays, per-tuple arrays, pinned staging for what travels, and
    // the table of identities the provider has met (slots | entries | bytes in one allocation, swapped whole under mu)
    DevBuf walk_env, walk_tup, idtab_buf;
    PinBuf stage_pin;                // fabgpu_arena_stage: pinned staging of arenas that arrive in pageable memory
    std::mutex stage_pin_mu;
    // arena_stage_keep: staging buffers that STAY with the caller after the upload (block_walk_dev.h HostCopy - the host copy of a block
    // the provider's digest memo compares bccsp.Hash callers' bytes with, until the block's validation has returned).  A small pool:
    // buffers are made on demand up to keep_max, handed back by host_copy_release, and reused (pinning 64 MiB costs milliseconds).
    struct KeepBuf {
        PinBuf pin;
        bool in_use = false;
        KeepBuf() {
            pin.flags = hipHostMallocPortable;
            pin.want_huge = true;                 // read by the validators' threads, a few KB at 40 000 random places per block
        }
    };
    std::mutex keep_mu;
    std::vector<std::unique_ptr<KeepBuf>> keep_pool;
    uint32_t keep_max = 8;
    uint64_t keep_refused = 0;
    // The copiers of this context's uploads: threads of its own (made with the first big upload), so that the uploads of a provider's G
    // devices - each behind its own PCIe link - run side by side instead of taking turns on the process-wide pool of the host passes.
    std::unique_ptr<WorkerPool> stage_pool;
    hipStream_t stream_copy = nullptr;
    hipStream_t stream_copy_more[3] = {nullptr, nullptr, nullptr};   // further upload queues: pieces alternate between DMA engines
    PinBuf walk_pin;
    // what the pass's kernels write for the HOST (block_walk_dev.h WalkHostOut): pinned, mapped, coherent - the host polls flags in it
    PinBuf walk_map;
    uint32_t walk_seq = 0;
    void* d_idtab = nullptr;
    uint32_t idtab_n = 0, idtab_mask = 0;
    uint64_t idtab_seed = 0;
    // Which verify kernels the pass queues BEFORE it knows what the gates found (one host round trip less): the keyed ones for a launch
    // class (creators / everybody else) whose tuples all had comb tables in the previous pass, the fresh-key ones - always correct, the
    // key travels in the row - otherwise.  A wrong "keyed" guess is repaired by launching that class again (walk_block_pass).
    bool pred_keyed_creators = false, pred_keyed_others = false;
    size_t idtab_entries_off = 0, idtab_bytes_off = 0;
    std::mutex qmu;   // guards qws only (the host-pointer entry points call the _dev ones while holding mu)
    int acquire_qws(size_t bytes, size_t* idx, void** p, hipStream_t st);
    void release_qws(size_t idx, hipStream_t st) {
        std::lock_guard<std::mutex> lk(qmu);
        QWs& w = qws[idx];
        if (hipEventRecord(w.done, st) == hipSuccess) {
            w.armed = true;
            w.reserved = false;
            w.last = st;
        } else if (hipStreamSynchronize(st) == hipSuccess) {
            // no event to watch: wait the launch out, then the workspace is simply free again (a flapping device must not eat one per failure)
            w.armed = false;
            w.reserved = false;
        }
        // (neither worked: the workspace stays reserved - never shared with a launch that may still be reading it)
    }
};

int fabgpu_ctx::acquire_qws(size_t bytes, size_t* idx, void** p, hipStream_t st) {
    if (fault == 2) return FABGPU_ENOMEM;
    std::lock_guard<std::mutex> lk(qmu);
    for (size_t i = 0; i < qws.size(); i++) {
        QWs& w = qws[i];
        if (w.reserved) continue;                                          // somebody is between acquire and release
        const bool same_stream = w.armed && w.last == st;                  // stream order serialises the two launches
        if (w.armed && !same_stream && hipEventQuery(w.done) != hipSuccess) continue;      // still in flight on another stream
        if (w.bytes < bytes) {
            if (same_stream && hipEventQuery(w.done) != hipSuccess) continue;              // cannot free what a queued launch still reads
            if (w.p) hipFree(w.p);
            w.p = nullptr;
            w.bytes = 0;
            if (hipMalloc(&w.p, bytes) != hipSuccess) return FABGPU_ENOMEM;
            w.bytes = bytes;
        }
        w.armed = false;
        w.reserved = true;
        *idx = i;
        *p = w.p;
        return FABGPU_OK;
    }
    QWs w;
    if (hipEventCreateWithFlags(&w.done, hipEventDisableTiming) != hipSuccess) return FABGPU_ENODEV;
    if (hipMalloc(&w.p, bytes) != hipSuccess) {
        hipEventDestroy(w.done);
        return FABGPU_ENOMEM;
    }
    w.bytes = bytes;
    w.reserved = true;
    qws.push_back(w);
    *idx = qws.size() - 1;
    *p = w.p;
    return FABGPU_OK;
}

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        hipGetDevice(&prev);
        if (prev != dev) hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) hipSetDevice(prev);
    }
};

int hip_to_rc(hipError_t e) { return e == hipSuccess ? FABGPU_OK : (e == hipErrorOutOfMemory ? FABGPU_ENOMEM : FABGPU_ELAUNCH); }
// the launch result an entry point reports: the real one, or the injected failure
inline hipError_t launched(const fabgpu_ctx* ctx, hipError_t e) { return ctx->fault == 1 ? hipErrorLaunchFailure : e; }

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" {

int fabgpu_abi_version(void) { return FABGPU_ABI_VERSION; }

const char* fabgpu_strerror(int code) {
    switch (code) {
        case FABGPU_OK: return "ok";
        case FABGPU_EINVAL: return "invalid argument";
        case FABGPU_ENODEV: return "no usable gfx950 device (HIP runtime/device unavailable); fall back to bccsp/sw";
        case FABGPU_ENOMEM: return "host or device allocation failed";
        case FABGPU_ELAUNCH: return "HIP launch/copy/execution failure";
        case FABGPU_ETOOBIG: return "batch or arena exceeds 32-bit offsets";
        default: return "unknown fabgpu error";
    }
}

int fabgpu_device_count(fabgpu_ctx*) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return FABGPU_ENODEV;
    return n;
}

int fabgpu_init(const fabgpu_cfg* cfg, fabgpu_ctx** out) {
    if (!out) return FABGPU_EINVAL;
    *out = nullptr;
    if (cfg && (cfg->flags & ~(uint32_t)(FABGPU_FLAG_ONE_LANE_ONLY | FABGPU_FLAG_TIME_KERNELS | FABGPU_FLAG_NO_QUAD | FABGPU_FLAG_PAIR_TABLE_LDS |
                                          FABGPU_FLAG_PAIR_TABLE_GLOBAL | FABGPU_FLAG_NO_WIDE | FABGPU_FLAG_NYM_FUSED_HASH | FABGPU_FLAG_NYM_NO_SIDE_STREAM |
                                          FABGPU_FLAG_KEY_TABLES_16BIT)) != 0) return FABGPU_EINVAL;
    if (cfg && (cfg->flags & FABGPU_FLAG_PAIR_TABLE_LDS) && (cfg->flags & FABGPU_FLAG_PAIR_TABLE_GLOBAL)) return FABGPU_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FABGPU_ENODEV;
    int dev = cfg ? cfg->device : -1;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return FABGPU_ENODEV;
    }
    if (dev >= ndev) return FABGPU_EINVAL;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return FABGPU_ENODEV;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FABGPU_ENODEV;  // the code object is gfx950-only
    fabgpu_ctx* ctx = new (std::nothrow) fabgpu_ctx();
    if (!ctx) return FABGPU_ENOMEM;
    ctx->device = dev;
    ctx->walk_map.flags = hipHostMallocMapped | hipHostMallocCoherent;
    ctx->allow_pair = !(cfg && (cfg->flags & FABGPU_FLAG_ONE_LANE_ONLY));
    ctx->allow_quad = !(cfg && (cfg->flags & FABGPU_FLAG_NO_QUAD));
    ctx->nym_two_phase = !(cfg && (cfg->flags & FABGPU_FLAG_NYM_FUSED_HASH));
    ctx->nym_side_stream = !(cfg && (cfg->flags & FABGPU_FLAG_NYM_NO_SIDE_STREAM));
    ctx->key_tables_16 = cfg && (cfg->flags & FABGPU_FLAG_KEY_TABLES_16BIT);
    // (the wide form is built from the two-lane form's reasons: a context that may not use two lanes per signature does not use eight)
    ctx->allow_wide = ctx->allow_pair && !(cfg && (cfg->flags & FABGPU_FLAG_NO_WIDE));
    ctx->pair_table_lds = cfg && (cfg->flags & FABGPU_FLAG_PAIR_TABLE_LDS) ? 1 : (cfg && (cfg->flags & FABGPU_FLAG_PAIR_TABLE_GLOBAL) ? 0 : pair_table_default());
    // The LDS form of the pair kernel asks for 130 KiB of dynamic LDS per workgroup (p256_pair29.h PAIR_LDS_CELLS_PER_SIG): a device that
    // grants less keeps the table in the global workspace instead of failing every large launch.  (That allocation is also
    // what keeps other kernels off the launch's CUs - more than the 84 KiB the pass reserves for that purpose.)
    if (std::max<size_t>(prop.sharedMemPerBlock, prop.maxSharedMemoryPerMultiProcessor) < pair_table_lds_bytes()) ctx->pair_table_lds = 0;
    ctx->time_kernels = cfg && (cfg->flags & FABGPU_FLAG_TIME_KERNELS);
    DeviceGuard g(dev);
    int rc = FABGPU_OK;
    do {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { rc = FABGPU_ENODEV; break; }
        if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) { rc = FABGPU_ENODEV; break; }
        if (hipEventCreateWithFlags(&ctx->ev_up, hipEventDisableTiming) != hipSuccess) { rc = FABGPU_ENODEV; break; }
        if (hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess) { rc = FABGPU_ENODEV; break; }
        if (hipStreamCreateWithFlags(&ctx->stream4, hipStreamNonBlocking) != hipSuccess) { rc = FABGPU_ENODEV; break; }
        if (hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking) != hipSuccess) { rc = FABGPU_ENODEV; break; }
        {
            bool ok = true;
            for (auto& sc : ctx->stream_copy_more) ok = ok && hipStreamCreateWithFlags(&sc, hipStreamNonBlocking) == hipSuccess;
            if (!ok) { rc = FABGPU_ENODEV; break; }
        }
        {
            bool ok = true;
            for (auto& e : ctx->ev_w) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
            if (!ok) { rc = FABGPU_ENODEV; break; }
        }
        if (const char* fi = getenv("FABGPU_FAULT_INJECT")) ctx->fault = !strcmp(fi, "launch") ? 1 : (!strcmp(fi, "oom") ? 2 : 0);
        if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { rc = FABGPU_ENODEV; break; }
        if (hipMalloc((void**)&ctx->d_gtab, sizeof(int32_t) * GTab16::TABLE_WORDS) != hipSuccess) { rc = FABGPU_ENOMEM; break; }
        // The generator's comb (80 MiB) is built on the device since round 6 (keytab_kernels.hip launch_gtab_build: one million lanes, a
        // few milliseconds) - byte-identical to p256_tables29.h's, which took 0.2 s on sixteen host threads plus the upload and which is
        // what a failed device build falls back to.  FABGPU_GTAB_HOST=1 (tests, A/B) takes the host builder.
        bool built = false;
        if (!getenv("FABGPU_GTAB_HOST") && ctx->fault == 0) {
            void* scr = nullptr;
            if (hipMalloc(&scr, gtab_scratch_bytes()) == hipSuccess) {
                built = launch_gtab_build(ctx->d_gtab, scr, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
                hipFree(scr);
            }
            if (!built) (void)hipGetLastError();
        }
        if (!built) {
            std::vector<int32_t> tab(GTab16::TABLE_WORDS);   // 80 MiB, ~0.2 s on 16 host threads
            build_g_comb_table16(tab.data());
            if (hipMemcpy(ctx->d_gtab, tab.data(), sizeof(int32_t) * GTab16::TABLE_WORDS, hipMemcpyHostToDevice) != hipSuccess) { rc = FABGPU_ELAUNCH; break; }
        }
        if (ctx->key_tables_16) {
            // FABGPU_FLAG_KEY_TABLES_16BIT: the first slab of 16-bit key tables (8 x 80 MiB), the builders' rooms and their stream now, so that a
            // registration only queues launches (best effort: without them the keys are served by their 8-bit combs)
            void* slab = nullptr;
            if (hipMalloc(&slab, sizeof(int32_t) * GTab16::TABLE_WORDS * fabgpu_ctx::KTAB16_SLAB) == hipSuccess) ctx->ktab16_slabs.push_back(slab);
            else (void)hipGetLastError();
            ctx->ktab16_room_bytes = (keytab16_scratch_bytes() + 128 + 255) & ~(size_t)255;
            if (hipMalloc(&ctx->ktab16_rooms, ctx->ktab16_room_bytes * fabgpu_ctx::KTAB16_MAX) != hipSuccess) {
                (void)hipGetLastError();
                ctx->ktab16_rooms = nullptr;
            }
            if (hipStreamCreateWithFlags(&ctx->stream_keytab16, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                ctx->stream_keytab16 = nullptr;
            }
        }
        if (cfg && cfg->max_batch) {
            size_t n = cfg->max_batch;
            if ((rc = ctx->fields.ensure(n * 160)) || (rc = ctx->offs.ensure((n + 1) * 4)) || (rc = ctx->out.ensure(n * 41 + 64))) break;
        }
        if (cfg && cfg->max_arena) {
            if ((rc = ctx->arena.ensure((size_t)cfg->max_arena + 128))) break;
        }
    } while (0);
    if (rc != FABGPU_OK) {
        fabgpu_shutdown(ctx);
        return rc;
    }
    *out = ctx;
    return FABGPU_OK;
}

void fabgpu_shutdown(fabgpu_ctx* ctx) {
    if (!ctx) return;
    ctx->stage_pool.reset();
    {
        DeviceGuard g(ctx->device);
        if (ctx->stream) hipStreamSynchronize(ctx->stream);
        ctx->fields.release();
        ctx->arena.release();
        ctx->offs.release();
        ctx->out.release();
        if (ctx->d_gtab) hipFree(ctx->d_gtab);
        for (auto* t : ctx->ktab_slabs) hipFree(t);              // (every key table lies in a slab)
        for (auto* t : ctx->retired) hipFree(t);
        if (ctx->d_ktab_in) hipFree(ctx->d_ktab_in);
        if (ctx->d_ktab_scr) hipFree(ctx->d_ktab_scr);
        if (ctx->stream_keytab) hipStreamDestroy(ctx->stream_keytab);
        for (auto* t : ctx->itabs) hipFree(t);
        if (ctx->d_issuers) hipFree(ctx->d_issuers);
        ctx->nym.release();
        ctx->nymout.release();
        if (ctx->stream2) hipStreamDestroy(ctx->stream2);
        if (ctx->ev_up) hipEventDestroy(ctx->ev_up);
        if (ctx->stream3) hipStreamDestroy(ctx->stream3);
        if (ctx->stream4) hipStreamDestroy(ctx->stream4);
        if (ctx->stream_copy) hipStreamDestroy(ctx->stream_copy);
        for (auto sc : ctx->stream_copy_more)
            if (sc) hipStreamDestroy(sc);
        ctx->stage_pin.release();
        for (auto& k : ctx->keep_pool) k->pin.release();
        ctx->keep_pool.clear();
        for (auto& e : ctx->ev_w)
            if (e) hipEventDestroy(e);
        ctx->gath.release();
        ctx->tailbuf.release();
        ctx->keyed.release();
        ctx->pre.release();
        if (ctx->d_gscr) hipFree(ctx->d_gscr);
        ctx->walk_env.release();
        ctx->walk_tup.release();
        ctx->walk_pin.release();
        ctx->walk_map.release();
        ctx->idtab_buf.release();
        for (auto& sl : ctx->staged_slots)
            if (sl.d) hipFree(sl.d);
        if (ctx->stream_keytab16) { hipStreamSynchronize(ctx->stream_keytab16); hipStreamDestroy(ctx->stream_keytab16); }       // (16-bit key tables may still be building)
        if (ctx->d_ktabs) hipFree((void*)ctx->d_ktabs);
        for (void* t : ctx->ktab16_slabs)
            if (t) hipFree(t);
        if (ctx->ktab16_rooms) hipFree(ctx->ktab16_rooms);
        if (ctx->ktab16_heads) hipHostFree(ctx->ktab16_heads);
        for (auto& w : ctx->qws) {
            if (w.p) hipFree(w.p);
            if (w.done) hipEventDestroy(w.done);
            if (w.nym_side.stream) { hipStreamSynchronize(w.nym_side.stream); hipStreamDestroy(w.nym_side.stream); }
            if (w.nym_side.fork) hipEventDestroy(w.nym_side.fork);
            if (w.nym_side.join) hipEventDestroy(w.nym_side.join);
        }
        if (ctx->ev0) hipEventDestroy(ctx->ev0);
        if (ctx->ev1) hipEventDestroy(ctx->ev1);
        if (ctx->stream) hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

}  // extern "C"
void fab::ctx_test_nym_side_after(fabgpu_ctx* ctx, bool on) {
    if (ctx) ctx->test_nym_side_after.store(on);
}
// what the test-hook library's fabgpu_last_kernel_ms reads (FABGPU_FLAG_TIME_KERNELS contexts; block_walk_dev.h)
float fab::ctx_last_kernel_ms(fabgpu_ctx* ctx) {
    if (!ctx || !ctx->timed) return -1.0f;
    DeviceGuard g(ctx->device);
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
    return ms;
}
extern "C" {

// ---- device-resident entry points ----------------------------------------------------------------
int fabgpu_p256_verify_batch_dev(fabgpu_ctx* ctx, size_t n, const void* qx, const void* qy, const void* e, const void* r,
                                 const void* s, void* verdict_bits, void* status, void* stream) {
    if (!ctx || (n && (!qx || !qy || !e || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0xFFFFFFF0ull) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    size_t wi = 0;
    void* wsp = nullptr;
    int rc = ctx->acquire_qws(verify_workspace_bytes((uint32_t)n, ctx->allow_pair), &wi, &wsp, st);
    if (rc != FABGPU_OK) return rc;
    if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
    hipError_t err = launch_p256_verify((uint32_t)n, qx, qy, e, r, s, ctx->d_gtab, wsp, verdict_bits, status, ctx->allow_pair, st, 0, ctx->pair_table_lds);
    if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
    ctx->release_qws(wi, st);
    ctx->timed = ctx->time_kernels;
    return hip_to_rc(launched(ctx, err));
}

int fabgpu_sha256_batch_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off, void* digests,
                            void* stream) {
    if (!ctx || (n && (!arena || !off || !digests))) return FABGPU_EINVAL;
    if (n > 0xFFFFFFF0ull || arena_bytes > 0xFFFFFFFFull) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
    hipError_t err = launch_sha256_batch((uint32_t)n, arena, arena_bytes, off, digests, st);
    if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
    ctx->timed = ctx->time_kernels;
    return hip_to_rc(launched(ctx, err));
}

int fabgpu_sha256_p256_verify_batch_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off,
                                        const void* qx, const void* qy, const void* r, const void* s, void* verdict_bits,
                                        void* status, void* stream) {
    if (!ctx || (n && (!arena || !off || !qx || !qy || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0xFFFFFFF0ull || arena_bytes > 0xFFFFFFFFull) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    size_t wi = 0;
    void* wsp = nullptr;
    int rc = ctx->acquire_qws(verify_workspace_bytes((uint32_t)n, ctx->allow_pair), &wi, &wsp, st);
    if (rc != FABGPU_OK) return rc;
    if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
    hipError_t err = launch_sha256_p256_verify((uint32_t)n, arena, arena_bytes, off, qx, qy, r, s, ctx->d_gtab, wsp, verdict_bits, status, ctx->allow_pair, ShaPrefixArgs(), st);
    if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
    ctx->release_qws(wi, st);
    ctx->timed = ctx->time_kernels;
    return hip_to_rc(launched(ctx, err));
}

// ---- idemix pseudonym signatures --------------------------------------------------------------------
int fabgpu_bn256_g1_on_curve(const uint8_t* x32, const uint8_t* y32) {
    if (!x32 || !y32) return 0;
    const u256 P = FAB_BN_P;
    u256 x, y;
    from_be32(x, x32);
    from_be32(y, y32);
    if (!lt256(x, P) || !lt256(y, P)) return 0;
    fbn fx, fy;
    fe_to_mont(fx, x);
    fe_to_mont(fy, y);
    return bn_on_curve29(fx, fy) ? 1 : 0;
}

int fabgpu_idemix_issuer_register(fabgpu_ctx* ctx, const uint8_t* hsk_x32, const uint8_t* hsk_y32, const uint8_t* hrand_x32,
                                  const uint8_t* hrand_y32, const uint8_t* ipk_hash32, uint32_t* issuer_id) {
    if (!ctx || !hsk_x32 || !hsk_y32 || !hrand_x32 || !hrand_y32 || !ipk_hash32 || !issuer_id) return FABGPU_EINVAL;
    if (!fabgpu_bn256_g1_on_curve(hsk_x32, hsk_y32) || !fabgpu_bn256_g1_on_curve(hrand_x32, hrand_y32)) return FABGPU_EINVAL;
    std::string k((const char*)hsk_x32, 32);
    k.append((const char*)hsk_y32, 32);
    k.append((const char*)hrand_x32, 32);
    k.append((const char*)hrand_y32, 32);
    k.append((const char*)ipk_hash32, 32);
    std::lock_guard<std::mutex> lk(ctx->imu);
    auto it = ctx->issuer_ids.find(k);
    if (it != ctx->issuer_ids.end()) {
        *issuer_id = it->second;
        return FABGPU_OK;
    }
    const size_t cur = ctx->itabs.size() / 2;
    if (cur >= FABGPU_MAX_ISSUERS) return FABGPU_ENOMEM;
    DeviceGuard g(ctx->device);
    const size_t slot = idemix_issuer_dev_bytes();
    if (!ctx->d_issuers && hipMalloc(&ctx->d_issuers, slot * FABGPU_MAX_ISSUERS) != hipSuccess) {
        ctx->d_issuers = nullptr;
        return FABGPU_ENOMEM;
    }
    const size_t tb = sizeof(int32_t) * KeyTab8::TABLE_WORDS;
    std::vector<int32_t> tab(KeyTab8::TABLE_WORDS);
    int32_t* d[2] = {nullptr, nullptr};
    const uint8_t* bx[2] = {hsk_x32, hrand_x32};
    const uint8_t* by[2] = {hsk_y32, hrand_y32};
    for (int b = 0; b < 2; b++) {
        u256 x, y;
        from_be32(x, bx[b]);
        from_be32(y, by[b]);
        build_bn_comb_table<8>(tab.data(), x, y);
        if (hipMalloc((void**)&d[b], tb) != hipSuccess || hipMemcpy(d[b], tab.data(), tb, hipMemcpyHostToDevice) != hipSuccess) {
            if (d[0]) hipFree(d[0]);
            if (d[1]) hipFree(d[1]);
            return FABGPU_ENOMEM;
        }
    }
    std::vector<uint8_t> hs(slot);
    idemix_issuer_dev_fill(hs.data(), d[0], d[1], ipk_hash32);
    if (hipMemcpy((uint8_t*)ctx->d_issuers + cur * slot, hs.data(), slot, hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(d[0]);
        hipFree(d[1]);
        return FABGPU_ELAUNCH;
    }
    ctx->itabs.push_back(d[0]);
    ctx->itabs.push_back(d[1]);
    *issuer_id = (uint32_t)cur;
    ctx->issuer_ids[k] = *issuer_id;
    return FABGPU_OK;
}

int fabgpu_idemix_issuer_count(fabgpu_ctx* ctx) {
    if (!ctx) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->imu);
    return (int)(ctx->itabs.size() / 2);
}

// spans: off holds n (start, end) pairs instead of n + 1 running offsets; timed: this launch is what fabgpu_last_kernel_ms reports
static int nym_verify_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off, bool spans, const void* issuer_id,
                          const void* nym_x, const void* nym_y, const void* proof_c, const void* proof_s_sk, const void* proof_s_r_nym,
                          const void* nonce, void* verdict_bits, void* status, void* stream, bool timed, const void* gather = nullptr, uint32_t lds_reserve = 0) {
    if (!ctx || (n && (!arena || !off || !nym_x || !nym_y || !proof_c || !proof_s_sk || !proof_s_r_nym || !nonce || !verdict_bits)))
        return FABGPU_EINVAL;
    if (n > 0xFFFFFFF0ull || arena_bytes > 0xFFFFFFFFull) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    uint32_t n_issuers;
    const void* issuers;
    {
        std::lock_guard<std::mutex> lk(ctx->imu);
        n_issuers = (uint32_t)(ctx->itabs.size() / 2);
        issuers = ctx->d_issuers;
    }
    if (n_issuers == 0) return FABGPU_EINVAL;   // nothing registered: every tuple would need bccsp/sw
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    size_t wi = 0;
    void* wsp = nullptr;
    int rc = ctx->acquire_qws(idemix_workspace_bytes((uint32_t)n, ctx->allow_pair, ctx->allow_quad), &wi, &wsp, st);
    if (rc != FABGPU_OK) return rc;
    // the side stream of this workspace slot (the slot is ours until release_qws: nobody else touches its entry)
    NymSide side;
    if (ctx->nym_two_phase && ctx->nym_side_stream && n <= (size_t)NYM_SIDE_STREAM_MAX) {
        std::lock_guard<std::mutex> lk(ctx->qmu);
        NymSide& sd = ctx->qws[wi].nym_side;
        if (!sd.stream) {
            NymSide made;
            if (hipStreamCreateWithFlags(&made.stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&made.fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&made.join, hipEventDisableTiming) == hipSuccess)
                sd = made;
            else {
                if (made.fork) hipEventDestroy(made.fork);
                if (made.stream) hipStreamDestroy(made.stream);
            }
        }
        side = sd;
        side.test_side_after = ctx->test_nym_side_after;
    }
    timed = timed && ctx->time_kernels;
    if (timed) hipEventRecord(ctx->ev0, st);
    hipError_t err = launch_idemix_nym_verify((uint32_t)n, arena, arena_bytes, off, issuer_id, issuers, n_issuers, nym_x, nym_y, proof_c,
                                              proof_s_sk, proof_s_r_nym, nonce, wsp, verdict_bits, status, ctx->allow_pair, ctx->allow_quad, spans, st, gather, lds_reserve,
                                              ctx->nym_two_phase, side.stream ? &side : nullptr);
    if (timed) hipEventRecord(ctx->ev1, st);
    ctx->release_qws(wi, st);
    if (timed) ctx->timed = true;
    return hip_to_rc(launched(ctx, err));
}

int fabgpu_idemix_nym_verify_batch_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off,
                                       const void* issuer_id, const void* nym_x, const void* nym_y, const void* proof_c,
                                       const void* proof_s_sk, const void* proof_s_r_nym, const void* nonce, void* verdict_bits,
                                       void* status, void* stream) {
    return nym_verify_dev(ctx, n, arena, arena_bytes, off, false, issuer_id, nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce, verdict_bits,
                          status, stream, true);
}

// ---- registered public keys -------------------------------------------------------------------------
// one key's comb table into one context: id of the key there (idempotent per (qx, qy)); `tab` = the table, built by the caller
static int key_install_table_locked(fabgpu_ctx* ctx, const std::string& k, int32_t* d, uint32_t* key_id);
static void key_queue_table16_locked(fabgpu_ctx* ctx, const std::string& k, uint32_t id);
// (kmu held, the context's device current) room for one table: out of the current slab, a new slab when that is used up
static int32_t* ktab_alloc_locked(fabgpu_ctx* ctx) {
    if (!ctx->ktab_free.empty()) {
        int32_t* d = ctx->ktab_free.back();
        ctx->ktab_free.pop_back();
        return d;
    }
    constexpr size_t table_bytes = sizeof(int32_t) * KeyTab8::TABLE_WORDS;
    if (ctx->ktab_slab_used >= fabgpu_ctx::KTAB_SLAB) {
        if (ctx->fault == 2) return nullptr;
        void* slab = nullptr;
        if (hipMalloc(&slab, table_bytes * fabgpu_ctx::KTAB_SLAB) != hipSuccess) return nullptr;
        ctx->ktab_slabs.push_back(slab);
        ctx->ktab_slab_used = 0;
    }
    return (int32_t*)((uint8_t*)ctx->ktab_slabs.back() + table_bytes * ctx->ktab_slab_used++);
}
static int key_install(fabgpu_ctx* ctx, const std::string& k, const std::vector<int32_t>* tab_in, const uint8_t* qx32, const uint8_t* qy32, uint32_t* key_id) {
    std::lock_guard<std::mutex> lk(ctx->kmu);
    auto it = ctx->key_ids.find(k);
    if (it != ctx->key_ids.end()) {
        *key_id = it->second;
        return FABGPU_OK;
    }
    if (ctx->ktabs.size() >= FABGPU_MAX_KEYS) return FABGPU_ENOMEM;
    DeviceGuard g(ctx->device);
    std::vector<int32_t> own;
    if (!tab_in) {
        u256 qx, qy;
        from_be32(qx, qx32);
        from_be32(qy, qy32);
        own.resize(KeyTab8::TABLE_WORDS);
        build_key_comb_table8(own.data(), qx, qy);
        tab_in = &own;
    }
    const std::vector<int32_t>& tab = *tab_in;
    int32_t* d = ktab_alloc_locked(ctx);
    if (!d) return FABGPU_ENOMEM;
    if (hipMemcpy(d, tab.data(), sizeof(int32_t) * KeyTab8::TABLE_WORDS, hipMemcpyHostToDevice) != hipSuccess) {
        ctx->ktab_free.push_back(d);
        return FABGPU_ELAUNCH;
    }
    return key_install_table_locked(ctx, k, d, key_id);
}
// (kmu held, the context's device current) a finished table in device memory becomes key number ktabs.size(); takes ownership of d
// FABGPU_FLAG_KEY_TABLES_16BIT: queue the build of key `id`'s 16-bit comb on stream_keytab and, behind it, the copy of its pointer into
// d_ktabs[2 id + 1].  Nothing waits: the registration returns, launches on other streams use the 8-bit comb until the pointer is there.
// kmu is held.  Failures (memory, a launch) leave the key without a 16-bit comb.
static void key_queue_table16_locked(fabgpu_ctx* ctx, const std::string& k, uint32_t id) {
    if (ctx->ktab16_count >= fabgpu_ctx::KTAB16_MAX || ctx->fault) return;
    if (ctx->ktab16.size() <= id) ctx->ktab16.resize((size_t)id + 1, nullptr);
    if (!ctx->stream_keytab16 && hipStreamCreateWithFlags(&ctx->stream_keytab16, hipStreamNonBlocking) != hipSuccess) return;
    const size_t tab_bytes = sizeof(int32_t) * GTab16::TABLE_WORDS;
    if (!ctx->ktab16_rooms) {
        ctx->ktab16_room_bytes = (keytab16_scratch_bytes() + 128 + 255) & ~(size_t)255;
        if (hipMalloc(&ctx->ktab16_rooms, ctx->ktab16_room_bytes * fabgpu_ctx::KTAB16_MAX) != hipSuccess) {
            (void)hipGetLastError();
            ctx->ktab16_rooms = nullptr;
            return;
        }
    }
    if (ctx->ktab16_carved == ctx->ktab16_slabs.size() * fabgpu_ctx::KTAB16_SLAB) {
        void* slab = nullptr;
        if (hipMalloc(&slab, tab_bytes * fabgpu_ctx::KTAB16_SLAB) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        ctx->ktab16_slabs.push_back(slab);
    }
    void* tab = (uint8_t*)ctx->ktab16_slabs.back() + tab_bytes * (ctx->ktab16_carved % fabgpu_ctx::KTAB16_SLAB);
    void* room = (uint8_t*)ctx->ktab16_rooms + ctx->ktab16_room_bytes * ctx->ktab16_count;
    hipStream_t st = ctx->stream_keytab16;
    // room: [0, 64) the key, [64, 72) the table's address (what the entries kernel reads), [128, ...) scratch
    if (!ctx->ktab16_heads && hipHostMalloc((void**)&ctx->ktab16_heads, 128 * fabgpu_ctx::KTAB16_MAX, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->ktab16_heads = nullptr;
        return;
    }
    uint8_t* head = ctx->ktab16_heads + 128 * ctx->ktab16_count;
    memcpy(head, k.data(), 64);
    memcpy(head + 64, &tab, sizeof(void*));
    hipError_t e = hipMemcpyAsync(room, head, 72, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_keytab16_build(1, room, (void* const*)((uint8_t*)room + 64), (uint8_t*)room + 128, st);
    if (e == hipSuccess) e = hipMemcpyAsync((void*)(ctx->d_ktabs + KTAB_STRIDE * (size_t)id + 1), (uint8_t*)room + 64, sizeof(void*), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        hipStreamSynchronize(st);                       // (the table's place in the slab and the room are simply used again by the next key)
        return;
    }
    ctx->ktab16[id] = tab;
    ctx->ktab16_carved++;
    ctx->ktab16_count++;
}

static int key_install_table_locked(fabgpu_ctx* ctx, const std::string& k, int32_t* d, uint32_t* key_id) {
    if (ctx->ktabs.size() >= FABGPU_MAX_KEYS) {
        ctx->ktab_free.push_back(d);
        return FABGPU_ENOMEM;
    }
    // grow the device-side pointer array by doubling; the old array is only released at shutdown, so launches already in
    // flight on other streams keep reading a valid (shorter) array
    if (ctx->ktabs.size() + 1 > ctx->d_ktabs_cap) {
        size_t cap = ctx->d_ktabs_cap ? ctx->d_ktabs_cap * 2 : 64;
        const int32_t** nd = nullptr;
        if (hipMalloc((void**)&nd, cap * KTAB_STRIDE * sizeof(int32_t*)) != hipSuccess) {
            ctx->ktab_free.push_back(d);
            return FABGPU_ENOMEM;
        }
        if (!ctx->ktabs.empty()) {
            // (the 16-bit builds queued on stream_keytab write their pointers into the OLD array: let them land before it is copied)
            if (ctx->ktab16_count && ctx->stream_keytab16) hipStreamSynchronize(ctx->stream_keytab16);
            std::vector<const int32_t*> both(ctx->ktabs.size() * KTAB_STRIDE, nullptr);
            for (size_t i = 0; i < ctx->ktabs.size(); i++) {
                both[KTAB_STRIDE * i] = ctx->ktabs[i];
                both[KTAB_STRIDE * i + 1] = i < ctx->ktab16.size() ? (const int32_t*)ctx->ktab16[i] : nullptr;
            }
            hipMemcpy((void*)nd, both.data(), both.size() * sizeof(int32_t*), hipMemcpyHostToDevice);
        }
        if (ctx->d_ktabs) ctx->retired.push_back((void*)ctx->d_ktabs);
        ctx->d_ktabs = nd;
        ctx->d_ktabs_cap = cap;
    }
    const int32_t* slot[KTAB_STRIDE] = {d, nullptr};
    if (hipMemcpy((void*)(ctx->d_ktabs + KTAB_STRIDE * ctx->ktabs.size()), slot, sizeof(slot), hipMemcpyHostToDevice) != hipSuccess) {
        ctx->ktab_free.push_back(d);
        return FABGPU_ELAUNCH;
    }
    ctx->ktabs.push_back(d);
    *key_id = (uint32_t)(ctx->ktabs.size() - 1);
    ctx->key_ids[k] = *key_id;
    if (ctx->key_tables_16) key_queue_table16_locked(ctx, k, *key_id);    // (best effort: a key without one is served by its 8-bit comb)
    return FABGPU_OK;
}

// n keys (qxy: n x 64 bytes, X || Y; every one an affine point of the curve - the caller's gate) registered on ONE context with their
// comb tables built ON THE DEVICE (keytab_kernels.hip: three launches for the whole batch, 0.7-0.9 ms, against 6 ms of host arithmetic
// per key).  key_ids[i] = the key's id (keys the context already has keep theirs); ids are handed out in the order of qxy.  Any failure:
// nothing was installed by this call that is not complete, the error is returned and the caller may fall back to the host builder.
static int key_register_batch_dev(fabgpu_ctx* ctx, int n, const uint8_t* qxy, uint32_t* key_ids) {
    if (!ctx || n <= 0 || !qxy || !key_ids) return FABGPU_EINVAL;
    if (ctx->fault) return ctx->fault == 2 ? FABGPU_ENOMEM : FABGPU_ELAUNCH;
    static const bool timing = getenv("FABGPU_PASS_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::lock_guard<std::mutex> blk(ctx->ktab_build_mu);        // one batch at a time per context: the builder's room and stream are one
    DeviceGuard g(ctx->device);
    std::vector<int> todo;                                     // indices into qxy that need a table (first occurrence of a key only)
    std::vector<std::string> names((size_t)n);
    std::vector<int32_t*> tabs;
    {
        std::lock_guard<std::mutex> lk(ctx->kmu);
        for (int i = 0; i < n; i++) {
            names[(size_t)i].assign((const char*)qxy + 64 * (size_t)i, 64);
            bool dup = false;
            for (int j : todo) dup = dup || names[(size_t)j] == names[(size_t)i];
            if (!dup && ctx->key_ids.find(names[(size_t)i]) == ctx->key_ids.end()) todo.push_back(i);
        }
        if (ctx->ktabs.size() + todo.size() > FABGPU_MAX_KEYS) return FABGPU_ENOMEM;
        for (size_t t = 0; t < todo.size(); t++) {
            int32_t* d = ktab_alloc_locked(ctx);
            if (!d) {
                for (auto* x : tabs) ctx->ktab_free.push_back(x);
                return FABGPU_ENOMEM;
            }
            tabs.push_back(d);
        }
    }
    const uint32_t m = (uint32_t)todo.size();
    auto give_back = [&](int code) {
        std::lock_guard<std::mutex> lk(ctx->kmu);
        for (auto* x : tabs)
            if (x) ctx->ktab_free.push_back(x);
        return code;
    };
    double t_room = 0, t_build = 0;
    if (m) {
        // one small upload: the keys, then the table pointers; the builder's room grows with the largest batch so far
        const size_t in_bytes = (size_t)64 * m + sizeof(void*) * m, scr_bytes = keytab_scratch_bytes(m);
        if (ctx->ktab_in_cap < in_bytes) {
            if (ctx->d_ktab_in) hipFree(ctx->d_ktab_in);
            ctx->d_ktab_in = nullptr;
            ctx->ktab_in_cap = 0;
            if (hipMalloc(&ctx->d_ktab_in, in_bytes * 2) != hipSuccess) return give_back(FABGPU_ENOMEM);
            ctx->ktab_in_cap = in_bytes * 2;
        }
        if (ctx->ktab_scr_cap < scr_bytes) {
            if (ctx->d_ktab_scr) hipFree(ctx->d_ktab_scr);
            ctx->d_ktab_scr = nullptr;
            ctx->ktab_scr_cap = 0;
            if (hipMalloc(&ctx->d_ktab_scr, scr_bytes * 2) != hipSuccess) return give_back(FABGPU_ENOMEM);
            ctx->ktab_scr_cap = scr_bytes * 2;
        }
        if (!ctx->stream_keytab && hipStreamCreateWithFlags(&ctx->stream_keytab, hipStreamNonBlocking) != hipSuccess) return give_back(FABGPU_ELAUNCH);
        std::vector<uint8_t> in(in_bytes);
        for (uint32_t t = 0; t < m; t++) memcpy(&in[64 * (size_t)t], qxy + 64 * (size_t)todo[t], 64);
        memcpy(&in[(size_t)64 * m], tabs.data(), sizeof(void*) * m);
        t_room = since();
        hipStream_t st = ctx->stream_keytab;
        hipError_t e = hipMemcpyAsync(ctx->d_ktab_in, in.data(), in_bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = launch_keytab_build(m, ctx->d_ktab_in, (void* const*)((uint8_t*)ctx->d_ktab_in + (size_t)64 * m), ctx->d_ktab_scr, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return give_back(hip_to_rc(e));
        t_build = since();
    }
    int rc = FABGPU_OK;
    {
        std::lock_guard<std::mutex> lk(ctx->kmu);
        for (uint32_t t = 0; t < m; t++) {
            const std::string& k = names[(size_t)todo[t]];
            uint32_t id = 0;
            if (rc == FABGPU_OK && ctx->key_ids.find(k) == ctx->key_ids.end()) rc = key_install_table_locked(ctx, k, tabs[t], &id);   // (takes the table, also on failure)
            else ctx->ktab_free.push_back(tabs[t]);             // (somebody registered it meanwhile, or an earlier install failed)
            tabs[t] = nullptr;
        }
        if (rc == FABGPU_OK)
            for (int i = 0; i < n; i++) {
                auto it = ctx->key_ids.find(names[(size_t)i]);
                if (it == ctx->key_ids.end()) {
                    rc = FABGPU_ELAUNCH;
                    break;
                }
                key_ids[i] = it->second;
            }
    }
    if (timing && m) fprintf(stderr, "fabgpu: %u key tables on the device: room %.2f ms, build %.2f ms, installed after %.2f ms\n", m, t_room, t_build - t_room, since());
    return rc;
}

int fabgpu_p256_key_register(fabgpu_ctx* ctx, const uint8_t* qx32, const uint8_t* qy32, uint32_t* key_id) {
    if (!ctx || !qx32 || !qy32 || !key_id) return FABGPU_EINVAL;
    if (!fabgpu_p256_pubkey_on_curve(qx32, qy32)) return FABGPU_EINVAL;   // KeyImport gate: such keys stay with bccsp/sw
    std::string k((const char*)qx32, 32);
    k.append((const char*)qy32, 32);
    // the table is built on the device; should that fail (memory, a launch), by the host builder as before round 6
    if (key_register_batch_dev(ctx, 1, (const uint8_t*)k.data(), key_id) == FABGPU_OK) return FABGPU_OK;
    return key_install(ctx, k, nullptr, qx32, qy32, key_id);
}

// The same key on n contexts (a provider that owns every GPU of the node): the 640 KiB comb table is built ONCE on the host (~6 ms)
// and uploaded to each context that does not have the key yet.  key_ids[g] = the key's id on ctxs[g].  The first failure is returned
// (contexts before it keep the key: registration is idempotent).
static int key_register_many_impl(fabgpu_ctx* const* ctxs, int n, const uint8_t* qx32, const uint8_t* qy32, const int32_t* prebuilt, uint32_t* key_ids) {
    if (!ctxs || n <= 0 || !qx32 || !qy32 || !key_ids) return FABGPU_EINVAL;
    for (int g = 0; g < n; g++)
        if (!ctxs[g]) return FABGPU_EINVAL;
    if (!fabgpu_p256_pubkey_on_curve(qx32, qy32)) return FABGPU_EINVAL;
    std::string k((const char*)qx32, 32);
    k.append((const char*)qy32, 32);
    std::vector<int32_t> tab;                              // built when the first context turns out to need it (unless the caller brought one)
    if (prebuilt) tab.assign(prebuilt, prebuilt + KeyTab8::TABLE_WORDS);
    for (int g = 0; g < n; g++) {
        bool have;
        {
            std::lock_guard<std::mutex> lk(ctxs[g]->kmu);
            auto it = ctxs[g]->key_ids.find(k);
            have = it != ctxs[g]->key_ids.end();
            if (have) key_ids[g] = it->second;
        }
        if (have) continue;
        if (tab.empty() && key_register_batch_dev(ctxs[g], 1, (const uint8_t*)k.data(), &key_ids[g]) == FABGPU_OK) continue;   // built on that device
        if (tab.empty()) {
            u256 qx, qy;
            from_be32(qx, qx32);
            from_be32(qy, qy32);
            tab.resize(KeyTab8::TABLE_WORDS);
            build_key_comb_table8(tab.data(), qx, qy);
        }
        const int rc = key_install(ctxs[g], k, &tab, qx32, qy32, &key_ids[g]);
        if (rc != FABGPU_OK) return rc;
    }
    return FABGPU_OK;
}
int fabgpu_p256_key_register_many(fabgpu_ctx* const* ctxs, int n, const uint8_t* qx32, const uint8_t* qy32, uint32_t* key_ids) {
    return key_register_many_impl(ctxs, n, qx32, qy32, nullptr, key_ids);
}

int fabgpu_p256_key_lookup(fabgpu_ctx* ctx, const uint8_t* qx32, const uint8_t* qy32, uint32_t* key_id) {
    if (!ctx || !qx32 || !qy32 || !key_id) return FABGPU_EINVAL;
    std::string k((const char*)qx32, 32);
    k.append((const char*)qy32, 32);
    std::lock_guard<std::mutex> lk(ctx->kmu);
    auto it = ctx->key_ids.find(k);
    if (it == ctx->key_ids.end()) return 1;
    *key_id = it->second;
    return FABGPU_OK;
}

int fabgpu_p256_key_count(fabgpu_ctx* ctx) {
    if (!ctx) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->kmu);
    return (int)ctx->ktabs.size();
}

int fabgpu_p256_verify_batch_keyed_dev(fabgpu_ctx* ctx, size_t n, const void* key_id, const void* e, const void* r, const void* s,
                                       void* verdict_bits, void* status, void* stream) {
    if (!ctx || (n && (!key_id || !e || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0xFFFFFFF0ull) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    uint32_t nkeys;
    const int32_t** kt;
    {
        std::lock_guard<std::mutex> lk(ctx->kmu);
        nkeys = (uint32_t)ctx->ktabs.size();
        kt = ctx->d_ktabs;
    }
    if (nkeys == 0) return FABGPU_EINVAL;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    if (ctx->allow_wide && n <= (size_t)WIDE_LAUNCH_MAX) {
        // a launch that cannot fill the chip: eight lanes per signature, the digest-independent half first (kernels.h, p256_wide29.h)
        size_t wi = 0;
        void* wsp = nullptr;
        int rc = ctx->acquire_qws(n * WIDE_SCRATCH_BYTES, &wi, &wsp, st);
        if (rc != FABGPU_OK) return rc;
        if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
        hipError_t err = launch_p256_wide_pre((uint32_t)n, key_id, nkeys, (const void*)kt, r, s, ctx->d_gtab, wsp, st);
        if (err == hipSuccess) err = launch_p256_wide_post((uint32_t)n, e, r, ctx->d_gtab, wsp, verdict_bits, status, st);
        if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
        ctx->release_qws(wi, st);
        ctx->timed = ctx->time_kernels;
        return hip_to_rc(launched(ctx, err));
    }
    if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
    hipError_t err = launch_p256_verify_keyed((uint32_t)n, key_id, nkeys, (const void*)kt, e, r, s, ctx->d_gtab, verdict_bits, status, ctx->allow_pair, st);
    if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
    ctx->timed = ctx->time_kernels;
    return hip_to_rc(launched(ctx, err));
}

// identity.Verify for registered keys on a launch that cannot fill the chip: the digest-independent half of the verification, the
// hashes (mid-states of shared prefixes first, unless the caller has them), then the half behind the digest.  One stream here - the
// block pass runs the first two side by side (walk_block_pass).  digests_out: the caller's n x 32 bytes, or nullptr (scratch is used).
static int keyed_wide_identity_dev(fabgpu_ctx* ctx, uint32_t n, const void* arena, size_t arena_bytes, const void* off, const void* key_id, uint32_t nkeys,
                                   const void* kt, const void* r, const void* s, void* verdict_bits, void* status, ShaPrefixArgs pa, hipStream_t st) {
    size_t wi = 0;
    void* wsp = nullptr;
    int rc = ctx->acquire_qws((size_t)n * (WIDE_SCRATCH_BYTES + 32), &wi, &wsp, st);
    if (rc != FABGPU_OK) return rc;
    void* dig = pa.digests ? pa.digests : (void*)((uint8_t*)wsp + (size_t)n * WIDE_SCRATCH_BYTES);
    if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
    hipError_t err = launch_p256_wide_pre(n, key_id, nkeys, kt, r, s, ctx->d_gtab, wsp, st);
    const bool coop = n <= SHA_COOP_MAX;                 // eight lanes per message, prefixed ones hashed whole: no mid-states (sha256_coop.h)
    if (err == hipSuccess && !coop && pa.m && pa.pre_idx && !pa.mid_ready) err = launch_sha256_midstates(arena, arena_bytes, pa, st);
    pa.digests = dig;
    if (err == hipSuccess) err = coop ? launch_sha256_messages_coop(n, arena, arena_bytes, off, pa, st) : launch_sha256_messages(n, arena, arena_bytes, off, pa, st);
    if (err == hipSuccess) err = launch_p256_wide_post(n, dig, r, ctx->d_gtab, wsp, verdict_bits, status, st);
    if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
    ctx->release_qws(wi, st);
    ctx->timed = ctx->time_kernels;
    return hip_to_rc(launched(ctx, err));
}

int fabgpu_sha256_p256_verify_batch_keyed_dev(fabgpu_ctx* ctx, size_t n, const void* arena, size_t arena_bytes, const void* off,
                                              const void* key_id, const void* r, const void* s, void* verdict_bits, void* status, void* stream) {
    if (!ctx || (n && (!arena || !off || !key_id || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0xFFFFFFF0ull || arena_bytes > 0xFFFFFFFFull) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    uint32_t nkeys;
    const int32_t** kt;
    {
        std::lock_guard<std::mutex> lk(ctx->kmu);
        nkeys = (uint32_t)ctx->ktabs.size();
        kt = ctx->d_ktabs;
    }
    if (nkeys == 0) return FABGPU_EINVAL;
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    if (ctx->allow_wide && n <= (size_t)WIDE_LAUNCH_MAX)
        return keyed_wide_identity_dev(ctx, (uint32_t)n, arena, arena_bytes, off, key_id, nkeys, (const void*)kt, r, s, verdict_bits, status, ShaPrefixArgs(), st);
    if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
    hipError_t err = launch_sha256_p256_verify_keyed((uint32_t)n, arena, arena_bytes, off, key_id, nkeys, (const void*)kt, r, s, ctx->d_gtab, verdict_bits,
                                                     status, ctx->allow_pair, ShaPrefixArgs(), st);
    if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
    ctx->timed = ctx->time_kernels;
    return hip_to_rc(launched(ctx, err));
}

// ---- host-pointer entry points (what the cgo provider binds) -----------------------------------------
int fabgpu_p256_verify_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* e, const uint8_t* r,
                             const uint8_t* s, uint64_t* verdict_bits, uint8_t* status) {
    if (!ctx || (n && (!qx || !qy || !e || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 160) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    const size_t fb = n * 32, words = (n + 63) / 64;
    const size_t st_off = round_up(words * 8, 64);
    int rc;
    if ((rc = ctx->fields.ensure(5 * fb)) || (rc = ctx->out.ensure(st_off + n))) return rc;
    uint8_t* h = (uint8_t*)ctx->fields.h;
    uint8_t* d = (uint8_t*)ctx->fields.d;
    uint8_t* dout = (uint8_t*)ctx->out.d;
    // Field by field: the copy of a field into the pinned staging buffer, then ITS transfer - the DMA of field k runs while the host
    // copies field k + 1 (round 5; one transfer of all five behind all five copies left the bus idle for the copies' 0.15 ms and the
    // host idle for the transfer's 0.1 ms).  Small batches stay one transfer: five API calls would cost more than they hide.
    const uint8_t* src[5] = {qx, qy, e, r, s};
    hipError_t err = hipSuccess;
    if (fb >= ((size_t)256 << 10)) {
        for (int f = 0; f < 5 && err == hipSuccess; f++) {
            memcpy(h + (size_t)f * fb, src[f], fb);
            err = hipMemcpyAsync(d + (size_t)f * fb, h + (size_t)f * fb, fb, hipMemcpyHostToDevice, ctx->stream);
        }
    } else {
        for (int f = 0; f < 5; f++) memcpy(h + (size_t)f * fb, src[f], fb);
        err = hipMemcpyAsync(d, h, 5 * fb, hipMemcpyHostToDevice, ctx->stream);
    }
    if (err != hipSuccess) {
        hipStreamSynchronize(ctx->stream);                  // (nothing may still be reading the staging buffer when the next call refills it)
        return hip_to_rc(err);
    }
    rc = fabgpu_p256_verify_batch_dev(ctx, n, d, d + fb, d + 2 * fb, d + 3 * fb, d + 4 * fb, dout, status ? dout + st_off : nullptr, ctx->stream);
    if (rc) return rc;
    err = hipMemcpyAsync(ctx->out.h, dout, status ? st_off + n : words * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    memcpy(verdict_bits, ctx->out.h, words * 8);
    if (status) memcpy(status, (uint8_t*)ctx->out.h + st_off, n);
    return FABGPU_OK;
}

int fabgpu_p256_verify_batch_keyed(fabgpu_ctx* ctx, size_t n, const uint32_t* key_id, const uint8_t* e, const uint8_t* r, const uint8_t* s,
                                   uint64_t* verdict_bits, uint8_t* status) {
    if (!ctx || (n && (!key_id || !e || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 100) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    const size_t fb = n * 32, kb = round_up(n * 4, 64), words = (n + 63) / 64;
    const size_t st_off = round_up(words * 8, 64);
    int rc;
    if ((rc = ctx->keyed.ensure(kb + 3 * fb)) || (rc = ctx->out.ensure(st_off + n))) return rc;
    uint8_t* h = (uint8_t*)ctx->keyed.h;
    memcpy(h, key_id, n * 4); memcpy(h + kb, e, fb); memcpy(h + kb + fb, r, fb); memcpy(h + kb + 2 * fb, s, fb);
    uint8_t* d = (uint8_t*)ctx->keyed.d;
    uint8_t* dout = (uint8_t*)ctx->out.d;
    hipError_t err = hipMemcpyAsync(d, h, kb + 3 * fb, hipMemcpyHostToDevice, ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    rc = fabgpu_p256_verify_batch_keyed_dev(ctx, n, d, d + kb, d + kb + fb, d + kb + 2 * fb, dout, status ? dout + st_off : nullptr, ctx->stream);
    if (rc) return rc;
    err = hipMemcpyAsync(ctx->out.h, dout, status ? st_off + n : words * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    memcpy(verdict_bits, ctx->out.h, words * 8);
    if (status) memcpy(status, (uint8_t*)ctx->out.h + st_off, n);
    return FABGPU_OK;
}

// copy the span of the arena the offsets reference; returns rebased offsets in ctx->offs.h
static int stage_messages(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, size_t* arena_bytes_out) {
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (size_t i = 0; i < n; i++) {
        if (off[i + 1] < off[i]) return FABGPU_EINVAL;
        if (off[i] < lo) lo = off[i];
        if (off[i + 1] > hi) hi = off[i + 1];
    }
    if (n == 0) { lo = hi = 0; }
    size_t span = hi >= lo ? (size_t)hi - lo : 0;
    int rc;
    if ((rc = ctx->arena.ensure(round_up(span, 4) + 128)) || (rc = ctx->offs.ensure((n + 1) * 4))) return rc;
    if (span) memcpy(ctx->arena.h, arena + lo, span);
    memset((uint8_t*)ctx->arena.h + span, 0, round_up(span, 4) + 64 - span);
    uint32_t* ho = (uint32_t*)ctx->offs.h;
    for (size_t i = 0; i <= n; i++) ho[i] = off[i] - lo;
    hipError_t err = hipMemcpyAsync(ctx->arena.d, ctx->arena.h, round_up(span, 4) + 64, hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(ctx->offs.d, ctx->offs.h, (n + 1) * 4, hipMemcpyHostToDevice, ctx->stream);
    *arena_bytes_out = round_up(span, 4) + 64;
    return hip_to_rc(err);
}

int fabgpu_sha256_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, uint8_t* digests) {
    if (!ctx || (n && (!off || !digests))) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 32) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    if (!arena && off[n] != off[0]) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    size_t ab = 0;
    int rc = stage_messages(ctx, n, arena, off, &ab);
    if (rc) return rc;
    if ((rc = ctx->out.ensure(n * 32))) return rc;
    rc = fabgpu_sha256_batch_dev(ctx, n, ctx->arena.d, ab, ctx->offs.d, ctx->out.d, ctx->stream);
    if (rc) return rc;
    hipError_t err = hipMemcpyAsync(ctx->out.h, ctx->out.d, n * 32, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    memcpy(digests, ctx->out.h, n * 32);
    return FABGPU_OK;
}

int fabgpu_sha256_p256_verify_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, const uint8_t* qx,
                                    const uint8_t* qy, const uint8_t* r, const uint8_t* s, uint64_t* verdict_bits, uint8_t* status) {
    if (!ctx || (n && (!off || !qx || !qy || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 160) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    if (!arena && off[n] != off[0]) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    size_t ab = 0;
    int rc = stage_messages(ctx, n, arena, off, &ab);
    if (rc) return rc;
    const size_t fb = n * 32, words = (n + 63) / 64;
    const size_t st_off = round_up(words * 8, 64);
    if ((rc = ctx->fields.ensure(4 * fb)) || (rc = ctx->out.ensure(st_off + n))) return rc;
    uint8_t* h = (uint8_t*)ctx->fields.h;
    memcpy(h, qx, fb); memcpy(h + fb, qy, fb); memcpy(h + 2 * fb, r, fb); memcpy(h + 3 * fb, s, fb);
    uint8_t* d = (uint8_t*)ctx->fields.d;
    uint8_t* dout = (uint8_t*)ctx->out.d;
    hipError_t err = hipMemcpyAsync(d, h, 4 * fb, hipMemcpyHostToDevice, ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    rc = fabgpu_sha256_p256_verify_batch_dev(ctx, n, ctx->arena.d, ab, ctx->offs.d, d, d + fb, d + 2 * fb, d + 3 * fb, dout,
                                             status ? dout + st_off : nullptr, ctx->stream);
    if (rc) return rc;
    err = hipMemcpyAsync(ctx->out.h, dout, status ? st_off + n : words * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    memcpy(verdict_bits, ctx->out.h, words * 8);
    if (status) memcpy(status, (uint8_t*)ctx->out.h + st_off, n);
    return FABGPU_OK;
}

int fabgpu_idemix_nym_verify_batch(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, const uint32_t* issuer_id,
                                   const uint8_t* nym_x, const uint8_t* nym_y, const uint8_t* proof_c, const uint8_t* proof_s_sk,
                                   const uint8_t* proof_s_r_nym, const uint8_t* nonce, uint64_t* verdict_bits, uint8_t* status) {
    if (!ctx || (n && (!off || !nym_x || !nym_y || !proof_c || !proof_s_sk || !proof_s_r_nym || !nonce || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 200) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    if (!arena && off[n] != off[0]) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    size_t ab = 0;
    int rc = stage_messages(ctx, n, arena, off, &ab);
    if (rc) return rc;
    const size_t fb = n * 32, kb = round_up(n * 4, 64), words = (n + 63) / 64;
    const size_t st_off = round_up(words * 8, 64);
    if ((rc = ctx->nym.ensure(kb + 6 * fb)) || (rc = ctx->out.ensure(st_off + n))) return rc;
    uint8_t* h = (uint8_t*)ctx->nym.h;
    if (issuer_id) memcpy(h, issuer_id, n * 4);
    const uint8_t* src[6] = {nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce};
    for (int f = 0; f < 6; f++) memcpy(h + kb + f * fb, src[f], fb);
    uint8_t* d = (uint8_t*)ctx->nym.d;
    uint8_t* dout = (uint8_t*)ctx->out.d;
    hipError_t err = hipMemcpyAsync(d, h, kb + 6 * fb, hipMemcpyHostToDevice, ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    rc = fabgpu_idemix_nym_verify_batch_dev(ctx, n, ctx->arena.d, ab, ctx->offs.d, issuer_id ? d : nullptr, d + kb, d + kb + fb, d + kb + 2 * fb,
                                            d + kb + 3 * fb, d + kb + 4 * fb, d + kb + 5 * fb, dout, status ? dout + st_off : nullptr, ctx->stream);
    if (rc) return rc;
    err = hipMemcpyAsync(ctx->out.h, dout, status ? st_off + n : words * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    memcpy(verdict_bits, ctx->out.h, words * 8);
    if (status) memcpy(status, (uint8_t*)ctx->out.h + st_off, n);
    return FABGPU_OK;
}

int fabgpu_sha256_p256_verify_batch_keyed(fabgpu_ctx* ctx, size_t n, const uint8_t* arena, const uint32_t* off, const uint32_t* key_id,
                                          const uint8_t* r, const uint8_t* s, uint64_t* verdict_bits, uint8_t* status) {
    if (!ctx || (n && (!off || !key_id || !r || !s || !verdict_bits))) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 160) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    if (!arena && off[n] != off[0]) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    size_t ab = 0;
    int rc = stage_messages(ctx, n, arena, off, &ab);
    if (rc) return rc;
    const size_t fb = n * 32, kb = round_up(n * 4, 64), words = (n + 63) / 64;
    const size_t st_off = round_up(words * 8, 64);
    if ((rc = ctx->keyed.ensure(kb + 2 * fb)) || (rc = ctx->out.ensure(st_off + n))) return rc;
    uint8_t* h = (uint8_t*)ctx->keyed.h;
    memcpy(h, key_id, n * 4); memcpy(h + kb, r, fb); memcpy(h + kb + fb, s, fb);
    uint8_t* d = (uint8_t*)ctx->keyed.d;
    uint8_t* dout = (uint8_t*)ctx->out.d;
    hipError_t err = hipMemcpyAsync(d, h, kb + 2 * fb, hipMemcpyHostToDevice, ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    rc = fabgpu_sha256_p256_verify_batch_keyed_dev(ctx, n, ctx->arena.d, ab, ctx->offs.d, d, d + kb, d + kb + fb, dout,
                                                   status ? dout + st_off : nullptr, ctx->stream);
    if (rc) return rc;
    err = hipMemcpyAsync(ctx->out.h, dout, status ? st_off + n : words * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    memcpy(verdict_bits, ctx->out.h, words * 8);
    if (status) memcpy(status, (uint8_t*)ctx->out.h + st_off, n);
    return FABGPU_OK;
}

// ---- identity.Verify over a described batch: optional shared prefixes, fresh or registered keys ------------------------
int fabgpu_identity_verify_batch_dev(fabgpu_ctx* ctx, const fabgpu_identity_batch* b, void* mid_scratch, void* stream) {
    if (!ctx || !b) return FABGPU_EINVAL;
    const size_t n = b->n;
    if (n == 0) return FABGPU_OK;
    const bool keyed = b->key_id != nullptr;
    const bool prefixed = b->n_prefixes != 0 && b->pre_idx != nullptr;
    if (!b->arena || !b->off || !b->r || !b->s || !b->verdict_bits || (!keyed && (!b->qx || !b->qy))) return FABGPU_EINVAL;
    if (prefixed && (!b->pre_off || !mid_scratch)) return FABGPU_EINVAL;
    if (b->flags & ~(uint32_t)FABGPU_IDB_SPANS) return FABGPU_EINVAL;
    if (n > 0xFFFFFFF0ull || b->arena_bytes > 0xFFFFFFFFull) return FABGPU_ETOOBIG;
    if (b->n_gather && (!b->gather_spans || !b->gather_digests || !b->gather_off || !b->gather_scratch || b->gather_scratch_bytes > 0xFFFFFFFFull))
        return FABGPU_EINVAL;
    ShaPrefixArgs pa;
    pa.spans = (b->flags & FABGPU_IDB_SPANS) != 0;
    pa.digests = b->digests;
    if (prefixed) {
        pa.m = b->n_prefixes;
        pa.pre_off = b->pre_off;
        pa.pre_idx = b->pre_idx;
        pa.mid_scratch = mid_scratch;
    }
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    hipError_t err;
    if (keyed) {
        uint32_t nkeys;
        const int32_t** kt;
        {
            std::lock_guard<std::mutex> lk(ctx->kmu);
            nkeys = (uint32_t)ctx->ktabs.size();
            kt = ctx->d_ktabs;
        }
        if (nkeys == 0) return FABGPU_EINVAL;
        if (ctx->allow_wide && n <= (size_t)WIDE_LAUNCH_MAX) {
            int rc = keyed_wide_identity_dev(ctx, (uint32_t)n, b->arena, b->arena_bytes, b->off, b->key_id, nkeys, (const void*)kt, b->r, b->s, b->verdict_bits,
                                             b->status, pa, st);
            if (rc != FABGPU_OK) return rc;
            err = hipSuccess;
            if (b->n_gather)
                err = launch_gather_sha256(b->n_gather, b->arena, b->arena_bytes, b->gather_spans, b->gather_off, b->gather_scratch, b->gather_scratch_bytes,
                                           b->gather_digests, st);
            return hip_to_rc(err);
        }
        if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
        err = launch_sha256_p256_verify_keyed((uint32_t)n, b->arena, b->arena_bytes, b->off, b->key_id, nkeys, (const void*)kt, b->r, b->s, ctx->d_gtab,
                                              b->verdict_bits, b->status, ctx->allow_pair, pa, st);
        if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
    } else {
        size_t wi = 0;
        void* wsp = nullptr;
        int rc = ctx->acquire_qws(verify_workspace_bytes((uint32_t)n, ctx->allow_pair), &wi, &wsp, st);
        if (rc != FABGPU_OK) return rc;
        if (ctx->time_kernels) hipEventRecord(ctx->ev0, st);
        err = launch_sha256_p256_verify((uint32_t)n, b->arena, b->arena_bytes, b->off, b->qx, b->qy, b->r, b->s, ctx->d_gtab, wsp,
                                        b->verdict_bits, b->status, ctx->allow_pair, pa, st);
        if (ctx->time_kernels) hipEventRecord(ctx->ev1, st);
        ctx->release_qws(wi, st);
    }
    ctx->timed = ctx->time_kernels;
    err = launched(ctx, err);
    if (err == hipSuccess && b->n_gather)   // behind the verification on the same stream; not part of fabgpu_last_kernel_ms
        err = launch_gather_sha256(b->n_gather, b->arena, b->arena_bytes, b->gather_spans, b->gather_off, b->gather_scratch, b->gather_scratch_bytes,
                                   b->gather_digests, st);
    return hip_to_rc(err);
}

}  // extern "C"

static int keep_acquire(fabgpu_ctx* ctx, size_t len);
namespace fab {
// `n` pool buffers of block_bytes each, pinned now (provider construction) instead of inside the first passes' uploads
void host_copy_preallocate(fabgpu_ctx* ctx, size_t block_bytes, uint32_t n) {
    if (!ctx || !block_bytes) return;
    DeviceGuard g(ctx->device);
    std::vector<int> got;
    for (uint32_t i = 0; i < n; i++) {
        const int idx = keep_acquire(ctx, block_bytes);
        if (idx < 0) break;
        got.push_back(idx);
    }
    // ... and sent through the DMA path once, whole: the first transfer out of a freshly registered range costs milliseconds (measured
    // with tools/gpu_r06_first_memo_probe.py: 2.2 ms "wait for upload" on a provider's first memo-seeding pass against 0.2-0.3 after)
    void* d = nullptr;
    if (!got.empty() && hipMalloc(&d, block_bytes) == hipSuccess) {
        for (int idx : got) (void)hipMemcpy(d, ctx->keep_pool[(size_t)idx]->pin.h, block_bytes, hipMemcpyHostToDevice);
        hipFree(d);
    }
    std::lock_guard<std::mutex> lk(ctx->keep_mu);
    for (int idx : got) ctx->keep_pool[(size_t)idx]->in_use = false;
    ctx->keep_refused = 0;
}
void host_copy_release(HostCopy* c) {
    if (!c || !c->ctx || c->idx < 0) {
        if (c) *c = HostCopy();
        return;
    }
    {
        std::lock_guard<std::mutex> lk(c->ctx->keep_mu);
        if ((size_t)c->idx < c->ctx->keep_pool.size()) c->ctx->keep_pool[(size_t)c->idx]->in_use = false;
    }
    *c = HostCopy();
}
void host_copy_limit(fabgpu_ctx* ctx, uint32_t blocks) {
    if (!ctx) return;
    std::lock_guard<std::mutex> lk(ctx->keep_mu);
    ctx->keep_max = blocks > 64 ? 64 : blocks;
}
void host_copy_stats(fabgpu_ctx* ctx, uint64_t* held, uint64_t* bytes_held, uint64_t* refused) {
    uint64_t h = 0, b = 0, r = 0;
    if (ctx) {
        std::lock_guard<std::mutex> lk(ctx->keep_mu);
        for (const auto& k : ctx->keep_pool)
            if (k->in_use) {
                h++;
                b += k->pin.cap;
            }
        r = ctx->keep_refused;
    }
    if (held) *held = h;
    if (bytes_held) *bytes_held = b;
    if (refused) *refused = r;
}
}  // namespace fab

// a buffer of the keep pool with room for `len` bytes, marked in use; -1: none to spare (or fault injection "oom")
static int keep_acquire(fabgpu_ctx* ctx, size_t len) {
    if (ctx->fault == 2) return -1;
    int idx = -1;
    {
        std::lock_guard<std::mutex> lk(ctx->keep_mu);
        if (ctx->keep_pool.capacity() < 64) ctx->keep_pool.reserve(64);   // (holders read their entry without the lock: the array never moves)
        // a free buffer that is already large enough, else any free one, else a new one while the pool may grow
        for (size_t i = 0; i < ctx->keep_pool.size() && idx < 0; i++)
            if (!ctx->keep_pool[i]->in_use && ctx->keep_pool[i]->pin.cap >= len) idx = (int)i;
        for (size_t i = 0; i < ctx->keep_pool.size() && idx < 0; i++)
            if (!ctx->keep_pool[i]->in_use) idx = (int)i;
        if (idx < 0 && ctx->keep_pool.size() < ctx->keep_max) {
            ctx->keep_pool.emplace_back(new fabgpu_ctx::KeepBuf);
            idx = (int)ctx->keep_pool.size() - 1;
        }
        if (idx < 0) {
            ctx->keep_refused++;
            return -1;
        }
        ctx->keep_pool[(size_t)idx]->in_use = true;
    }
    // (pinning outside the pool's lock: the buffer is ours)
    if (ctx->keep_pool[(size_t)idx]->pin.ensure(len) != FABGPU_OK) {
        std::lock_guard<std::mutex> lk(ctx->keep_mu);
        ctx->keep_pool[(size_t)idx]->in_use = false;
        ctx->keep_refused++;
        return -1;
    }
    return idx;
}

static int arena_stage_impl(fabgpu_ctx* ctx, const void* arena, size_t len, uint64_t* token, fab::HostCopy* keep);

extern "C" int fabgpu_arena_stage(fabgpu_ctx* ctx, const void* arena, size_t len, uint64_t* token) { return arena_stage_impl(ctx, arena, len, token, nullptr); }
namespace fab {
int arena_stage_keep(fabgpu_ctx* ctx, const void* arena, size_t len, uint64_t* token, HostCopy* keep) {
    if (keep) *keep = HostCopy();
    return arena_stage_impl(ctx, arena, len, token, keep);
}
}  // namespace fab

static int arena_stage_impl(fabgpu_ctx* ctx, const void* arena, size_t len, uint64_t* token, fab::HostCopy* keep) {
    if (!ctx || !arena || !token || len == 0) return FABGPU_EINVAL;
    if (len > 0xFFFFFF00ull) return FABGPU_ETOOBIG;
    // the least recently filled slot nobody is using; all in use: wait for the oldest
    fabgpu_ctx::Staged* sl = nullptr;
    std::unique_lock<std::mutex> slot_lk;
    {
        std::lock_guard<std::mutex> lk(ctx->smu);
        int order[fabgpu_ctx::N_STAGED];
        for (int i = 0; i < fabgpu_ctx::N_STAGED; i++) order[i] = i;
        std::sort(order, order + fabgpu_ctx::N_STAGED, [&](int x, int y) { return ctx->staged_slots[x].token.load() < ctx->staged_slots[y].token.load(); });
        for (int i = 0; i < fabgpu_ctx::N_STAGED && !sl; i++) {
            std::unique_lock<std::mutex> t(ctx->staged_slots[order[i]].m, std::try_to_lock);
            if (t.owns_lock()) {
                sl = &ctx->staged_slots[order[i]];
                slot_lk = std::move(t);
            }
        }
        if (!sl) sl = &ctx->staged_slots[order[0]];
    }
    if (!slot_lk.owns_lock()) slot_lk = std::unique_lock<std::mutex>(sl->m);
    DeviceGuard g(ctx->device);
    // the caller keeps a host copy: the bytes go through a pool buffer that is then his (whatever the arena's size)
    const int keep_idx = keep ? keep_acquire(ctx, len) : -1;
    struct KeepBack {                                      // an early exit hands the buffer back
        fabgpu_ctx* c;
        int idx;
        ~KeepBack() {
            if (idx < 0) return;
            std::lock_guard<std::mutex> lk(c->keep_mu);
            c->keep_pool[(size_t)idx]->in_use = false;
        }
    } keep_back{ctx, keep_idx};
    uint8_t* const keep_pin = keep_idx >= 0 ? (uint8_t*)ctx->keep_pool[(size_t)keep_idx]->pin.h : nullptr;
    const size_t need = round_up(len, 64) + 128;          // + room for a batch's tail (fabgpu_identity_batch.tail) in the slack below
    sl->token.store(0);                                    // the previous upload is gone from here on
    sl->len = 0;
    if (sl->cap < need + (64 << 10)) {                     // (a batch reading the old buffer would hold the slot's mutex)
        if (sl->d) hipFree(sl->d);
        sl->d = nullptr;
        sl->cap = 0;
        if (hipMalloc(&sl->d, need + need / 8 + (64 << 10)) != hipSuccess) return FABGPU_ENOMEM;
        sl->cap = need + need / 8 + (64 << 10);
    }
    // A block arrives in memory the runtime has never seen (a peer's blocks are fresh allocations): a pageable hipMemcpy of such
    // memory spends more time making the pages DMA-able than moving them - 50 MB took 2.0-2.5 ms against 0.9 ms for a buffer that was
    // sent before, which the runtime keeps pinned (tools/gpu_probe_fresh_buffers.py).  Arenas of 4 MiB and more therefore travel
    // through a pinned staging buffer the context owns: a few threads copy 2 MiB pieces into it and queue each piece's DMA as soon as
    // it is there (four copiers): 2.2 ms per 10 000-transaction pass whatever the
    // buffer's history, against 2.9-3.9 ms (fresh) / 1.85 ms (re-sent, which a peer never does) on the pageable path.  Splitting the pageable copy itself over threads
    // was measured too: slower than one call (the pinning serialises in the driver).
    static const bool stage_timing = getenv("FABGPU_PASS_TIMING") != nullptr;
    const auto stage_t0 = std::chrono::steady_clock::now();
    constexpr int stage_threads = 4;             // measured from 0 (the runtime's pageable path) to 16 in rounds 2-3; no difference from two up
    hipError_t err = hipSuccess;
    if (stage_threads == 0 || len < ((size_t)4 << 20)) {
        if (keep_pin) {                                                    // a small arena somebody keeps: one copy, then the DMA from pinned memory
            memcpy(keep_pin, arena, len);
            err = hipMemcpy(sl->d, keep_pin, len, hipMemcpyHostToDevice);
        } else {
            err = hipMemcpy(sl->d, arena, len, hipMemcpyHostToDevice);
        }
    } else {
        std::lock_guard<std::mutex> plk(ctx->stage_pin_mu);              // one upload at a time per device: uploads share the bus anyway
        if (ctx->fault == 2 || (!keep_pin && ctx->stage_pin.ensure(len) != FABGPU_OK)) return FABGPU_ENOMEM;
        // pieces: 256 KiB, 512 KiB, 1 MiB, then 2 MiB each - the first DMA starts after 25 us of copying instead of 200
        std::vector<size_t> cut;
        constexpr size_t max_piece = (size_t)2048 << 10;
        for (size_t at = 0, sz = (size_t)256 << 10; at < len; at += sz, sz = std::min(sz * 2, max_piece)) cut.push_back(at);
        cut.push_back(len);
        const size_t n_pieces = cut.size() - 1;
        std::atomic<size_t> next(0);
        std::unique_ptr<std::atomic<uint8_t>[]> done(new std::atomic<uint8_t>[n_pieces]);
        for (size_t k = 0; k < n_pieces; k++) done[k].store(0, std::memory_order_relaxed);
        int failed = 0;
        uint8_t* dst = (uint8_t*)sl->d;
        uint8_t* pin = keep_pin ? keep_pin : (uint8_t*)ctx->stage_pin.h;
        const uint8_t* src = (const uint8_t*)arena;
        // Pieces go round-robin over four upload queues so that one piece's set-up overlaps another's transfer:
        // measured (round 3, 48.6 MB) 1.33 ms with one queue, 1.23 with two, 1.13 with four - 43 GB/s; the
        // number of copier threads makes no difference from two up.
        constexpr int n_queues = 4;
        hipStream_t cs[4] = {ctx->stream_copy, ctx->stream_copy_more[0], ctx->stream_copy_more[1], ctx->stream_copy_more[2]};
        const int dev = ctx->device;
        // The copiers (the host side's worker pool: idle while a block travels) claim pieces in order; this thread queues a piece's DMA
        // as soon as it is in the staging buffer - it alone talks to the runtime (several threads queueing on one stream spend their
        // time on its lock) - and copies pieces itself while it has nothing to queue.
        auto copy_one = [&]() -> bool {
            const size_t k = next.fetch_add(1, std::memory_order_relaxed);
            if (k >= n_pieces) return false;
            memcpy(pin + cut[k], src + cut[k], cut[k + 1] - cut[k]);
            done[k].store(1, std::memory_order_release);
            return true;
        };
        const int nth = (int)std::min<size_t>((size_t)stage_threads, n_pieces);
        if (!ctx->stage_pool) ctx->stage_pool.reset(new (std::nothrow) WorkerPool(stage_threads));   // (stage_pin_mu is held)
        run_workers(nth + 1, [&](int w) {
            if (w != 0) {
                while (copy_one()) {}
                return;
            }
            hipSetDevice(dev);
            for (size_t k = 0; k < n_pieces; k++) {
                while (!done[k].load(std::memory_order_acquire))
                    if (!copy_one()) std::this_thread::yield();
                if (hipMemcpyAsync(dst + cut[k], pin + cut[k], cut[k + 1] - cut[k], hipMemcpyHostToDevice, cs[k % (size_t)n_queues]) != hipSuccess) failed = 1;
            }
        }, ctx->stage_pool.get());
        for (int q = 0; q < n_queues; q++) {
            const hipError_t e = hipStreamSynchronize(cs[q]);
            if (err == hipSuccess) err = e;
        }
        if (err == hipSuccess && failed) err = hipErrorUnknown;
    }
    if (err == hipSuccess) err = hipMemset((uint8_t*)sl->d + len, 0, need - len);
    if (err != hipSuccess) return hip_to_rc(err);
    if (stage_timing)
        fprintf(stderr, "fabgpu arena stage: %.1f MB in %.2f ms\n", len / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - stage_t0).count());
    sl->len = len;
    uint64_t t;
    {
        std::lock_guard<std::mutex> lk(ctx->smu);
        t = ++ctx->stage_seq;
    }
    sl->token.store(t);
    *token = t;
    if (keep && keep_idx >= 0) {
        keep->ctx = ctx;
        keep->idx = keep_idx;
        keep->p = keep_pin;
        keep->len = len;
        keep_back.idx = -1;                                // (the caller's now: host_copy_release)
    }
    return FABGPU_OK;
}

extern "C" {

int fabgpu_identity_verify_batch(fabgpu_ctx* ctx, const fabgpu_identity_batch* b) {
    if (!ctx || !b) return FABGPU_EINVAL;
    const size_t n = b->n;
    if (n == 0) return FABGPU_OK;
    const bool keyed = b->key_id != nullptr;
    const uint32_t m = (b->n_prefixes != 0 && b->pre_idx != nullptr) ? b->n_prefixes : 0;
    if (!b->off || !b->r || !b->s || !b->verdict_bits || (!keyed && (!b->qx || !b->qy)) || (m && !b->pre_off)) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 160) return FABGPU_ETOOBIG;
    const uint8_t* arena = (const uint8_t*)b->arena;
    const bool spans = (b->flags & FABGPU_IDB_SPANS) != 0;
    const bool staged = (b->flags & FABGPU_IDB_ARENA_STAGED) != 0;
    if (b->flags & ~(uint32_t)(FABGPU_IDB_SPANS | FABGPU_IDB_ARENA_STAGED)) return FABGPU_EINVAL;
    const size_t nn = b->n_nym;            // pseudonym signatures over the same arena, on the second stream
    if (nn && (!spans || !b->nym_off || !b->nym_issuer || !b->nym_fields || !b->nym_verdict_bits)) return FABGPU_EINVAL;
    if (nn > 0x7FFFFFF0ull / 200) return FABGPU_ETOOBIG;
    // a staged arena: its slot stays locked - the stager kept out of it - until this batch has run
    fabgpu_ctx::Staged* sl = nullptr;
    std::unique_lock<std::mutex> slk;
    if (staged) {
        if (b->stage_token == 0) return FABGPU_EINVAL;
        for (auto& c : ctx->staged_slots)
            if (c.token.load() == b->stage_token) sl = &c;
        if (!sl) return FABGPU_EINVAL;                                                          // replaced meanwhile
        slk = std::unique_lock<std::mutex>(sl->m);
        if (sl->token.load() != b->stage_token || sl->len == 0) return FABGPU_EINVAL;           // ... or just now
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    // the span of the arena that messages and prefixes reference; spans at or beyond tail_base address the tail
    const size_t noff = spans ? 2 * n : n + 1, npre = m ? (spans ? 2 * (size_t)m : (size_t)m + 1) : 0;
    const bool has_tail = b->tail != nullptr && b->tail_len != 0;
    const uint32_t tbase = has_tail ? b->tail_base : 0xFFFFFFFFu;
    if (has_tail && ((tbase & 63u) || !spans || (uint64_t)tbase + b->tail_len > 0xFFFFFFF0ull)) return FABGPU_EINVAL;
    uint32_t lo = 0xFFFFFFFFu, hi = 0;      // of the caller's arena
    bool tail_used = false, bad = false;
    auto see = [&](uint32_t s0, uint32_t s1) {
        if (s1 < s0) { bad = true; return; }
        if (s1 == s0) return;
        if (s0 >= tbase) {
            if ((uint64_t)s1 > (uint64_t)tbase + b->tail_len) bad = true;
            tail_used = true;
            return;
        }
        if (s1 > tbase) { bad = true; return; }            // straddles
        if (s0 < lo) lo = s0;
        if (s1 > hi) hi = s1;
    };
    for (size_t i = 0; i < n; i++) {
        uint32_t s0 = b->off[spans ? 2 * i : i], s1 = b->off[spans ? 2 * i + 1 : i + 1];
        if (!spans) {                                      // consecutive messages: empty ones still bound the span
            if (s1 < s0) return FABGPU_EINVAL;
            if (s0 < lo) lo = s0;
            if (s1 > hi) hi = s1;
        } else {
            see(s0, s1);
        }
    }
    for (size_t i = 0; i < nn; i++) see(b->nym_off[2 * i], b->nym_off[2 * i + 1]);
    for (uint32_t p = 0; p < m; p++) {
        uint32_t s0 = b->pre_off[spans ? 2 * p : p], s1 = b->pre_off[spans ? 2 * p + 1 : p + 1];
        if (!spans) {
            if (s1 < s0) return FABGPU_EINVAL;
            if (s0 < lo) lo = s0;
            if (s1 > hi) hi = s1;
        } else {
            see(s0, s1);
        }
    }
    const uint32_t ng = b->n_gather;
    if (ng && (!b->gather_spans || !b->gather_digests)) return FABGPU_EINVAL;
    uint64_t gtotal = 0;
    for (size_t j = 0; j < (size_t)ng * 3; j++) {
        uint32_t s0 = b->gather_spans[2 * j], s1 = b->gather_spans[2 * j + 1];
        if (s1 < s0) return FABGPU_EINVAL;
        if (s1 == s0) continue;
        if (s1 > tbase) return FABGPU_EINVAL;              // gathered pieces come from the caller's arena only
        if (s0 < lo) lo = s0;
        if (s1 > hi) hi = s1;
        gtotal += s1 - s0;
    }
    if (bad) return FABGPU_EINVAL;
    if (gtotal > 0x7FFFFFF0ull) return FABGPU_ETOOBIG;
    if (lo == 0xFFFFFFFFu) lo = hi = 0;                    // nothing references the caller's arena
    if (spans && tail_used) lo = 0;                        // keeps tail_base an absolute offset of the device arena
    size_t span = hi >= lo ? (size_t)hi - lo : 0;          // bytes taken from the caller's arena
    if (staged) {
        if (hi > sl->len) return FABGPU_EINVAL;
        if (tail_used && ((size_t)tbase < round_up(sl->len, 64) || (size_t)tbase + b->tail_len + 128 > sl->cap)) return FABGPU_EINVAL;
        lo = 0;                                           // offsets are offsets into the staged bytes
        span = sl->len;
    } else if (span && !arena) {
        return FABGPU_EINVAL;
    }
    // extent of the device arena: the caller's bytes, then (if used) zero padding up to tail_base and the tail
    const size_t extent = tail_used ? (size_t)tbase - lo + b->tail_len : span;
    const size_t fb = n * 32, ib = round_up(n * 4, 64), pob = round_up((npre + 1) * 4, 64), words = (n + 63) / 64;
    const size_t st_off = round_up(words * 8, 64), ab = round_up(extent, 4) + 64;
    const size_t dg_off = round_up(st_off + n, 64);        // out buffer: verdict words | status bytes | digests
    int rc;
    if ((!staged && (rc = ctx->arena.ensure(ab + 64))) || (rc = ctx->offs.ensure(noff * 4)) || (rc = ctx->fields.ensure(4 * fb + ib)) ||
        (rc = ctx->out.ensure(dg_off + (b->digests ? fb : 0))) || (rc = ctx->pre.ensure(pob + ib + (size_t)m * 32 + 64)))
        return rc;
    // small arenas go through the pinned staging buffer; big ones (a marshalled block) are handed to the driver directly -
    // one copy less on the host (the tail padding the kernels may touch is zeroed on the device)
    const bool direct = span >= ((size_t)4 << 20);
    if (!staged && !direct) {
        if (span) memcpy(ctx->arena.h, arena + lo, span);
        memset((uint8_t*)ctx->arena.h + span, 0, ab - span);
        if (tail_used) memcpy((uint8_t*)ctx->arena.h + (tbase - lo), b->tail, b->tail_len);
    }
    if (tail_used && (staged || direct)) {                 // the tail travels through its own pinned staging
        if ((rc = ctx->tailbuf.ensure(b->tail_len))) return rc;
        memcpy(ctx->tailbuf.h, b->tail, b->tail_len);
    }
    uint32_t* ho = (uint32_t*)ctx->offs.h;
    if (spans) {
        for (size_t i = 0; i < n; i++) {                   // an empty span carries no address
            const uint32_t s0 = b->off[2 * i], s1 = b->off[2 * i + 1];
            ho[2 * i] = s1 > s0 ? s0 - lo : 0;
            ho[2 * i + 1] = s1 > s0 ? s1 - lo : 0;
        }
    } else {
        for (size_t i = 0; i < noff; i++) ho[i] = b->off[i] - lo;
    }
    uint8_t* ph = (uint8_t*)ctx->pre.h;
    if (m) {
        uint32_t* po = (uint32_t*)ph;
        if (spans) {
            for (uint32_t p = 0; p < m; p++) {
                const uint32_t s0 = b->pre_off[2 * p], s1 = b->pre_off[2 * p + 1];
                po[2 * p] = s1 > s0 ? s0 - lo : 0;
                po[2 * p + 1] = s1 > s0 ? s1 - lo : 0;
            }
        } else {
            for (size_t p = 0; p < npre; p++) po[p] = b->pre_off[p] - lo;
        }
        memcpy(ph + pob, b->pre_idx, n * 4);
    }
    uint8_t* fh = (uint8_t*)ctx->fields.h;
    if (keyed) {
        memcpy(fh, b->key_id, n * 4);
        memcpy(fh + ib, b->r, fb); memcpy(fh + ib + fb, b->s, fb);
    } else {
        memcpy(fh, b->qx, fb); memcpy(fh + fb, b->qy, fb); memcpy(fh + 2 * fb, b->r, fb); memcpy(fh + 3 * fb, b->s, fb);
    }
    hipError_t err;
    if (staged) {
        err = hipSuccess;                                 // fabgpu_arena_stage put the bytes (and their zero tail) there
        if (tail_used) {
            err = hipMemcpyAsync((uint8_t*)sl->d + tbase, ctx->tailbuf.h, b->tail_len, hipMemcpyHostToDevice, ctx->stream);
            if (err == hipSuccess) err = hipMemsetAsync((uint8_t*)sl->d + tbase + b->tail_len, 0, 128, ctx->stream);
        }
    } else if (direct) {
        err = hipMemcpyAsync(ctx->arena.d, arena + lo, span, hipMemcpyHostToDevice, ctx->stream);
        if (err == hipSuccess) err = hipMemsetAsync((uint8_t*)ctx->arena.d + span, 0, ab - span, ctx->stream);
        if (err == hipSuccess && tail_used)
            err = hipMemcpyAsync((uint8_t*)ctx->arena.d + (tbase - lo), ctx->tailbuf.h, b->tail_len, hipMemcpyHostToDevice, ctx->stream);
    } else {
        err = hipMemcpyAsync(ctx->arena.d, ctx->arena.h, ab, hipMemcpyHostToDevice, ctx->stream);
    }
    if (err == hipSuccess) err = hipMemcpyAsync(ctx->offs.d, ctx->offs.h, noff * 4, hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess && m) err = hipMemcpyAsync(ctx->pre.d, ctx->pre.h, pob + ib, hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(ctx->fields.d, ctx->fields.h, keyed ? ib + 2 * fb : 4 * fb, hipMemcpyHostToDevice, ctx->stream);
    if (err != hipSuccess) return hip_to_rc(err);
    uint8_t* fd = (uint8_t*)ctx->fields.d;
    uint8_t* pd = (uint8_t*)ctx->pre.d;
    uint8_t* dout = (uint8_t*)ctx->out.d;
    fabgpu_identity_batch d = *b;
    d.flags = b->flags & FABGPU_IDB_SPANS;
    d.arena = staged ? sl->d : ctx->arena.d;
    d.arena_bytes = staged ? round_up(tail_used ? (size_t)tbase + b->tail_len : span, 4) + 64 : ab;
    d.tail = nullptr;
    d.tail_base = d.tail_len = 0;
    d.off = (const uint32_t*)ctx->offs.d;
    d.n_prefixes = m;
    d.pre_off = m ? (const uint32_t*)pd : nullptr;
    d.pre_idx = m ? (const uint32_t*)(pd + pob) : nullptr;
    if (keyed) {
        d.key_id = (const uint32_t*)fd;
        d.qx = d.qy = nullptr;
        d.r = fd + ib;
        d.s = fd + ib + fb;
    } else {
        d.qx = fd; d.qy = fd + fb; d.r = fd + 2 * fb; d.s = fd + 3 * fb;
    }
    d.verdict_bits = dout;
    d.status = b->status ? dout + st_off : nullptr;
    d.digests = b->digests ? dout + dg_off : nullptr;
    const size_t gsb = round_up((size_t)ng * 24, 64), gob = round_up(((size_t)ng + 1) * 4, 64), gscr = round_up((size_t)gtotal, 4) + 64;
    if (ng) {
        if ((rc = ctx->gath.ensure(gsb + gob + (size_t)ng * 32))) return rc;
        if (ctx->gscr_cap < gscr) {
            if (ctx->d_gscr) hipFree(ctx->d_gscr);
            ctx->d_gscr = nullptr;
            ctx->gscr_cap = 0;
            if (hipMalloc(&ctx->d_gscr, gscr + gscr / 4) != hipSuccess) return FABGPU_ENOMEM;
            ctx->gscr_cap = gscr + gscr / 4;
        }
        uint32_t* gs = (uint32_t*)ctx->gath.h;
        uint32_t* go = (uint32_t*)((uint8_t*)ctx->gath.h + gsb);
        uint32_t run = 0;
        for (uint32_t j = 0; j < ng; j++) {
            go[j] = run;
            for (int p = 0; p < 3; p++) {
                uint32_t s0 = b->gather_spans[6 * (size_t)j + 2 * p], s1 = b->gather_spans[6 * (size_t)j + 2 * p + 1];
                gs[6 * (size_t)j + 2 * p] = s1 > s0 ? s0 - lo : 0;
                gs[6 * (size_t)j + 2 * p + 1] = s1 > s0 ? s1 - lo : 0;
                run += s1 - s0;
            }
        }
        go[ng] = run;
        err = hipMemcpyAsync(ctx->gath.d, ctx->gath.h, gsb + gob, hipMemcpyHostToDevice, ctx->stream);
        if (err != hipSuccess) return hip_to_rc(err);
        d.gather_spans = (const uint32_t*)ctx->gath.d;
        d.gather_off = (const uint32_t*)((uint8_t*)ctx->gath.d + gsb);
        d.gather_digests = (uint8_t*)ctx->gath.d + gsb + gob;
        d.gather_scratch = ctx->d_gscr;
        d.gather_scratch_bytes = gscr;
    }
    // the pseudonym signatures: staged and launched on the second stream BEFORE the ECDSA kernels are queued, so that the two run
    // side by side (a block's idemix creators are a latency-bound launch of their own: bn_quad29.h)
    const size_t nkb = round_up(nn * 4, 64), nfb = nn * 32, nwords = (nn + 63) / 64, nst_off = round_up(nwords * 8, 64);
    if (nn) {
        if ((rc = ctx->nym.ensure(nkb + 6 * nfb + nn * 8)) || (rc = ctx->nymout.ensure(nst_off + nn))) return rc;
        uint8_t* nh = (uint8_t*)ctx->nym.h;
        memcpy(nh, b->nym_issuer, nn * 4);
        memcpy(nh + nkb, b->nym_fields, 6 * nfb);
        uint32_t* no = (uint32_t*)(nh + nkb + 6 * nfb);
        for (size_t i = 0; i < nn; i++) {
            const uint32_t s0 = b->nym_off[2 * i], s1 = b->nym_off[2 * i + 1];
            no[2 * i] = s1 > s0 ? s0 - lo : 0;
            no[2 * i + 1] = s1 > s0 ? s1 - lo : 0;
        }
        err = hipEventRecord(ctx->ev_up, ctx->stream);                       // arena (and tail) uploads are queued on `stream`
        if (err == hipSuccess) err = hipStreamWaitEvent(ctx->stream2, ctx->ev_up, 0);
        if (err == hipSuccess) err = hipMemcpyAsync(ctx->nym.d, ctx->nym.h, nkb + 6 * nfb + nn * 8, hipMemcpyHostToDevice, ctx->stream2);
        if (err != hipSuccess) return hip_to_rc(err);
        uint8_t* nd = (uint8_t*)ctx->nym.d;
        uint8_t* nout = (uint8_t*)ctx->nymout.d;
        rc = nym_verify_dev(ctx, nn, d.arena, d.arena_bytes, nd + nkb + 6 * nfb, true, nd, nd + nkb, nd + nkb + nfb, nd + nkb + 2 * nfb, nd + nkb + 3 * nfb,
                            nd + nkb + 4 * nfb, nd + nkb + 5 * nfb, nout, b->nym_status ? nout + nst_off : nullptr, ctx->stream2, false);
        if (rc) {
            hipStreamSynchronize(ctx->stream2);
            return rc;
        }
        err = hipMemcpyAsync(ctx->nymout.h, nout, b->nym_status ? nst_off + nn : nwords * 8, hipMemcpyDeviceToHost, ctx->stream2);
        if (err != hipSuccess) {
            hipStreamSynchronize(ctx->stream2);
            return hip_to_rc(err);
        }
    }
    rc = fabgpu_identity_verify_batch_dev(ctx, &d, m ? pd + pob + ib : nullptr, ctx->stream);
    if (rc) {
        if (nn) hipStreamSynchronize(ctx->stream2);                         // nothing of this call may still be running when it returns
        return rc;
    }
    err = hipMemcpyAsync(ctx->out.h, dout, b->digests ? dg_off + fb : (b->status ? st_off + n : words * 8), hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess && ng)
        err = hipMemcpyAsync((uint8_t*)ctx->gath.h + gsb + gob, (uint8_t*)ctx->gath.d + gsb + gob, (size_t)ng * 32, hipMemcpyDeviceToHost, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
    if (nn) {
        hipError_t e2 = hipStreamSynchronize(ctx->stream2);
        if (err == hipSuccess) err = e2;
    }
    if (err != hipSuccess) return hip_to_rc(err);
    memcpy(b->verdict_bits, ctx->out.h, words * 8);
    if (b->status) memcpy(b->status, (uint8_t*)ctx->out.h + st_off, n);
    if (b->digests) memcpy(b->digests, (uint8_t*)ctx->out.h + dg_off, fb);
    if (ng) memcpy(b->gather_digests, (uint8_t*)ctx->gath.h + gsb + gob, (size_t)ng * 32);
    if (nn) {
        memcpy(b->nym_verdict_bits, ctx->nymout.h, nwords * 8);
        if (b->nym_status) memcpy(b->nym_status, (uint8_t*)ctx->nymout.h + nst_off, nn);
    }
    return FABGPU_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// the block pass on the device (block_walk_dev.h)
// ------------------------------------------------------------------------------------------------
namespace fab {

int walk_idtab_set(fabgpu_ctx* ctx, uint32_t n, const DevIdEntry* entries, const uint8_t* bytes, size_t nbytes, uint64_t seed) {
    if (!ctx || (n && (!entries || !bytes))) return FABGPU_EINVAL;
    if (n > (1u << 20) || nbytes > 0x7FFFFFF0ull) return FABGPU_ETOOBIG;
    uint32_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    std::vector<uint32_t> slots(cap, 0);
    for (uint32_t i = 0; i < n; i++) {
        if ((uint64_t)entries[i].off + entries[i].len > nbytes) return FABGPU_EINVAL;
        uint32_t at = (uint32_t)entries[i].hash & (cap - 1), d = 0;
        while (slots[at] && d < bccsp::walk::WALK_ID_PROBE_MAX) {
            at = (at + 1) & (cap - 1);
            d++;
        }
        if (d < bccsp::walk::WALK_ID_PROBE_MAX) slots[at] = i + 1;     // (else: left out - the lookup would not walk that far either)
    }
    const size_t eo = round_up((size_t)cap * 4, 256), bo = round_up(eo + (size_t)n * sizeof(DevIdEntry), 256), total = bo + round_up(nbytes, 64) + 64;
    // The table lives in ONE grow-only allocation that is overwritten in place (it used to be a fresh hipMalloc per version with a
    // hipFree of the old one: a provider whose cache changes with every block - new clients - paid both before every pass).
    std::lock_guard<std::mutex> lk(ctx->mu);             // no pass is in flight while mu is held (a pass synchronises before it returns)
    DeviceGuard g(ctx->device);
    if (ctx->fault == 2) return FABGPU_ENOMEM;
    int rc = ctx->idtab_buf.ensure(total);
    if (rc != FABGPU_OK) return rc;
    void* d = ctx->idtab_buf.d;
    hipError_t err = hipMemcpy(d, slots.data(), (size_t)cap * 4, hipMemcpyHostToDevice);
    if (err == hipSuccess && n) err = hipMemcpy((uint8_t*)d + eo, entries, (size_t)n * sizeof(DevIdEntry), hipMemcpyHostToDevice);
    if (err == hipSuccess && nbytes) err = hipMemcpy((uint8_t*)d + bo, bytes, nbytes, hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        ctx->d_idtab = nullptr;                          // (half-written: the device treats everybody as unknown until the next table)
        ctx->idtab_n = 0;
        ctx->idtab_mask = 0;
        return hip_to_rc(err);
    }
    // The verify launches of a pass are queued on a prediction of "every tuple of this class has a key table" (pred_keyed_*: what held for
    // the previous block).  A table whose every P-256 identity HAS a key table is better evidence than the previous block: the second
    // pass of a fresh provider - identities learned and registered during the first - used to run on the fresh-key kernels once more
    // (2.9 ms instead of 2.0; profiles/r05_fresh_provider_probe.txt).  A wrong "keyed" guess only costs the relaunch of that class.
    {
        bool any = false, all = true;
        for (uint32_t i = 0; i < n; i++)
            if (entries[i].p256) {
                any = true;
                all = all && entries[i].key_id >= 0;
            }
        if (any && all) ctx->pred_keyed_creators = ctx->pred_keyed_others = true;
    }
    ctx->d_idtab = d;
    ctx->idtab_n = n;
    ctx->idtab_mask = cap - 1;
    ctx->idtab_seed = seed;
    ctx->idtab_entries_off = eo;
    ctx->idtab_bytes_off = bo;
    return FABGPU_OK;
}

// TEST HOOK: the wavefront signature gate of the device walk over n signatures held in host memory
int walk_gate_probe(fabgpu_ctx* ctx, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* r, uint8_t* s_out) {
    if (!ctx || (n && (!arena || !spans || !code || !r || !s_out))) return FABGPU_EINVAL;
    if (n == 0) return FABGPU_OK;
    for (uint32_t i = 0; i < n; i++)
        if (spans[2 * i + 1] < spans[2 * i] || spans[2 * i + 1] > arena_len) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    void *da = nullptr, *ds = nullptr, *dout = nullptr;
    int rc = FABGPU_ENOMEM;
    if (hipMalloc(&da, arena_len + 256) == hipSuccess && hipMalloc(&ds, (size_t)n * 8) == hipSuccess && hipMalloc(&dout, (size_t)n * 65) == hipSuccess) {
        hipError_t e = hipMemcpy(da, arena, arena_len, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ds, spans, (size_t)n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_walk_gate_probe(n, da, ds, (uint8_t*)dout + (size_t)n * 64, dout, (uint8_t*)dout + (size_t)n * 32, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(r, dout, (size_t)n * 32, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(s_out, (uint8_t*)dout + (size_t)n * 32, (size_t)n * 32, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(code, (uint8_t*)dout + (size_t)n * 64, n, hipMemcpyDeviceToHost);
        rc = hip_to_rc(e);
    }
    if (da) hipFree(da);
    if (ds) hipFree(ds);
    if (dout) hipFree(dout);
    return rc;
}

// TEST HOOK: the device's identity decoder (certificate -> P-256 key) over n identities held in host memory
int walk_idfix_probe(fabgpu_ctx* ctx, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* key) {
    if (!ctx || (n && (!arena || !spans || !code || !key))) return FABGPU_EINVAL;
    if (n == 0) return FABGPU_OK;
    for (uint32_t i = 0; i < n; i++)
        if (spans[2 * i + 1] < spans[2 * i] || spans[2 * i + 1] > arena_len) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    void *da = nullptr, *ds = nullptr, *dout = nullptr;
    int rc = FABGPU_ENOMEM;
    if (hipMalloc(&da, arena_len + 256) == hipSuccess && hipMalloc(&ds, (size_t)n * 8) == hipSuccess && hipMalloc(&dout, (size_t)n * 65) == hipSuccess) {
        hipError_t e = hipMemcpy(da, arena, arena_len, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ds, spans, (size_t)n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_walk_idfix_probe(n, da, ds, (uint8_t*)dout + (size_t)n * 64, dout, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(key, dout, (size_t)n * 64, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(code, (uint8_t*)dout + (size_t)n * 64, n, hipMemcpyDeviceToHost);
        rc = hip_to_rc(e);
    }
    if (da) hipFree(da);
    if (ds) hipFree(ds);
    if (dout) hipFree(dout);
    return rc;
}

namespace {
// The host's side of a WalkHostOut flag: poll the mapped word until the kernel has stored `seq` (everything it wrote before is then
// visible).  Every ~50 us the stream is asked whether it has drained: a drained stream without the flag is a launch that failed.
int wait_host_flag(const uint32_t* flag, uint32_t seq, hipStream_t st) {
    for (uint32_t spin = 1;; spin++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return FABGPU_OK;
        if ((spin & 0x3FFu) == 0) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq ? FABGPU_OK : FABGPU_ELAUNCH;
            if (q != hipErrorNotReady) return hip_to_rc(q);
        }
        __builtin_ia32_pause();
    }
}
}  // namespace

int walk_block_pass(fabgpu_ctx* ctx, WalkRequest& rq) {
    if (!ctx || !rq.sizes || !rq.env_spans || rq.stage_token == 0) return FABGPU_EINVAL;
    if (rq.n_block_sigs && !rq.block_sigs) return FABGPU_EINVAL;
    const bool has_tail = rq.tail != nullptr && rq.tail_len != 0;
    if (has_tail && ((rq.tail_base & 63u) || (uint64_t)rq.tail_base + rq.tail_len > 0xFFFFFFF0ull)) return FABGPU_EINVAL;
    auto decline = [&](const char* why) {
        rq.declined_why = why;
        return WALK_DECLINED;
    };
    if (rq.n_env == 0) return decline("no envelopes");
    if (rq.n_env > 0x7FFFFFF0u / 64) return FABGPU_ETOOBIG;
    // the staged block: its slot stays locked - the stager kept out of it - until this pass has run
    fabgpu_ctx::Staged* sl = nullptr;
    for (auto& c : ctx->staged_slots)
        if (c.token.load() == rq.stage_token) sl = &c;
    if (!sl) return decline("the staged block was replaced");
    std::unique_lock<std::mutex> slk(sl->m);
    if (sl->token.load() != rq.stage_token || sl->len == 0 || sl->len != rq.block_len) return decline("the staged block was replaced");
    if (has_tail && ((size_t)rq.tail_base < round_up(sl->len, 64) || (size_t)rq.tail_base + rq.tail_len + 128 > sl->cap)) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->fault) return ctx->fault == 2 ? FABGPU_ENOMEM : FABGPU_ELAUNCH;
    DeviceGuard g(ctx->device);
    hipStream_t st = ctx->stream;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const auto t_start = now();
    // (developer aid: FABGPU_WALK_TRACE=1 prints where the HOST side of a pass spends its time, checkpoint by checkpoint)
    static const bool walk_trace = getenv("FABGPU_WALK_TRACE") != nullptr;
    auto mark = [&](const char* what) {
        if (walk_trace) fprintf(stderr, "fabgpu walk %p %-28s %8.3f ms\n", (void*)&rq, what, ms_since(t_start));
    };
    const uint32_t ne = rq.n_env;
    // ---- per-envelope arrays ----
    size_t o = 0;
    auto carve = [&](size_t bytes) { size_t at = o; o = round_up(o + bytes, 256); return at; };
    // (what the host sends up sits together - envelope spans, payload spans, idemix MSPs and, when the host counted, the count kernel's
    //  and the scan's arrays - mirrored at the same offsets in the pinned staging buffer: ONE copy)
    const bool host_counted = rq.host_counts && rq.host_tx_type && rq.host_tx_understood;
    const uint32_t n_msps = rq.idemix_msps && !rq.walk_only ? std::min(rq.n_idemix_msps, WALK_IDEMIX_MSPS_MAX) : 0u;
    // The creators' messages are whole envelope payloads - the longest hashes of a block, a serial chain per message, and for a
    // block of a few hundred transactions THE critical path (300 tx: the chain is 240 us of a 600 us device phase).  With the host's
    // outline of where they are they start before anything is walked, beside the walk's two runs and the gates.
    const bool early_hash = rq.payload_spans && !rq.walk_only && ctx->allow_pair && (uint64_t)ne * 2 <= 65536u;
    const size_t o_env = carve((size_t)ne * 8), o_pay = carve(early_hash ? (size_t)ne * 8 : 0), o_msps = carve(sizeof(DevIdemixMsp) * n_msps),
                 o_ihash = carve(rq.idemix_issuer_hashes ? (size_t)32 * n_msps : 0),
                 o_up_small = o,                                                // ... the copy ends here unless the host counted
                 o_cnt = carve((size_t)ne * 16), o_base = carve((size_t)ne * 16), o_cbase = carve((size_t)ne * 4), o_type = carve(ne), o_und = carve(ne),
                 o_up_all = o,
                 o_tot = carve(sizeof(WalkTotals)), o_mask = carve((size_t)ne * 4), o_flags = carve(ne), o_sum = carve(sizeof(WalkSummary)),
                 o_learn = carve(sizeof(WalkLearn) * WALK_LEARN_SLOTS), o_done = carve(64), o_denv = carve(early_hash ? (size_t)ne * 32 : 0),
                 o_stash = carve(host_counted ? 0 : (size_t)ne * sizeof(bccsp::walk::EnvStash));   // what the count kernel keeps for the emit kernel
    int rc;
    if ((rc = ctx->walk_env.ensure(o))) return rc;
    // pinned staging: the region above as it goes up (the result arrays are sized further down)
    const size_t up_bytes = host_counted ? o_up_all : o_up_small, p_first = round_up(o_up_all, 256);
    if ((rc = ctx->walk_pin.ensure(p_first))) return rc;
    // host-mapped results: [0, 64) the totals' flag, [64, 128) the totals, [128, 192) the final flag, [192, 256) the summary; the arrays follow
    constexpr size_t m_totflag = 0, m_tot = 64, m_finflag = 128, m_sum = 192, m_mtot = 256, m_arrays = 320;   // (the summary is 48 bytes, the memo's totals 24)
    if ((rc = ctx->walk_map.ensure(m_arrays + sizeof(WalkLearn) * WALK_LEARN_SLOTS + 3 * round_up(ne, 64) + ((size_t)8 << 10)))) return rc;
    if (++ctx->walk_seq == 0) ++ctx->walk_seq;
    const uint32_t seq_tot = ctx->walk_seq;
    uint8_t* de = (uint8_t*)ctx->walk_env.d;
    WalkArrays a;
    a.block = (const uint8_t*)sl->d;
    a.block_len = (uint32_t)sl->len;
    a.arena_len = has_tail ? rq.tail_base + rq.tail_len : (uint32_t)sl->len;
    a.env_spans = (const uint32_t*)(de + o_env);
    a.n_env = ne;
    a.counts = (uint4*)(de + o_cnt);
    a.stash = host_counted ? nullptr : (bccsp::walk::EnvStash*)(de + o_stash);
    a.bases = (uint4*)(de + o_base);
    a.cbase = (uint32_t*)(de + o_cbase);
    a.totals = (WalkTotals*)(de + o_tot);
    a.tx_type = de + o_type;
    a.tx_understood = de + o_und;
    a.tx_mask = (uint32_t*)(de + o_mask);
    a.tx_flags = de + o_flags;
    a.summary = (WalkSummary*)(de + o_sum);
    a.learn = (WalkLearn*)(de + o_learn);
    uint8_t* up = (uint8_t*)ctx->walk_pin.h;
    memcpy(up + o_env, rq.env_spans, (size_t)ne * 8);
    if (early_hash) memcpy(up + o_pay, rq.payload_spans, (size_t)ne * 8);
    if (n_msps) memcpy(up + o_msps, rq.idemix_msps, sizeof(DevIdemixMsp) * n_msps);
    if (n_msps && rq.idemix_issuer_hashes) memcpy(up + o_ihash, rq.idemix_issuer_hashes, (size_t)32 * n_msps);
    WalkTotals host_tot = {};
    if (host_counted) {
        // the scan, here: exclusive prefix sums per envelope (walk_scan_kernel's outputs), and the totals the host would otherwise wait for
        memcpy(up + o_cnt, rq.host_counts, (size_t)ne * 16);
        memcpy(up + o_type, rq.host_tx_type, ne);
        memcpy(up + o_und, rq.host_tx_understood, ne);
        uint32_t* bases = (uint32_t*)(up + o_base);
        uint32_t* cb = (uint32_t*)(up + o_cbase);
        uint64_t bt = 0, bp = 0, bc = 0, bg = 0;
        uint32_t bk = 0;
        for (uint32_t e = 0; e < ne; e++) {
            const uint32_t* c = rq.host_counts + 4 * (size_t)e;
            bases[4 * (size_t)e] = (uint32_t)bt; bases[4 * (size_t)e + 1] = (uint32_t)bp; bases[4 * (size_t)e + 2] = (uint32_t)bc; bases[4 * (size_t)e + 3] = (uint32_t)bg;
            cb[e] = bk;
            bt += c[0]; bp += c[1]; bc += c[2]; bg += c[3];
            bk += c[0] ? 1u : 0u;
        }
        if (bt > 0x7FFFFFF0ull || bp > 0x7FFFFFF0ull || bc > 0x7FFFFFF0ull) return FABGPU_ETOOBIG;
        host_tot.tuples = (uint32_t)bt; host_tot.prefixes = (uint32_t)bp; host_tot.checks = (uint32_t)bc; host_tot.creators = bk;
        host_tot.gather_bytes = bg;
    }
    hipError_t err = hipMemcpyAsync(de + o_env, up + o_env, up_bytes, hipMemcpyHostToDevice, st);
    bool s2_busy = false;
    struct Drain2 {                                                         // (an early exit must not leave the hashes running)
        hipStream_t s;
        bool* busy;
        bool armed = true;
        ~Drain2() { if (armed && *busy) hipStreamSynchronize(s); }
    } drain2{ctx->stream2, &s2_busy};
    if (err == hipSuccess && early_hash) {
        a.payload_spans = (const uint32_t*)(de + o_pay);
        a.digest_env = de + o_denv;
        err = hipEventRecord(ctx->ev_w[3], st);                            // "the span lists are on the device"
        if (err == hipSuccess) err = hipStreamWaitEvent(ctx->stream2, ctx->ev_w[3], 0);
        if (err == hipSuccess) {
            s2_busy = true;
            // (thousands of them: on CUs of their own, like every long-running launch of a big block)
            // (up to SHA_COOP_MAX of them: eight lanes on a message, the workgroups spread by what a block of ne transactions with four
            //  signatures each will run beside them - `pre` and the endorsements' hashes, nt / 8 workgroups each, the hash checks' ne / 4)
            err = launch_sha256_spans(ne, sl->d, round_up(sl->len, 4) + 64, a.payload_spans, a.digest_env, ctx->stream2, ne > 2048 ? 84u << 10 : 0u,
                                      spread_waves_per_cu(ne / 8 + ne + ne / 4 + 3));
        }
    }
    if (err == hipSuccess && has_tail) {
        if ((rc = ctx->tailbuf.ensure(rq.tail_len))) return rc;
        memcpy(ctx->tailbuf.h, rq.tail, rq.tail_len);
        err = hipMemcpyAsync((uint8_t*)sl->d + rq.tail_base, ctx->tailbuf.h, rq.tail_len, hipMemcpyHostToDevice, st);
        if (err == hipSuccess) err = hipMemsetAsync((uint8_t*)sl->d + rq.tail_base + rq.tail_len, 0, 128, st);
    }
    if (err == hipSuccess) err = hipMemsetAsync(de + o_mask, 0, (size_t)ne * 4, st);
    if (err == hipSuccess) err = hipMemsetAsync(de + o_sum, 0, sizeof(WalkSummary), st);
    if (err == hipSuccess) err = hipMemsetAsync(de + o_learn, 0, sizeof(WalkLearn) * WALK_LEARN_SLOTS, st);
    if (err == hipSuccess) err = hipMemsetAsync(de + o_done, 0, 64, st);
    if (!host_counted) {
        uint8_t* mh = (uint8_t*)ctx->walk_map.h;
        void* md = nullptr;
        if (err == hipSuccess) err = hipHostGetDevicePointer(&md, mh, 0);
        if (err == hipSuccess) err = launch_walk_count(a, (WalkTotals*)((uint8_t*)md + m_tot), (uint32_t*)((uint8_t*)md + m_totflag), seq_tot, st);
        if (err != hipSuccess) return hip_to_rc(err);
        // the one thing the host must know before it can go on - how many tuples, prefixes, checks - arrives in mapped memory
        if ((rc = wait_host_flag((const uint32_t*)(mh + m_totflag), seq_tot, st))) return rc;
    }
    if (err != hipSuccess) return hip_to_rc(err);
    const WalkTotals tot = host_counted ? host_tot : *(const WalkTotals*)((uint8_t*)ctx->walk_map.h + m_tot);
    if (tot.gather_bytes > 0x7FFFFFF0ull) return decline("gathered hash inputs exceed 2 GiB");
    const uint64_t nt64 = (uint64_t)tot.tuples + rq.n_block_sigs;
    if (nt64 > 0x7FFFFFF0ull / 160) return FABGPU_ETOOBIG;
    const uint32_t nt = (uint32_t)nt64, np = tot.prefixes, nc = tot.checks;
    if (nt == 0) return decline("no signature in the block");
    WalkCounts cnts;
    cnts.n_tx = ne; cnts.n_tuples = nt; cnts.n_prefixes = np; cnts.n_checks = nc; cnts.n_creators = tot.creators;
    WalkOut out;
    mark("totals known");
    if (!rq.sizes(rq.user, cnts, out)) return FABGPU_ETOOBIG;
    mark("sizes callback");
    // ---- per-tuple arrays ----
    o = 0;
    const size_t words = (nt + 63) / 64;
    const size_t o_tup = carve((size_t)nt * sizeof(bccsp::BlockTuple)), o_pre = carve(((size_t)np + 1) * 8), o_chk = carve(((size_t)nc + 1) * sizeof(bccsp::BlockHashCheck)),
                 o_gsp = carve(((size_t)nc + 1) * 24), o_gof = carve(((size_t)nc + 1) * 4), o_gdg = carve(((size_t)nc + 1) * 32), o_idx = carve((size_t)nt * 4),
                 o_off = carve((size_t)nt * 8), o_pix = carve((size_t)nt * 4), o_kid = carve((size_t)nt * 4), o_qx = carve((size_t)nt * 32),
                 o_qy = carve((size_t)nt * 32), o_r = carve((size_t)nt * 32), o_s = carve((size_t)nt * 32), o_gst = carve(nt), o_bits = carve(words * 8),
                 o_dst = carve(nt), o_tst = carve(nt), o_hsh = carve(nt), o_dig = carve((size_t)nt * 32), o_mid = carve(((size_t)np + 1) * 32),
                 o_row = carve((size_t)nt * 4), o_bitc = carve(words * 8), o_tdig = carve((size_t)nt * 32), o_cspan = carve(((size_t)tot.creators + 1) * 8), o_tfl = carve(nt),
                 o_nymf = carve(n_msps ? (size_t)tot.creators * 192 : 0), o_nymi = carve(n_msps ? (size_t)tot.creators * 4 : 0),
                 o_nymsp = carve(n_msps ? (size_t)tot.creators * 8 : 0), o_nymio = carve(n_msps ? (size_t)tot.creators * 4 : 0),
                 o_nymb = carve(n_msps ? ((size_t)tot.creators + 63) / 64 * 8 + 8 : 0), o_nymst = carve(n_msps ? (size_t)tot.creators + 64 : 0),
                 o_nymga = carve(n_msps ? (size_t)tot.creators * 4 + 256 : 0), o_nymsl = carve(n_msps ? (size_t)tot.creators * 4 : 0),
                 o_tqxy = carve(out.tuple_qxy ? (size_t)nt * 64 : 0), o_sparts = carve(((size_t)nt + 255) / 256 * sizeof(WalkSummary)),
                 o_wide = carve(nt <= (uint32_t)WIDE_LAUNCH_MAX ? (size_t)nt * WIDE_SCRATCH_BYTES : 0),
                 o_bita = carve(nt <= (uint32_t)WIDE_LAUNCH_MAX ? words * 8 : 0);                        // ... and the one bitmap of their second phase   // w and u2 Q of every row between the two phases of the wide kernels
    // the verdict memo, if the caller gave room for it (WalkOut::memo_*): built behind the status kernel, copied straight into that room
    const bool memo = out.memo_slots && out.memo_key_off && out.memo_keys && out.memo_status && out.memo_digests && out.memo_slot_cap >= 16 &&
                      (out.memo_slot_cap & (out.memo_slot_cap - 1)) == 0 && out.memo_slot_cap >= 2 * (uint64_t)nt && out.memo_keys_cap != 0 &&
                      out.memo_keys_cap < 0xFFFFFFF0ull;
    // (on the device the keys have room for the longest signatures the memo takes - 1 024 bytes each; what is copied ahead is the room
    //  the caller gave, sized for ordinary signatures; a block that needs more gets the rest through WalkRequest::memo_grow at the end)
    const size_t dev_keys_cap = memo ? (size_t)std::min<uint64_t>(0xFFFFFFE0ull, std::max<uint64_t>(out.memo_keys_cap, (uint64_t)nt * (141 + 1024) + 256)) : 0;
    const size_t o_ment = carve(memo ? (size_t)nt * 4 : 0), o_mslots = carve(memo ? (size_t)out.memo_slot_cap * 4 : 0),
                 o_mkoff = carve(memo ? ((size_t)nt + 1) * 4 : 0), o_mkeys = carve(memo ? dev_keys_cap : 0), o_mst = carve(memo ? nt : 0),
                 o_mtot = carve(memo ? sizeof(WalkMemoTotals) : 0), o_mdig = carve(memo ? (size_t)nt * 32 : 0);
    // ... and the digest memo's index beside it (message spans by entry + a second slot table), if the caller gave room for that too
    const bool hmemo = memo && out.memo_hspans && out.memo_hslots;
    const size_t o_mhsp = carve(hmemo ? (size_t)nt * 16 : 0), o_mhsl = carve(hmemo ? (size_t)out.memo_slot_cap * 4 : 0);
    const size_t o_mtiles = carve(memo ? ((size_t)nt / 2048 + 2) * 24 : 0);                     // the entry scan's tiles (walk_memo_tile_*_kernel)
    if ((rc = ctx->walk_tup.ensure(o))) return rc;
    uint8_t* dt = (uint8_t*)ctx->walk_tup.d;
    a.tuples = (bccsp::BlockTuple*)(dt + o_tup);
    a.n_tuples = nt;
    a.pre_off2 = (uint32_t*)(dt + o_pre);
    a.checks = (bccsp::BlockHashCheck*)(dt + o_chk);
    a.gather_spans = (uint32_t*)(dt + o_gsp);
    a.gather_off = (uint32_t*)(dt + o_gof);
    a.gather_digests = dt + o_gdg;
    a.id_idx = (uint32_t*)(dt + o_idx);
    a.off2 = (uint32_t*)(dt + o_off);
    a.pre_idx = (uint32_t*)(dt + o_pix);
    a.key_id = (uint32_t*)(dt + o_kid);
    a.qx = dt + o_qx; a.qy = dt + o_qy; a.r = dt + o_r; a.s = dt + o_s;
    a.gate_st = dt + o_gst;
    a.tflags = dt + o_tfl;
    const bool has_nym_rows = n_msps != 0 && tot.creators != 0;
    if (has_nym_rows) {
        a.idemix_msps = (const DevIdemixMsp*)(de + o_msps);
        a.n_idemix_msps = n_msps;
        a.nym_fields = dt + o_nymf;
        a.nym_issuer = (uint32_t*)(dt + o_nymi);
        a.nym_issuer_out = (int32_t*)(dt + o_nymio);
        a.nym_spans = (uint32_t*)(dt + o_nymsp);
    }
    a.row_of = (uint32_t*)(dt + o_row);
    a.creator_spans = (uint32_t*)(dt + o_cspan);
    a.n_dev_tuples = tot.tuples;
    a.n_creators = tot.creators;
    // Split the submission (WalkArrays::row_of) when one launch would have to run one lane per signature (more than VERIFY_PAIR_MAX
    // tuples) although creators on two lanes + everybody else on one still fit the chip's 65 536 lanes: the creators hash the longest
    // messages (whole payloads), so they get the shorter arithmetic.
    a.split = (ctx->allow_pair && nt > (uint32_t)VERIFY_PAIR_MAX && tot.creators != 0 && tot.creators <= (uint32_t)VERIFY_PAIR_MAX &&
               (uint64_t)2 * tot.creators + (nt - tot.creators) <= 65536u) ? 1u : 0u;
    // Smaller blocks split too - both launches with two lanes per signature, sharing the chip: what counts there is that the creators'
    // 40-block hashes start behind the emit kernel instead of inside the one fused launch (300 tx: 0.92 -> 0.82 ms, 1 000 tx:
    // 1.09 -> 0.99 ms; round-2 probe gpu_dw_small.sh, since removed)
    bool both_pair = false;
    if (!a.split && ctx->allow_pair && tot.creators != 0 && nt > tot.creators && (uint64_t)2 * nt <= 65536u) {
        a.split = 1;
        both_pair = true;
    }
    const bool exclusive = a.split && !both_pair;
    a.verdict_bits = (const uint64_t*)(dt + o_bits);
    a.verdict_bits_c = (const uint64_t*)(dt + o_bitc);
    a.row_digests = (out.tuple_digest || memo) ? dt + o_dig : nullptr;    // (the memo's keys hold the digests)
    a.tuple_digests = dt + o_tdig;
    a.tuple_qxy = out.tuple_qxy ? dt + o_tqxy : nullptr;
    a.summary_parts = (uint32_t*)(dt + o_sparts);
    if (memo) {
        a.memo_ent = (uint32_t*)(dt + o_ment);
        a.memo_slots = (uint32_t*)(dt + o_mslots);
        a.memo_mask = out.memo_slot_cap - 1;
        a.memo_key_off = (uint32_t*)(dt + o_mkoff);
        a.memo_keys = dt + o_mkeys;
        a.memo_keys_cap = (uint32_t)dev_keys_cap;
        a.memo_status = dt + o_mst;
        a.memo_totals = (WalkMemoTotals*)(dt + o_mtot);
        a.memo_digests = dt + o_mdig;
        a.memo_tiles = dt + o_mtiles;
        if (hmemo) {
            a.memo_hspans = (uint32_t*)(dt + o_mhsp);
            a.memo_hslots = (uint32_t*)(dt + o_mhsl);
        }
        if (n_msps && rq.idemix_issuer_hashes) a.issuer_hashes = de + o_ihash;
    }
    a.dev_status = dt + o_dst;
    a.tuple_status = dt + o_tst;
    a.tuple_hashed = dt + o_hsh;
    if (ctx->d_idtab) {
        a.id_slots = (const uint32_t*)ctx->d_idtab;
        a.id_mask = ctx->idtab_mask;
        a.id_seed = ctx->idtab_seed;
        a.id_entries = (const DevIdEntry*)((uint8_t*)ctx->d_idtab + ctx->idtab_entries_off);
        a.id_bytes = (uint8_t*)ctx->d_idtab + ctx->idtab_bytes_off;
    }
    // pinned room for everything that comes back
    size_t po = p_first;
    auto pin = [&](size_t bytes) { size_t at = po; po = round_up(po + bytes, 64); return at; };
    const size_t p_type = pin(ne), p_und = pin(ne), p_tup = pin((size_t)nt * sizeof(bccsp::BlockTuple)), p_dig = pin((size_t)nt * 32),
                 p_pre = pin(((size_t)np + 1) * 8), p_chk = pin(((size_t)nc + 1) * sizeof(bccsp::BlockHashCheck)),
                 p_sigs = pin((size_t)rq.n_block_sigs * sizeof(bccsp::BlockTuple) + 64), p_qxy = pin(out.tuple_qxy ? (size_t)nt * 64 : 0),
                 p_nymi = pin(out.nym_issuer ? (size_t)tot.creators * 4 : 0);
    {
        // (growing the pinned buffer moves it: nothing above is still needed from the old one)
        if ((rc = ctx->walk_pin.ensure(po))) return rc;
    }
    uint8_t* ph = (uint8_t*)ctx->walk_pin.h;
    // ... and mapped room for what the last kernel writes itself (flags, statuses, identity indices, learn records, summary)
    size_t mo = m_arrays;
    auto mapped = [&](size_t bytes) { size_t at = mo; mo = round_up(mo + bytes, 64); return at; };
    const size_t m_learn = mapped(sizeof(WalkLearn) * WALK_LEARN_SLOTS), m_flags = mapped(ne), m_type = mapped(ne), m_und = mapped(ne), m_tst = mapped(nt),
                 m_hsh = mapped(nt), m_idx = mapped((size_t)nt * 4);
    if ((rc = ctx->walk_map.ensure(mo))) return rc;                        // (may move it: the totals above have been read)
    uint8_t* mh = (uint8_t*)ctx->walk_map.h;
    WalkHostOut ho;
    {
        void* md = nullptr;
        if (hipHostGetDevicePointer(&md, mh, 0) != hipSuccess) return FABGPU_ELAUNCH;
        uint8_t* m8 = (uint8_t*)md;
        ho.flag = (uint32_t*)(m8 + m_finflag);
        ho.done = (uint32_t*)(de + o_done);
        ho.summary = (WalkSummary*)(m8 + m_sum);
        ho.memo_totals = memo ? (WalkMemoTotals*)(m8 + m_mtot) : nullptr;
        ho.learn = (WalkLearn*)(m8 + m_learn);
        ho.tx_flags = m8 + m_flags; ho.tx_type = m8 + m_type; ho.tx_understood = m8 + m_und;
        ho.tuple_status = m8 + m_tst; ho.tuple_hashed = m8 + m_hsh;
        ho.id_idx = (uint32_t*)(m8 + m_idx);
    }
    mark("buffers ensured");
    err = launch_walk_emit(a, tot, st);
    if (err == hipSuccess && rq.n_block_sigs) {
        memcpy(ph + p_sigs, rq.block_sigs, (size_t)rq.n_block_sigs * sizeof(bccsp::BlockTuple));
        err = hipMemcpyAsync(a.tuples + tot.tuples, ph + p_sigs, (size_t)rq.n_block_sigs * sizeof(bccsp::BlockTuple), hipMemcpyHostToDevice, st);
    }
    if (err != hipSuccess) return hip_to_rc(err);
    auto fetch = [&](void* host_dst, size_t pin_off, const void* dev_src, size_t bytes) {
        if (!host_dst || bytes == 0 || err != hipSuccess) return;
        err = hipMemcpyAsync(ph + pin_off, dev_src, bytes, hipMemcpyDeviceToHost, st);
    };
    auto deliver = [&](void* host_dst, size_t pin_off, size_t bytes) {
        if (host_dst && bytes) memcpy(host_dst, ph + pin_off, bytes);
    };
    if (rq.walk_only) {
        fetch(out.tx_type, p_type, a.tx_type, ne);
        fetch(out.tx_understood, p_und, a.tx_understood, ne);
        fetch(out.tuples, p_tup, a.tuples, (size_t)nt * sizeof(bccsp::BlockTuple));
        fetch(out.prefixes, p_pre, a.pre_off2, (size_t)np * 8);
        fetch(out.checks, p_chk, a.checks, (size_t)nc * sizeof(bccsp::BlockHashCheck));
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) return hip_to_rc(err);
        deliver(out.tx_type, p_type, ne);
        deliver(out.tx_understood, p_und, ne);
        deliver(out.tuples, p_tup, (size_t)nt * sizeof(bccsp::BlockTuple));
        deliver(out.checks, p_chk, (size_t)nc * sizeof(bccsp::BlockHashCheck));
        if (out.prefixes)                                          // (start, end) pairs on the device, (offset, length) for the caller
            for (uint32_t p = 0; p < np; p++) {
                const uint32_t s0 = ((const uint32_t*)(ph + p_pre))[2 * p], s1 = ((const uint32_t*)(ph + p_pre))[2 * p + 1];
                out.prefixes[p].off = s0;
                out.prefixes[p].len = s1 - s0;
            }
        rq.ms_walk = ms_since(t_start);
        return FABGPU_OK;
    }
    // From here on three streams work side by side; whatever happens, none of them may still be running when this call returns
    // (the next pass reuses every buffer).
    hipStream_t s2 = ctx->stream2, s3 = ctx->stream3, s4 = ctx->stream4;
    struct Drain {
        hipStream_t a, b, c, d;
        bool armed = true;
        ~Drain() {
            if (!armed) return;
            hipStreamSynchronize(a);
            hipStreamSynchronize(b);
            hipStreamSynchronize(c);
            hipStreamSynchronize(d);
        }
    } drain{s2, s3, s4, st};
    const size_t arena_bytes = round_up(has_tail ? (size_t)rq.tail_base + rq.tail_len : sl->len, 4) + 64;
    ShaPrefixArgs pa;
    pa.spans = true;
    if (np) {
        pa.m = np;
        pa.pre_off = a.pre_off2;
        pa.pre_idx = a.pre_idx;
        pa.mid_scratch = dt + o_mid;
    }
    const size_t gscr = round_up((size_t)tot.gather_bytes, 4) + 64;
    if (nc && ctx->gscr_cap < gscr) {
        if (ctx->d_gscr) hipFree(ctx->d_gscr);
        ctx->d_gscr = nullptr;
        ctx->gscr_cap = 0;
        if (hipMalloc(&ctx->d_gscr, gscr + gscr / 4) != hipSuccess) return FABGPU_ENOMEM;
        ctx->gscr_cap = gscr + gscr / 4;
    }
    // the emitted prefixes / hash checks are all stream2 and stream3 need: mid-states and the TxID / proposal-hash digests run while
    // the main stream looks identities up and gates signatures
    hipStream_t sc = s2;                                                   // the creators' stream
    bool memo_early_pending = false;
    uint32_t nkeys = 0;
    const int32_t** kt = nullptr;
    {
        std::lock_guard<std::mutex> klk(ctx->kmu);
        nkeys = (uint32_t)ctx->ktabs.size();
        kt = ctx->d_ktabs;
    }
    bool keyed_c = ctx->pred_keyed_creators && nkeys != 0, keyed_o = ctx->pred_keyed_others && nkeys != 0;
    if (!a.split) keyed_c = keyed_o = keyed_c && keyed_o;                  // one launch serves both classes
    // A block of a few hundred transactions (what a default network cuts: sampleconfig/configtx.yaml:284 MaxMessageCount 500) cannot
    // fill the chip, and its device phase is a chain of LATENCIES: the creators' payload hashes (76 serial SHA-256 blocks), then a
    // keyed verification (78 000 instructions per wavefront on two lanes).  The wide kernels (p256_wide29.h) put eight lanes on a
    // signature and cut the verification in two: `pre` - s^-1, u2, u2 Q: everything but the digest - runs right behind the gates,
    // BESIDE the hashes; `post` (e w, u1 G, the final addition and comparison: 18 000 instructions) is all that is left behind them.
    const bool wide = ctx->allow_wide && both_pair && keyed_c && keyed_o && nt <= (uint32_t)WIDE_LAUNCH_MAX && !has_nym_rows;
    // The one-wavefront workgroups of a small block's launches - `pre` (nt / 8), the endorsements' hashes (nt / 8), the hash checks'
    // (nc / 8), the creators' early hashes still running - are placed together (kernels.h spread_waves_per_cu).
    const uint32_t spread = spread_waves_per_cu(nt / 8 + nt / 8 + nc / 8 + 3);
    err = hipEventRecord(ctx->ev_w[0], st);
    // (The gates are queued right here, ahead of the side streams' work: on a small block the host's calls, not the kernels, set the pace,
    //  and the gate kernel is the main stream's critical path.)
    if (a.split && early_hash) a.early_creator_hash = 1;
    if (err == hipSuccess && has_nym_rows) {
        // rows of creators that are not idemix stay all-zero; issuer_out -1 = inactive
        err = hipMemsetAsync(dt + o_nymf, 0, (o_nymio - o_nymf), st);
        if (err == hipSuccess) err = hipMemsetAsync(dt + o_nymio, 0xFF, (size_t)tot.creators * 4, st);
    }
    // With idemix creators expected the creators' tuples are gated in a launch of their own, first: the nym launch (a long kernel: the
    // pass's critical path on such a block) waits for that one only.
    const bool creators_first = has_nym_rows && ctx->pred_has_nym;
    if (creators_first) {
        a.gate_mode = 1;
        if (err == hipSuccess) err = launch_walk_gate(a, st);
        if (err == hipSuccess) err = hipEventRecord(ctx->ev_w[5], st);   // "the creators' gates are through"
        a.gate_mode = 2;
        if (err == hipSuccess) err = launch_walk_gate(a, st);
        a.gate_mode = 0;
    } else {
        if (err == hipSuccess) err = launch_walk_gate(a, st);
        if (err == hipSuccess && has_nym_rows) err = hipEventRecord(ctx->ev_w[5], st);   // "the gates are through" (a nym launch waits for it)
    }
    if (err == hipSuccess && memo) {
        // stream4, behind its digests: the EARLY half of the verdict memo - candidates, keys up to the digest, offsets - and its copy into
        // the caller's table, all of it beside the verify launches (block_walk_kernels.hip "the block's verdict memo")
        err = hipEventRecord(ctx->ev_w[7], st);                            // "every gate is through"
        memo_early_pending = true;
    }
    // A block of a few hundred transactions: the endorsements' messages are hashed WHOLE, eight lanes on each (sha256_coop.h) - no
    // mid-state launch in front of them; the mid-states follow on the same stream, off the critical path, for the relaunch below that
    // continues from them should "keyed" have been a wrong guess.
    const bool coop_messages = wide && nt - tot.creators <= SHA_COOP_MAX;
    auto queue_midstates = [&]() {
        // stream3: the mid-states of the shared prefixes (the endorsements' launch continues from them), on CUs of their own: 40
        // workgroups that would otherwise share SIMDs with the 40 000 short-lived wavefronts of the gate kernel and take 4x as long,
        // with the endorsements' launch waiting for them (measured, round-2 probe gpu_dw_sched.sh, since removed: device phase 1.04 -> 0.90 ms)
        ShaPrefixArgs pm = pa;
        pm.lds_reserve = coop_messages ? 0u : 84u << 10;
        if (!coop_messages) err = hipStreamWaitEvent(s3, ctx->ev_w[0], 0);
        if (err == hipSuccess) err = launch_sha256_midstates(sl->d, arena_bytes, pm, s3);
        if (err == hipSuccess) err = hipEventRecord(ctx->ev_w[1], s3);
    };
    auto queue_messages = [&]() {
        // stream3 (behind the mid-states unless the messages are hashed whole): the endorsements' (and block signatures') digests, rows
        // [n_creators, nt) - hash only, beside `pre`
        ShaPrefixArgs ph = pa;
        ph.mid_ready = true;
        ph.digests = dt + o_dig + 32 * (size_t)tot.creators;
        if (np) ph.pre_idx = a.pre_idx + tot.creators;
        err = hipStreamWaitEvent(s3, ctx->ev_w[3], 0);                     // (recorded behind the gates: the emit kernel's event is implied)
        if (err == hipSuccess)
            err = coop_messages ? launch_sha256_messages_coop(nt - tot.creators, sl->d, arena_bytes, a.off2 + 2 * (size_t)tot.creators, ph, s3, spread)
                                : launch_sha256_messages(nt - tot.creators, sl->d, arena_bytes, a.off2 + 2 * (size_t)tot.creators, ph, s3);
        if (err == hipSuccess) err = hipEventRecord(ctx->ev_w[10], s3);
    };
    if (err == hipSuccess && wide) {
        err = hipEventRecord(ctx->ev_w[3], st);                            // the submission arrays are complete (the endorsements' hashes read them)
        // (the order of these calls is the order the kernels start in - the host's calls, 3 us each, set the pace of a small block - and
        //  since the hashes run on eight lanes `pre` is the longer leg of what `post` waits for: 75 us against 62)
        if (err == hipSuccess) err = launch_p256_wide_pre(nt, a.key_id, nkeys, (const void*)kt, a.r, a.s, ctx->d_gtab, dt + o_wide, st, spread);
        if (err == hipSuccess && coop_messages) queue_messages();
    }
    if (err == hipSuccess && a.split) {
        // stream2: the creators' digests into rows [0, n_creators), so that their launch only has the arithmetic left.  Either they
        // were hashed per envelope from the host's outline (early_hash, queued before the walk: a scatter by the scan's creator ranks
        // is all that is left), or they are hashed now, beside the identity lookup and the gates.
        err = hipStreamWaitEvent(sc, ctx->ev_w[0], 0);
        if (err == hipSuccess && early_hash) {
            err = launch_walk_creator_digests(a, dt + o_dig, sc);
        } else if (err == hipSuccess) {
            err = launch_sha256_spans(tot.creators, sl->d, arena_bytes, a.creator_spans, dt + o_dig, sc, exclusive ? 84u << 10 : 0u);
        }
        if (err == hipSuccess && wide) err = hipEventRecord(ctx->ev_w[4], sc);   // "the creators' digests are in their rows" (the one `post` launch waits for it)
    }
    if (err == hipSuccess && np && !coop_messages) queue_midstates();
    if (err == hipSuccess && wide && !coop_messages) queue_messages();
    if (err == hipSuccess && np && coop_messages) queue_midstates();
    if (err == hipSuccess && nc) {       // stream4: the TxID / proposal-hash digests (only the flags at the very end wait for them)
        err = hipStreamWaitEvent(s4, ctx->ev_w[0], 0);
        if (err == hipSuccess) err = launch_gather_sha256(nc, sl->d, arena_bytes, a.gather_spans, a.gather_off, ctx->d_gscr, gscr, dt + o_gdg, s4, 0, spread);
        if (err == hipSuccess) err = hipEventRecord(ctx->ev_w[2], s4);
    }
    mark("side streams queued");
    if (err == hipSuccess && memo_early_pending) {
        if (hmemo) err = hipMemsetAsync(dt + o_mhsl, 0, (size_t)out.memo_slot_cap * 4, s4);   // (ahead of the wait for the gates)
        if (err == hipSuccess) err = hipStreamWaitEvent(s4, ctx->ev_w[7], 0);
        // len, the tiled scan (entry indices, key offsets), then the key bytes on stream4 and - beside them, on stream3 - the digest
        // memo's index: both only need the scan (round 6: one after the other they were 85 us between the gates and the keys' copy)
        if (err == hipSuccess) err = launch_walk_memo_early(a, s4, hmemo ? ctx->ev_w[9] : nullptr);
        if (err == hipSuccess && hmemo) err = hipStreamWaitEvent(s3, ctx->ev_w[9], 0);          // "the scan is through"
        if (err == hipSuccess && hmemo) err = launch_walk_memo_index(a, s3);
        if (err == hipSuccess && hmemo) err = hipEventRecord(ctx->ev_w[11], s3);
        mark("memo early launched");
        if (err == hipSuccess) err = hipMemcpyAsync(out.memo_key_off, dt + o_mkoff, ((size_t)nt + 1) * 4, hipMemcpyDeviceToHost, s4);
        mark("memo key_off copy queued");
        if (err == hipSuccess) err = hipMemcpyAsync(out.memo_keys, dt + o_mkeys, out.memo_keys_cap, hipMemcpyDeviceToHost, s4);
        mark("memo keys copy queued");
        if (err == hipSuccess && hmemo) err = hipStreamWaitEvent(s4, ctx->ev_w[11], 0);
        if (err == hipSuccess && hmemo) err = hipMemcpyAsync(out.memo_hspans, dt + o_mhsp, (size_t)nt * 16, hipMemcpyDeviceToHost, s4);
        if (err == hipSuccess && hmemo) err = hipMemcpyAsync(out.memo_hslots, dt + o_mhsl, (size_t)out.memo_slot_cap * 4, hipMemcpyDeviceToHost, s4);
        if (err == hipSuccess) err = hipEventRecord(ctx->ev_w[8], s4);
    }
    // The block's idemix creators: ONE nym launch over their rows, packed (walk_nym_pack_kernel: 2 000 idemix creators among 10 000 are
    // 125 wavefronts of the four-lane kernel, not 625 - the ECDSA launches beside it keep their SIMDs), on stream3 behind the mid-states.
    // Queued on a prediction like the key tables - the previous block had idemix creators, about so many - and caught up with below if
    // the gates say there are some and nobody launched, or more than the launch had rows for.
    bool nym_ran = false;
    uint32_t nym_cap = 0;
    auto run_nym = [&](uint32_t cap) -> int {
        cap = std::min<uint32_t>(tot.creators, (cap + 63u) & ~63u);
        // The launch's rows take their inputs through the pack kernel's list (row -> creator rank); rows nobody claims stay idle (~0).
        // The status bytes start out as "not decided" so that a row the nym kernel never ran over can not read as valid.  (All of that
        // ahead of the wait for the gates.)
        hipError_t e = hipMemsetAsync(dt + o_nymga, 0xFF, (size_t)cap * 4, s3);
        if (e == hipSuccess) e = hipMemsetAsync(dt + o_nymb, 0, ((size_t)cap + 63) / 64 * 8 + 8, s3);
        if (e == hipSuccess) e = hipMemsetAsync(dt + o_nymst, 6 /* FABGPU_NYM_NEEDS_SW */, (size_t)cap + 64, s3);
        if (e == hipSuccess) e = hipStreamWaitEvent(s3, ctx->ev_w[5], 0);
        a.nym_slot = (uint32_t*)(dt + o_nymsl);
        if (e == hipSuccess) e = launch_walk_nym_pack(a, (uint32_t*)(dt + o_nymga), cap, s3);
        if (e != hipSuccess) return hip_to_rc(e);
        const size_t col = (size_t)32 * tot.creators;
        uint8_t* f = dt + o_nymf;
        int r2 = nym_verify_dev(ctx, cap, sl->d, arena_bytes, dt + o_nymsp, true, dt + o_nymi, f, f + col, f + 2 * col, f + 3 * col, f + 4 * col, f + 5 * col,
                                dt + o_nymb, dt + o_nymst, s3, false, dt + o_nymga,
                                exclusive ? 84u << 10 : 0u);   // on CUs of its own, like the two ECDSA launches (kernels.h)
        if (r2 != FABGPU_OK) return r2;
        e = hipEventRecord(ctx->ev_w[6], s3);                              // (the main stream waits for it in front of the status kernel, not here:
        if (e != hipSuccess) return hip_to_rc(e);                          // the ECDSA launches queued next must run BESIDE the nym kernel)
        a.nym_cap = cap;
        a.nym_status = dt + o_nymst;
        nym_ran = true;
        nym_cap = cap;
        return FABGPU_OK;
    };
    if (has_nym_rows && ctx->pred_has_nym && (rc = run_nym(ctx->pred_nym_rows + ctx->pred_nym_rows / 4 + 64))) return rc;
    // What the gates found decides which kernels SHOULD run per launch class - registered comb tables when every submitted tuple of the
    // class has one, keys carried in the rows otherwise - and whether this pass may answer at all.  Neither is waited for: the launches
    // are queued on a prediction (fabgpu_ctx::pred_keyed_*: what held for the previous block; "fresh keys" is always correct) and the
    // summary is read once, at the very end; a class that was predicted "keyed" and was not is launched again.
    rq.ms_walk = ms_since(t_start);
    const auto t_verify = now();
    pa.mid_ready = true;
    pa.digests = (out.tuple_digest || memo) ? dt + o_dig : nullptr;
    // rows [row0, row0 + n) as one fused (hash + verify) launch on stream `ls`
    auto verify_rows = [&](uint32_t row0, uint32_t n, bool prefixed, bool pair, bool keyed, void* bits, hipStream_t ls) -> int {
        ShaPrefixArgs p = pa;
        if (exclusive) p.lds_reserve = 84u << 10;                          // the two launches of a split submission on disjoint CUs (kernels.h)
        if (!prefixed) {
            p.m = 0;
            p.pre_off = p.pre_idx = nullptr;
            p.mid_scratch = nullptr;
        } else {
            p.pre_idx = a.pre_idx + row0;
        }
        if (p.digests) p.digests = dt + o_dig + 32 * (size_t)row0;
        hipError_t e;
        if (keyed) {
            e = launch_sha256_p256_verify_keyed(n, sl->d, arena_bytes, a.off2 + 2 * (size_t)row0, a.key_id + row0, nkeys, (const void*)kt, a.r + 32 * (size_t)row0,
                                                a.s + 32 * (size_t)row0, ctx->d_gtab, bits, dt + o_dst + row0, pair, p, ls);
        } else {
            size_t wi = 0;
            void* wsp = nullptr;
            int r2 = ctx->acquire_qws(verify_workspace_bytes(n, pair), &wi, &wsp, ls);
            if (r2 != FABGPU_OK) return r2;
            e = launch_sha256_p256_verify(n, sl->d, arena_bytes, a.off2 + 2 * (size_t)row0, a.qx + 32 * (size_t)row0, a.qy + 32 * (size_t)row0,
                                          a.r + 32 * (size_t)row0, a.s + 32 * (size_t)row0, ctx->d_gtab, wsp, bits, dt + o_dst + row0, pair, p, ls);
            ctx->release_qws(wi, ls);
        }
        return hip_to_rc(e);
    };
    // the creators of a split submission: rows [0, n_creators), digests ready (or about to be: same stream), arithmetic only
    auto verify_creators = [&](bool keyed) -> int {
        hipError_t e;
        if (keyed) {
            e = launch_p256_verify_keyed(tot.creators, a.key_id, nkeys, (const void*)kt, dt + o_dig, a.r, a.s, ctx->d_gtab, dt + o_bitc, dt + o_dst, true, s2, exclusive ? 84u << 10 : 0u);
        } else {
            size_t wi = 0;
            void* wsp = nullptr;
            int r2 = ctx->acquire_qws(verify_workspace_bytes(tot.creators, true), &wi, &wsp, s2);
            if (r2 != FABGPU_OK) return r2;
            e = launch_p256_verify(tot.creators, a.qx, a.qy, dt + o_dig, a.r, a.s, ctx->d_gtab, wsp, dt + o_bitc, dt + o_dst, true, s2, exclusive ? 84u << 10 : 0u, ctx->pair_table_lds);
            ctx->release_qws(wi, s2);
        }
        return hip_to_rc(e);
    };
    // the end of a pass on the main stream: statuses and digest comparisons, the big optional arrays as copies (tuple records, digests,
    // keys: only a caller that seeds the memo or wants spans asks for them), then the kernel that writes the small results into mapped
    // memory and raises the flag - behind the copies, so the flag covers them too
    auto finish = [&]() -> int {
        if (++ctx->walk_seq == 0) ++ctx->walk_seq;
        ho.seq = ctx->walk_seq;
        if (err == hipSuccess && nym_ran) err = hipStreamWaitEvent(st, ctx->ev_w[6], 0);     // the nym kernel's answers
        if (err == hipSuccess && memo) err = hipMemsetAsync(dt + o_mslots, 0, (size_t)out.memo_slot_cap * 4, st);
        if (err == hipSuccess && memo) err = hipMemsetAsync(&((WalkMemoTotals*)(dt + o_mtot))->live, 0, 4, st);
        mark("finish: memsets queued");
        if (!memo && !out.tuples && !out.tuple_digest && !out.tuple_qxy && !(has_nym_rows && out.nym_issuer) && walk_small_finish_fits(a, nc)) {
            // a small block, flags only: statuses, digest comparisons and the finish in one launch
            if (err == hipSuccess) err = launch_walk_status_finish_small(a, nc, ho, st);
            mark("finish: last kernel queued");
            if (err != hipSuccess) return hip_to_rc(err);
            int r2 = wait_host_flag((const uint32_t*)(mh + m_finflag), ho.seq, st);
            if (r2 != FABGPU_OK) return r2;
            rq.summary = *(const WalkSummary*)(mh + m_sum);
            return FABGPU_OK;
        }
        if (err == hipSuccess) err = launch_walk_status_checks(a, nc, st);
        if (memo) {
            // the LATE half of the memo: digests, status bytes and slots of the candidates that were hashed and decided
            if (err == hipSuccess) err = launch_walk_memo_late(a, st);
            if (err == hipSuccess) err = hipMemcpyAsync(out.memo_slots, dt + o_mslots, (size_t)out.memo_slot_cap * 4, hipMemcpyDeviceToHost, st);
            if (err == hipSuccess) err = hipMemcpyAsync(out.memo_digests, dt + o_mdig, (size_t)nt * 32, hipMemcpyDeviceToHost, st);
            if (err == hipSuccess) err = hipMemcpyAsync(out.memo_status, dt + o_mst, nt, hipMemcpyDeviceToHost, st);
            mark("finish: late memo copies queued");
        }
        fetch(out.tuples, p_tup, a.tuples, (size_t)nt * sizeof(bccsp::BlockTuple));
        fetch(out.tuple_digest, p_dig, dt + o_tdig, (size_t)nt * 32);
        fetch(out.tuple_qxy, p_qxy, dt + o_tqxy, (size_t)nt * 64);
        if (has_nym_rows) fetch(out.nym_issuer, p_nymi, dt + o_nymio, (size_t)tot.creators * 4);
        if (err == hipSuccess && memo) err = hipStreamWaitEvent(st, ctx->ev_w[8], 0);        // the memo's early half (long done: it ran beside the verify launches)
        if (err == hipSuccess) err = launch_walk_finish(a, ho, st);
        mark("finish: last kernel queued");
        if (err != hipSuccess) return hip_to_rc(err);
        int r2 = wait_host_flag((const uint32_t*)(mh + m_finflag), ho.seq, st);
        if (r2 != FABGPU_OK) return r2;
        rq.summary = *(const WalkSummary*)(mh + m_sum);
        if (memo) {
            const WalkMemoTotals mt = *(const WalkMemoTotals*)(mh + m_mtot);
            rq.memo_n = mt.overflow ? 0 : mt.n;
            rq.memo_live = mt.overflow ? 0 : mt.live;
            rq.memo_bytes = mt.overflow ? 0 : mt.bytes;
            if (rq.memo_n && mt.bytes > out.memo_keys_cap) {
                // more key bytes than the room that was copied ahead (signatures far longer than an ECDSA signature's 72 bytes): the caller
                // makes room, the keys are copied again - whole - and the pass answers a little later; no room: no memo for this block
                uint8_t* big = rq.memo_grow ? rq.memo_grow(rq.user, (size_t)mt.bytes) : nullptr;
                if (!big || hipMemcpy(big, dt + o_mkeys, (size_t)mt.bytes, hipMemcpyDeviceToHost) != hipSuccess) rq.memo_n = rq.memo_live = 0;
            }
        }
        return FABGPU_OK;
    };
    if (np && !wide) err = hipStreamWaitEvent(st, ctx->ev_w[1], 0);      // the mid-states (long done: they ran beside the gates)
    if (err != hipSuccess) return hip_to_rc(err);
    if (wide) {
        // the second phase of every row's verification as ONE launch, on this stream behind `pre`, once both kinds of digests are in
        // their rows (the endorsements' hashes: stream3; the creators' scatter: stream2 - both long done when `pre` ends): one launch
        // and one cross-stream wait fewer in front of the finish than a launch per class had
        err = hipStreamWaitEvent(st, ctx->ev_w[10], 0);
        if (err == hipSuccess) err = hipStreamWaitEvent(st, ctx->ev_w[4], 0);
        a.verdict_bits_all = (const uint64_t*)(dt + o_bita);
        a.all_creators = a.all_others = 1;
        if (err == hipSuccess) err = launch_p256_wide_post(nt, dt + o_dig, a.r, ctx->d_gtab, dt + o_wide, dt + o_bita, dt + o_dst, st, spread);
    } else if (a.split) {
        // creators on stream2 (two lanes per signature), everybody else on the main stream: side by side
        err = hipEventRecord(ctx->ev_w[3], st);                            // the submission arrays are complete
        if (err == hipSuccess) err = hipStreamWaitEvent(s2, ctx->ev_w[3], 0);
        if (err != hipSuccess) return hip_to_rc(err);
        if ((rc = verify_creators(keyed_c))) return rc;
        err = hipEventRecord(ctx->ev_w[4], s2);
        if (err != hipSuccess) return hip_to_rc(err);
        if ((rc = verify_rows(tot.creators, nt - tot.creators, np != 0, both_pair, keyed_o, dt + o_bits, st))) return rc;
        err = hipStreamWaitEvent(st, ctx->ev_w[4], 0);
    } else {
        if ((rc = verify_rows(0, nt, np != 0, ctx->allow_pair, keyed_o, dt + o_bits, st))) return rc;
    }
    if (err == hipSuccess && nc) err = hipStreamWaitEvent(st, ctx->ev_w[2], 0);
    if (err != hipSuccess) return hip_to_rc(err);
    mark("verify launches queued");
    if ((rc = finish())) return rc;
    mark("finish flag seen");
    if (rq.summary.n_outline_differs) return decline("the device's walk of an envelope differs from the host's outline of it (creator message span, or the counts)");
    // (summary.n_undecided - certificates whose key lies beyond the decoder's window - are TUPLE_ST_NEEDS_SW tuples of their own
    //  transactions, not a reason to give the block up)
    if (rq.summary.n_submitted == 0) return decline("no tuple for the device to decide");
    {
        // Was "keyed" a wrong guess for a class?  Then its rows ran against filler tables: launch it again with the keys carried along
        // (a new client's first block, an endorser's first 64 signatures: once per change of regime), and redo the flags.
        const bool unk_c = rq.summary.n_unkeyed_creator != 0, unk_o = rq.summary.n_unkeyed_other != 0;
        const bool redo_c = a.split ? (keyed_c && unk_c) : false, redo_o = a.split ? (keyed_o && unk_o) : (keyed_o && (unk_c || unk_o));
        // idemix creators turned up and nobody had launched for them, or for fewer
        const bool redo_nym = has_nym_rows && rq.summary.n_nym != 0 && (!nym_ran || rq.summary.n_nym > nym_cap);
        ctx->pred_keyed_creators = !unk_c;
        ctx->pred_keyed_others = !unk_o;
        ctx->pred_has_nym = has_nym_rows && rq.summary.n_nym != 0;
        ctx->pred_nym_rows = rq.summary.n_nym;
        if (redo_c || redo_o || redo_nym) {
            rq.relaunched = (redo_c ? 1u : 0u) + (redo_o ? 1u : 0u) + (redo_nym ? 1u : 0u);
            if (redo_nym && (rc = run_nym(rq.summary.n_nym))) return rc;
            if (redo_c) {
                keyed_c = false;
                a.all_creators = 0;                                        // (their verdicts come from their own launch's bitmap now)
                if ((rc = verify_creators(false))) return rc;
                err = hipEventRecord(ctx->ev_w[4], s2);
                if (err == hipSuccess) err = hipStreamWaitEvent(st, ctx->ev_w[4], 0);
                if (err != hipSuccess) return hip_to_rc(err);
            }
            if (redo_o) {
                keyed_o = false;
                a.all_others = 0;
                if (!a.split) keyed_c = false;
                if (wide && np && (err = hipStreamWaitEvent(st, ctx->ev_w[1], 0)) != hipSuccess) return hip_to_rc(err);   // (wide: nobody waited for the mid-states yet)
                if ((rc = a.split ? verify_rows(tot.creators, nt - tot.creators, np != 0, both_pair, false, dt + o_bits, st)
                                  : verify_rows(0, nt, np != 0, ctx->allow_pair, false, dt + o_bits, st)))
                    return rc;
            }
            err = hipMemsetAsync(de + o_mask, 0, (size_t)ne * 4, st);
            if (err == hipSuccess) err = hipMemsetAsync(a.summary, 0, offsetof(WalkSummary, n_learn), st);   // (the status kernel adds them up again)
            if (err == hipSuccess) err = hipMemsetAsync(de + o_done, 0, 64, st);
            if (err != hipSuccess) return hip_to_rc(err);
            if ((rc = finish())) return rc;
        }
    }
    rq.keyed_creators = keyed_c;
    rq.keyed_others = keyed_o;
    rq.ms_verify = ms_since(t_verify);
    auto from_map = [&](void* host_dst, size_t map_off, size_t bytes) {
        if (host_dst && bytes) memcpy(host_dst, mh + map_off, bytes);
    };
    from_map(out.tx_flags, m_flags, ne);
    from_map(out.tx_type, m_type, ne);
    from_map(out.tx_understood, m_und, ne);
    from_map(out.tuple_status, m_tst, nt);
    from_map(out.tuple_hashed, m_hsh, nt);
    from_map(out.id_idx, m_idx, (size_t)nt * 4);
    from_map(rq.learn_out, m_learn, sizeof(WalkLearn) * WALK_LEARN_SLOTS);
    deliver(out.tuples, p_tup, (size_t)nt * sizeof(bccsp::BlockTuple));
    deliver(out.tuple_digest, p_dig, (size_t)nt * 32);
    deliver(out.tuple_qxy, p_qxy, (size_t)nt * 64);
    if (has_nym_rows) deliver(out.nym_issuer, p_nymi, (size_t)tot.creators * 4);
    else if (out.nym_issuer && tot.creators) memset(out.nym_issuer, 0xFF, (size_t)tot.creators * 4);
    drain.armed = false;                                                    // (the flag was raised behind everything: all four streams are idle)
    drain2.armed = false;
    return FABGPU_OK;
}

// What `slots` overlapping passes over blocks of up to block_bytes / n_tx transactions / n_tuples signatures need on this device, made
// NOW (provider construction: ProviderOptions::concurrent_passes) instead of when passes first overlap - the staging slots for uploaded
// blocks, the pinned staging buffer, the pass's device / pinned / host-mapped arrays at generous upper bounds, the gather scratch.  A peer
// that joins a channel gets no untimed rounds: measured in round 3, the first overlapping passes of a provider took 5-16 ms (a second
// and third 63 MB staging slot, the arrays of the first memo-seeding pass).  Best effort: a failed allocation is simply made later.
int walk_preallocate(fabgpu_ctx* ctx, size_t block_bytes, uint32_t n_tx, uint32_t n_tuples, int slots) {
    if (!ctx) return FABGPU_EINVAL;
    if (block_bytes > 0xFFFFFF00ull) return FABGPU_ETOOBIG;
    DeviceGuard g(ctx->device);
    int rc = FABGPU_OK;
    const size_t need = round_up(block_bytes, 64) + 128;
    // ALL staging slots, whatever `slots` says: fabgpu_arena_stage takes the least recently filled free slot, i.e. it walks round all
    // N_STAGED of them even when passes never overlap - with two of three made here, the third pass of a fresh provider paid for the
    // third slot's 57 MB hipMalloc inside its upload (1.5 ms instead of 1.05: profiles/r05_fresh_provider_probe.txt).
    (void)slots;
    // every kernel function the passes will launch, resolved now (kernels.h warm_kernel_functions_*)
    (void)warm_kernel_functions_kernels();
    (void)warm_kernel_functions_wide();
    (void)warm_kernel_functions_idemix();
    (void)warm_kernel_functions_walk();
    (void)warm_kernel_functions_keytab();
    for (int i = 0; i < fabgpu_ctx::N_STAGED; i++) {
        fabgpu_ctx::Staged& sl = ctx->staged_slots[i];
        std::lock_guard<std::mutex> lk(sl.m);
        if (sl.cap >= need + (64 << 10)) continue;
        if (sl.d) hipFree(sl.d);
        sl.d = nullptr;
        sl.cap = 0;
        sl.token.store(0);
        sl.len = 0;
        if (hipMalloc(&sl.d, need + need / 8 + (64 << 10)) != hipSuccess) {
            sl.d = nullptr;
            rc = FABGPU_ENOMEM;
            continue;
        }
        sl.cap = need + need / 8 + (64 << 10);
    }
    {
        std::lock_guard<std::mutex> plk(ctx->stage_pin_mu);
        if (block_bytes >= ((size_t)4 << 20) && ctx->stage_pin.ensure(block_bytes) != FABGPU_OK) rc = FABGPU_ENOMEM;
        if (!ctx->stage_pool) ctx->stage_pool.reset(new (std::nothrow) WorkerPool(4));
    }
    {
        // the device's copy of the identity cache (walk_idtab_set): a fresh provider's first non-empty table arrives before its SECOND
        // pass - its room and its three small synchronous uploads are rehearsed with a one-entry table, then the table is empty again
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            if (ctx->idtab_buf.ensure((size_t)1 << 20) != FABGPU_OK) rc = FABGPU_ENOMEM;
        }
        DevIdEntry e;
        memset(&e, 0, sizeof(e));
        e.len = 4;
        const uint8_t four[4] = {0, 0, 0, 0};
        if (walk_idtab_set(ctx, 1, &e, four, sizeof(four), 0) == FABGPU_OK) (void)walk_idtab_set(ctx, 0, nullptr, nullptr, 0, 0);
    }
    {
        // what the first registrations of a channel's signers need (keytab_kernels.hip): a slab of key tables, the builder's room for
        // sixteen keys, its stream - a provider's FIRST block makes its signers eligible, and hipMalloc inside that pass cost more than the build
        std::lock_guard<std::mutex> bl(ctx->ktab_build_mu);
        std::lock_guard<std::mutex> kl(ctx->kmu);
        if (ctx->ktab_slabs.empty()) {
            int32_t* first = ktab_alloc_locked(ctx);
            if (first) ctx->ktab_free.push_back(first);
            else rc = FABGPU_ENOMEM;
        }
        const size_t in_bytes = (size_t)(64 + sizeof(void*)) * 16 * 2, scr_bytes = keytab_scratch_bytes(16) * 2;
        if (ctx->ktab_in_cap < in_bytes && hipMalloc(&ctx->d_ktab_in, in_bytes) == hipSuccess) ctx->ktab_in_cap = in_bytes;
        if (ctx->ktab_scr_cap < scr_bytes && hipMalloc(&ctx->d_ktab_scr, scr_bytes) == hipSuccess) ctx->ktab_scr_cap = scr_bytes;
        if (!ctx->stream_keytab) (void)hipStreamCreateWithFlags(&ctx->stream_keytab, hipStreamNonBlocking);
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t ne = n_tx, nt = n_tuples;
    // upper bounds of walk_block_pass's carves: per envelope 91 bytes of arrays + the count kernel's record slot, per tuple ~1.9 KB with the memo's key room (1 165 bytes)
    if (ctx->walk_env.ensure(ne * (128 + sizeof(bccsp::walk::EnvStash)) + ((size_t)256 << 10)) != FABGPU_OK) rc = FABGPU_ENOMEM;
    if (ctx->walk_tup.ensure(nt * 2112 + ne * 256 + ((size_t)1 << 20)) != FABGPU_OK) rc = FABGPU_ENOMEM;
    if (ctx->walk_pin.ensure(nt * 256 + ne * 160 + ((size_t)256 << 10)) != FABGPU_OK) rc = FABGPU_ENOMEM;
    if (ctx->walk_map.ensure(nt * 8 + ne * 8 + sizeof(WalkLearn) * WALK_LEARN_SLOTS + ((size_t)64 << 10)) != FABGPU_OK) rc = FABGPU_ENOMEM;
    const size_t gscr = ne * 4096 + ((size_t)64 << 10);                         // TxID + proposal-hash inputs: a few KB per transaction
    if (ctx->gscr_cap < gscr) {
        if (ctx->d_gscr) hipFree(ctx->d_gscr);
        ctx->d_gscr = nullptr;
        ctx->gscr_cap = 0;
        if (hipMalloc(&ctx->d_gscr, gscr) == hipSuccess) ctx->gscr_cap = gscr;
        else rc = FABGPU_ENOMEM;
    }
    if (ctx->tailbuf.ensure((size_t)64 << 10) != FABGPU_OK) rc = FABGPU_ENOMEM;
    // The copy ENGINES, too.  The runtime creates a DMA queue the first time a copy of a direction needs one more of them - ~10 ms each,
    // on the calling thread, inside hipMemcpyAsync: measured (rocprofv3 --memory-copy-trace, profiles/r04_trace_first_memo_passes.txt) as
    // 11 ms of nothing on the device in front of the first memo-seeding passes' key copies (device -> pinned host, 8.5 MB, beside the
    // uploads of the other pass), four passes in a row - what round 3 booked as "pinned memo tables allocated on first overlap".  Copies
    // of both directions on all four pass streams at once, twice, make the runtime create them now.
    {
        const size_t piece = (size_t)4 << 20;
        void *d = nullptr, *h = nullptr;
        if (hipMalloc(&d, 4 * piece) == hipSuccess && hipHostMalloc(&h, 4 * piece, hipHostMallocPortable) == hipSuccess) {
            hipStream_t ss[4] = {ctx->stream, ctx->stream2, ctx->stream3, ctx->stream4};
            for (int rep = 0; rep < 2; rep++) {
                for (int k = 0; k < 4; k++) (void)hipMemcpyAsync((uint8_t*)h + k * piece, (uint8_t*)d + k * piece, piece, hipMemcpyDeviceToHost, ss[k]);
                for (int k = 0; k < 4; k++) (void)hipMemcpyAsync((uint8_t*)d + k * piece, (uint8_t*)h + k * piece, piece, hipMemcpyHostToDevice, ss[3 - k]);
                for (int k = 0; k < 4; k++) (void)hipStreamSynchronize(ss[k]);
            }
        }
        if (d) hipFree(d);
        if (h) hipHostFree(h);
    }
    return rc;
}

// Device -> pinned-host copies into the given buffers on the pass's streams, as a memo-seeding pass issues them; returns the longest
// time one hipMemcpyAsync call kept the calling thread (ms; < 0: a copy failed).  GPUCSP::Preallocate runs this beside an upload until
// the calls return at once: see walk_preallocate on the runtime's lazily created DMA queues.
double walk_warm_copies(fabgpu_ctx* ctx, void* const* pinned, const size_t* bytes, int n) {
    if (!ctx || !pinned || !bytes || n <= 0) return -1;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    if (!ctx->walk_tup.d || !ctx->walk_pin.h || ctx->walk_tup.cap < ((size_t)8 << 20)) return -1;
    hipStream_t ss[4] = {ctx->stream4, ctx->stream, ctx->stream2, ctx->stream3};
    uint8_t* dt = (uint8_t*)ctx->walk_tup.d;
    double worst = 0;
    auto timed = [&](hipError_t e, const std::chrono::steady_clock::time_point& t0) {
        worst = std::max(worst, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return e == hipSuccess;
    };
    for (int i = 0; i < n; i++) {
        const size_t b = std::min(bytes[i], (size_t)8 << 20);
        // every kind of command a pass queues, on every stream it queues them on: fills (large, small), copies both ways (large, small),
        // an event recorded on one stream and awaited on the next
        for (int k = 0; k < 4; k++) {
            auto t0 = std::chrono::steady_clock::now();
            if (!timed(hipMemsetAsync(dt + ((size_t)k << 20), 0, (size_t)512 << 10, ss[k]), t0)) return -1;
            t0 = std::chrono::steady_clock::now();
            if (!timed(hipMemsetAsync(dt + ((size_t)k << 20), 0, 4, ss[k]), t0)) return -1;
            t0 = std::chrono::steady_clock::now();
            if (!timed(hipMemcpyAsync(dt + ((size_t)(4 + k) << 20), ctx->walk_pin.h, std::min<size_t>(ctx->walk_pin.cap, 80000), hipMemcpyHostToDevice, ss[k]), t0)) return -1;
            t0 = std::chrono::steady_clock::now();
            if (!timed(hipEventRecord(ctx->ev_w[k], ss[k]), t0)) return -1;
            t0 = std::chrono::steady_clock::now();
            if (!timed(hipStreamWaitEvent(ss[(k + 1) & 3], ctx->ev_w[k], 0), t0)) return -1;
        }
        auto t0 = std::chrono::steady_clock::now();
        if (!timed(hipMemcpyAsync(pinned[i], dt, b, hipMemcpyDeviceToHost, ss[i & 1]), t0)) return -1;
        t0 = std::chrono::steady_clock::now();
        // (the small ones that follow the keys in a pass: offsets, statuses)
        if (!timed(hipMemcpyAsync(pinned[i], dt, std::min<size_t>(b, 160000), hipMemcpyDeviceToHost, ss[(i + 1) & 1]), t0)) return -1;
    }
    for (int k = 0; k < 4; k++) (void)hipStreamSynchronize(ss[k]);
    return worst;
}

// A registered key's comb table built OUTSIDE the registration (6 ms of host arithmetic per key: the provider builds the tables of the
// identities a block made eligible side by side on its worker pool, then installs them one after the other so that every device hands
// out the same ids).
size_t key_table_words() { return KeyTab8::TABLE_WORDS; }
bool key_table_build(const uint8_t* qx32, const uint8_t* qy32, int32_t* out) {
    if (!qx32 || !qy32 || !out || !fabgpu_p256_pubkey_on_curve(qx32, qy32)) return false;
    u256 qx, qy;
    from_be32(qx, qx32);
    from_be32(qy, qy32);
    build_key_comb_table8(out, qx, qy);
    return true;
}
// TEST HOOK support: the first word at which the context's generator comb differs from the host builder's table (-1: identical; -2: error)
int64_t gtab_compare_with_host(fabgpu_ctx* ctx) {
    if (!ctx || !ctx->d_gtab) return -2;
    std::vector<int32_t> host(GTab16::TABLE_WORDS), dev(GTab16::TABLE_WORDS);
    build_g_comb_table16(host.data());
    DeviceGuard g(ctx->device);
    if (hipMemcpy(dev.data(), ctx->d_gtab, sizeof(int32_t) * GTab16::TABLE_WORDS, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (size_t i = 0; i < host.size(); i++)
        if (host[i] != dev[i]) return (int64_t)i;
    return -1;
}
// TEST HOOK support (FABGPU_FLAG_KEY_TABLES_16BIT): waits for the queued builds, then checks key `key_id`'s 16-bit comb against its 8-bit
// comb - T16[w][d] = T8[2 w][d] and T16[w][256 d] = T8[2 w + 1][d] for d = 1 .. 255, entry for entry (two tables built by different
// launches that must agree where they overlap).  Returns the number of 16-bit tables of the context; -1: the key has none; -2: error;
// -(1000 + w): window w disagrees.
int64_t key_tables16_check(fabgpu_ctx* ctx, uint32_t key_id) {
    if (!ctx) return -2;
    DeviceGuard g(ctx->device);
    void* t16 = nullptr;
    int32_t* t8 = nullptr;
    int64_t count = 0;
    {
        std::lock_guard<std::mutex> lk(ctx->kmu);
        if (ctx->stream_keytab16 && hipStreamSynchronize(ctx->stream_keytab16) != hipSuccess) return -2;
        if (key_id >= ctx->ktabs.size()) return -2;
        count = (int64_t)ctx->ktab16_count;
        t8 = ctx->ktabs[key_id];
        t16 = key_id < ctx->ktab16.size() ? ctx->ktab16[key_id] : nullptr;
    }
    if (!t16) return -1;
    const int32_t* slot[KTAB_STRIDE] = {nullptr, nullptr};
    if (hipMemcpy(slot, (const void*)(ctx->d_ktabs + KTAB_STRIDE * (size_t)key_id), sizeof(slot), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (slot[0] != t8 || slot[1] != (const int32_t*)t16) return -2;       // the device's pointer array says the same
    std::vector<int32_t> h8(KeyTab8::TABLE_WORDS), e16(COMB_ENTRY_WORDS);
    if (hipMemcpy(h8.data(), t8, sizeof(int32_t) * KeyTab8::TABLE_WORDS, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (int w = 0; w < GTab16::WINDOWS; w++)
        for (uint32_t d = 1; d < 256; d++)
            for (int hi = 0; hi < 2; hi++) {
                const size_t at16 = GTab16::index(w, hi ? d << 8 : d), at8 = KeyTab8::index(2 * w + hi, d);
                if (hipMemcpy(e16.data(), (const int32_t*)t16 + at16, sizeof(int32_t) * COMB_ENTRY_WORDS, hipMemcpyDeviceToHost) != hipSuccess) return -2;
                if (memcmp(e16.data(), &h8[at8], sizeof(int32_t) * COMB_ENTRY_WORDS) != 0) return -(1000 + w);
            }
    return count;
}
int key_register_batch(fabgpu_ctx* ctx, int n, const uint8_t* qxy, uint32_t* key_ids) { return key_register_batch_dev(ctx, n, qxy, key_ids); }
// TEST HOOK support: the device's table of key `key_id` copied to the host (KeyTab8::TABLE_WORDS words)
int key_table_copy(fabgpu_ctx* ctx, uint32_t key_id, int32_t* out) {
    if (!ctx || !out) return FABGPU_EINVAL;
    int32_t* d = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->kmu);
        if (key_id >= ctx->ktabs.size()) return FABGPU_EINVAL;
        d = ctx->ktabs[key_id];
    }
    DeviceGuard g(ctx->device);
    return hip_to_rc(hipMemcpy(out, d, sizeof(int32_t) * KeyTab8::TABLE_WORDS, hipMemcpyDeviceToHost));
}
int key_register_many_prebuilt(fabgpu_ctx* const* ctxs, int n, const uint8_t* qx32, const uint8_t* qy32, const int32_t* table, uint32_t* key_ids) {
    return key_register_many_impl(ctxs, n, qx32, qy32, table, key_ids);
}

static std::mutex g_huge_mu;
static std::set<void*> g_huge_tables;      // memo tables that came from pinned_huge_alloc (the others from hipHostMalloc)
void* walk_pinned_alloc(fabgpu_ctx* ctx, size_t bytes) {
    if (!ctx || bytes == 0) return nullptr;
    DeviceGuard g(ctx->device);
    static const bool timing = getenv("FABGPU_PASS_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // (portable: a memo table is recycled by whichever device of the provider's pool runs the next pass; on huge pages where the kernel
    //  grants them: the table is probed at random by every validator thread)
    void* p = pinned_huge_alloc(bytes);
    if (p) {
        std::lock_guard<std::mutex> lk(g_huge_mu);
        g_huge_tables.insert(p);
    } else if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
        return nullptr;
    }
    if (timing) fprintf(stderr, "fabgpu: %.1f MB of pinned memory for a memo table in %.2f ms\n", bytes / 1e6, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return p;
}
void walk_pinned_free(fabgpu_ctx* ctx, void* p) {
    if (!ctx || !p) return;
    static const bool timing = getenv("FABGPU_PASS_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    DeviceGuard g(ctx->device);
    bool huge = false;
    {
        std::lock_guard<std::mutex> lk(g_huge_mu);
        huge = g_huge_tables.erase(p) != 0;
    }
    if (huge) pinned_huge_free(p);
    else (void)hipHostFree(p);
    if (timing) fprintf(stderr, "fabgpu: a memo table's pinned memory freed in %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

}  // namespace fab
