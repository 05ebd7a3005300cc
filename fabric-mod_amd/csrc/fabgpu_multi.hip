// One batch, G devices (BASELINE.json configs[2] "same block sharded across 8 MI355X, RCCL all-gather of the
// verdict bitmap over xGMI"): ONE process, one fabgpu context + one HIP stream per device, the batch cut into contiguous 64-aligned
// shards (multi_plan.h), every shard uploaded from pinned staging and verified on its device, then ONE ncclAllGather (RCCL) of the
// shard bitmaps - the merged bitmap is then resident on EVERY device (for a later on-device policy step) - and a single D2H from
// device 0.  No other data-path collective exists: signatures are independent.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a peer with one GPU, or a box without RCCL, still loads libfabgpu.so; and
// FABGPU_MULTI_HOST_MERGE replaces the collective by G small D2H copies (the survey's "the host could equally do G small D2H
// copies") - also what a test uses to run several shards on ONE physical device, which RCCL refuses (duplicate device in a
// communicator).
#include <dlfcn.h>
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fabgpu.h"
#include "multi_plan.h"
#include "worker_pool.h"

using namespace fab;

namespace {

// the five RCCL entry points this file needs, resolved once
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
    void* so = nullptr;
    bool ok = false;
    static constexpr int kUint64 = 5;   // ncclUint64 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5)
    bool load() {
        if (ok) return true;
        so = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!so) return false;
        CommInitAll = (int (*)(comm_t*, int, const int*))dlsym(so, "ncclCommInitAll");
        CommDestroy = (int (*)(comm_t))dlsym(so, "ncclCommDestroy");
        GroupStart = (int (*)())dlsym(so, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(so, "ncclGroupEnd");
        AllGather = (int (*)(const void*, void*, size_t, int, comm_t, hipStream_t))dlsym(so, "ncclAllGather");
        ok = CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather;
        return ok;
    }
};

struct Dev {
    int ordinal = 0;
    fabgpu_ctx* ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_verified = nullptr;   // "this shard's verify launch (and status copy) is queued": recorded in front of the collective
    void* h_in = nullptr;      // pinned staging of the shard: fields (+ message bytes + offsets in hash mode)
    void* d_in = nullptr;
    size_t in_cap = 0, din_cap = 0;
    void* d_words = nullptr;   // this shard's verdict words (words_per_rank, zero padded)
    void* d_merged = nullptr;  // G x words_per_rank after the all-gather
    void* d_status = nullptr;
    void* h_out = nullptr;     // pinned: merged words (device 0) / own words (host merge) | status bytes
    size_t words_cap = 0, merged_cap = 0, status_cap = 0, out_cap = 0;
};

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
int hip_rc(hipError_t e) { return e == hipSuccess ? FABGPU_OK : (e == hipErrorOutOfMemory ? FABGPU_ENOMEM : FABGPU_ELAUNCH); }

int grow(void** p, size_t* cap, size_t need, bool pinned) {
    if (need <= *cap) return FABGPU_OK;
    if (*p) { if (pinned) hipHostFree(*p); else hipFree(*p); }
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 256;
    hipError_t e = pinned ? hipHostMalloc(p, want, hipHostMallocDefault) : hipMalloc(p, want);
    if (e != hipSuccess) { *p = nullptr; return FABGPU_ENOMEM; }
    *cap = want;
    return FABGPU_OK;
}

}  // namespace

struct fabgpu_multi {
    std::vector<Dev> dev;
    std::vector<Rccl::comm_t> comms;
    Rccl rccl;
    bool host_merge = false;
    std::string why;            // why the merge is what it is (fabgpu_multi_collective)
    bool comms_poisoned = false;  // a collective never came back / failed half-way: the communicators are abandoned, not destroyed
    bool devices_suspect = false; // ... and one may still sit on a device: shutdown neither synchronises nor frees there (it would wait for it)
    int selfcheck_deadline_s = 10;
    std::mutex mu;
};

namespace {

void to_host_merge(fabgpu_multi* m, const char* why) {
    m->host_merge = true;
    m->why = why;
}
// A collective that never came back, or a group call that failed half-way: the communicators are never touched again (not even
// destroyed: destroying one with a collective in flight can hang too), and the streams it sits on are abandoned with it - the shards
// get fresh ones, so that the host merge's copies are not queued behind it.  may_still_run: something may still occupy the devices
// (shutdown then leaks their memory rather than wait for it).
void abandon_collective(fabgpu_multi* m, bool may_still_run) {
    m->comms_poisoned = true;
    m->devices_suspect = m->devices_suspect || may_still_run;
    for (size_t g = 0; g < m->dev.size(); g++) {
        Dev& d = m->dev[g];
        hipStream_t fresh = nullptr;
        if (hipSetDevice(d.ordinal) == hipSuccess && hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) == hipSuccess) d.stream = fresh;   // (the old one leaks)
    }
}

// One all-gather of ONE word per device through the communicators just made, checked on EVERY device, with a deadline: a node whose
// xGMI / RCCL set-up is broken shows up here, at construction, as "host merge" - not as a wrong or missing verdict bitmap in the
// first block (the collective with G > 1 distinct devices had never executed anywhere).  The check runs on a
// helper thread; if it does not come back within the deadline the communicators are abandoned (never touched again, never destroyed:
// destroying a communicator with a collective in flight can hang too) and the host merges.
bool rccl_self_check(fabgpu_multi* m, std::string* why) {
    const int G = (int)m->dev.size();
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false, ok = false;
        std::string why;
    };
    auto sh = std::make_shared<Shared>();
    // the helper owns copies of everything it touches: if it is abandoned, `m` may change (fresh streams) or go away under it
    std::vector<int> ords((size_t)G);
    std::vector<hipStream_t> streams((size_t)G);
    for (int g = 0; g < G; g++) {
        ords[(size_t)g] = m->dev[(size_t)g].ordinal;
        streams[(size_t)g] = m->dev[(size_t)g].stream;
    }
    const std::vector<Rccl::comm_t> comms = m->comms;
    const Rccl rc = m->rccl;
    std::thread t([G, sh, ords, streams, comms, rc]() {
        bool ok = true;
        std::string w;
        std::vector<uint64_t*> d_one((size_t)G, nullptr), d_all((size_t)G, nullptr);
        for (int g = 0; g < G && ok; g++) {
            const int ord = ords[(size_t)g];
            const uint64_t word = 0xFAB6A75E1FC0DE00ull + (uint64_t)g;
            ok = hipSetDevice(ord) == hipSuccess && hipMalloc((void**)&d_one[(size_t)g], 8) == hipSuccess &&
                 hipMalloc((void**)&d_all[(size_t)g], 8 * (size_t)G) == hipSuccess &&
                 hipMemcpy(d_one[(size_t)g], &word, 8, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemset(d_all[(size_t)g], 0, 8 * (size_t)G) == hipSuccess &&
                 hipDeviceSynchronize() == hipSuccess;   // (the shards' streams are non-blocking: nothing orders them behind the null stream's memset / copy)
            if (!ok) w = "self-check: device memory";
        }
        if (ok) {
            ok = rc.GroupStart() == 0;
            for (int g = 0; g < G && ok; g++)
                ok = rc.AllGather(d_one[(size_t)g], d_all[(size_t)g], 1, Rccl::kUint64, comms[(size_t)g], streams[(size_t)g]) == 0;
            ok = (rc.GroupEnd() == 0) && ok;
            if (!ok) w = "self-check: ncclAllGather returned an error";
        }
        std::vector<uint64_t> got((size_t)G);
        for (int g = 0; g < G && ok; g++) {
            ok = hipSetDevice(ords[(size_t)g]) == hipSuccess && hipStreamSynchronize(streams[(size_t)g]) == hipSuccess &&
                 hipMemcpy(got.data(), d_all[(size_t)g], 8 * (size_t)G, hipMemcpyDeviceToHost) == hipSuccess;
            if (!ok) { w = "self-check: the all-gather's stream failed"; break; }
            for (int k = 0; k < G && ok; k++)
                if (got[(size_t)k] != 0xFAB6A75E1FC0DE00ull + (uint64_t)k) {
                    ok = false;
                    w = "self-check: device " + std::to_string(ords[(size_t)g]) + " gathered a wrong word from rank " + std::to_string(k);
                }
        }
        for (int g = 0; g < G; g++) {
            if (hipSetDevice(ords[(size_t)g]) != hipSuccess) continue;
            if (d_one[(size_t)g]) hipFree(d_one[(size_t)g]);
            if (d_all[(size_t)g]) hipFree(d_all[(size_t)g]);
        }
        std::lock_guard<std::mutex> lk(sh->mu);
        sh->ok = ok;
        sh->why = w;
        sh->done = true;
        sh->cv.notify_all();
    });
    const int deadline_s = m->selfcheck_deadline_s;           // FABGPU_MULTI_SELFCHECK_SECONDS(s) of the init flags; 10 s by default
    std::unique_lock<std::mutex> lk(sh->mu);
    if (!sh->cv.wait_for(lk, std::chrono::seconds(deadline_s), [&] { return sh->done; })) {
        lk.unlock();
        t.detach();
        abandon_collective(m, /*may_still_run=*/true);
        *why = "self-check: the all-gather did not complete within " + std::to_string(deadline_s) + " s";
        return false;
    }
    lk.unlock();
    t.join();
    *why = sh->why;
    return sh->ok;
}

}  // namespace

extern "C" {

int fabgpu_multi_init(const int32_t* devices, int n_devices, uint32_t flags, fabgpu_multi** out) {
    if (!out || n_devices <= 0 || n_devices > 64) return FABGPU_EINVAL;
    if (flags & ~(uint32_t)(FABGPU_MULTI_HOST_MERGE | 0xFF00u)) return FABGPU_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FABGPU_ENODEV;
    fabgpu_multi* m = new (std::nothrow) fabgpu_multi();
    if (!m) return FABGPU_ENOMEM;
    m->host_merge = (flags & FABGPU_MULTI_HOST_MERGE) != 0;
    if ((flags >> 8) & 0xFFu) m->selfcheck_deadline_s = (int)((flags >> 8) & 0xFFu);
    m->dev.resize((size_t)n_devices);
    int rc = FABGPU_OK;
    std::vector<int> ords((size_t)n_devices);
    for (int g = 0; g < n_devices && rc == FABGPU_OK; g++) {
        const int o = devices ? devices[g] : g;
        if (o < 0 || o >= ndev) { rc = FABGPU_EINVAL; break; }
        ords[g] = o;
        m->dev[g].ordinal = o;
        fabgpu_cfg cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.device = o;
        rc = fabgpu_init(&cfg, &m->dev[g].ctx);                 // generator comb table per device: replicated, as the survey says
        if (rc != FABGPU_OK) break;
        if (hipSetDevice(o) != hipSuccess || hipStreamCreateWithFlags(&m->dev[g].stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->dev[g].ev_verified, hipEventDisableTiming) != hipSuccess)
            rc = FABGPU_ENODEV;
    }
    if (rc == FABGPU_OK && m->host_merge) m->why = "FABGPU_MULTI_HOST_MERGE asked for it";
    if (rc == FABGPU_OK && !m->host_merge) {
        // Whatever stands between this process and a working collective - one device, a repeated ordinal (RCCL refuses duplicates in a
        // communicator), no librccl, ncclCommInitAll failing, an all-gather that returns garbage or never returns - ends in the host
        // merge (G small D2H copies: SURVEY 8(e) "the host could equally do G small D2H copies"), never in a failed batch.
        bool dup = false;
        for (int a = 0; a < n_devices; a++)
            for (int b = a + 1; b < n_devices; b++) dup = dup || ords[a] == ords[b];
        if (dup) to_host_merge(m, "a device ordinal is repeated: RCCL cannot form a communicator");
        else if (!m->rccl.load()) to_host_merge(m, "librccl.so not found");
        else {
            m->comms.assign((size_t)n_devices, nullptr);
            if (m->rccl.CommInitAll(m->comms.data(), n_devices, ords.data()) != 0) {
                m->comms.clear();
                to_host_merge(m, "ncclCommInitAll failed");
            } else {
                std::string why;
                if (!rccl_self_check(m, &why)) to_host_merge(m, why.c_str());
                else m->why = "ncclAllGather over " + std::to_string(n_devices) + " ranks, self-checked at init";
            }
        }
    }
    if (rc != FABGPU_OK) {
        fabgpu_multi_shutdown(m);
        return rc;
    }
    *out = m;
    return FABGPU_OK;
}

void fabgpu_multi_shutdown(fabgpu_multi* m) {
    if (!m) return;
    if (m->devices_suspect) {
        // a collective that never completed may still occupy a device: hipStreamSynchronize, hipFree and fabgpu_shutdown would all wait for
        // it, possibly for ever.  The process keeps its device memory until it exits; what can be released without touching a device is.
        fprintf(stderr, "fabgpu_multi_shutdown: a collective was abandoned on these devices; their buffers are left to process exit\n");
        for (size_t g = 0; g < m->dev.size(); g++) {
            Dev& d = m->dev[g];
            if (d.h_in) hipHostFree(d.h_in);
            if (d.h_out) hipHostFree(d.h_out);
        }
        delete m;
        return;
    }
    for (size_t g = 0; g < m->dev.size(); g++) {
        Dev& d = m->dev[g];
        hipSetDevice(d.ordinal);
        if (d.stream) hipStreamSynchronize(d.stream);
    }
    if (!m->comms_poisoned)
        for (auto c : m->comms)
            if (c) m->rccl.CommDestroy(c);
    for (size_t g = 0; g < m->dev.size(); g++) {
        Dev& d = m->dev[g];
        hipSetDevice(d.ordinal);
        if (d.h_in) hipHostFree(d.h_in);
        if (d.h_out) hipHostFree(d.h_out);
        if (d.d_in) hipFree(d.d_in);
        if (d.d_words) hipFree(d.d_words);
        if (d.d_merged) hipFree(d.d_merged);
        if (d.d_status) hipFree(d.d_status);
        if (d.stream) hipStreamDestroy(d.stream);
        if (d.ev_verified) hipEventDestroy(d.ev_verified);
        if (d.ctx) fabgpu_shutdown(d.ctx);
    }
    delete m;
}

int fabgpu_multi_device_count(fabgpu_multi* m) { return m ? (int)m->dev.size() : FABGPU_EINVAL; }

// the shard boundaries a batch of n tuples gets on G devices (pure host; off == NULL: by count, else by message bytes)
int fabgpu_multi_plan(size_t n, const uint32_t* off, uint32_t n_devices, uint64_t* lo, uint64_t* hi, uint64_t* words_per_rank) {
    if (!n_devices || !lo || !hi) return FABGPU_EINVAL;
    if (off)
        for (size_t i = 0; i < n; i++)
            if (off[i + 1] < off[i]) return FABGPU_EINVAL;
    ShardPlan p = off ? plan_by_bytes(n, off, n_devices) : plan_by_count(n, n_devices);
    for (uint32_t g = 0; g < n_devices; g++) {
        lo[g] = p.lo[g];
        hi[g] = p.hi[g];
    }
    if (words_per_rank) *words_per_rank = p.words_per_rank;
    return FABGPU_OK;
}

// arena / off == NULL: verify-only (e given); else fused hash + verify (e ignored)
static int multi_verify(fabgpu_multi* m, size_t n, const uint8_t* arena, const uint32_t* off, const uint8_t* qx, const uint8_t* qy, const uint8_t* e,
                        const uint8_t* r, const uint8_t* s, uint64_t* verdict_bits, uint8_t* status) {
    const bool hash = off != nullptr;
    if (!m || (n && (!qx || !qy || !r || !s || !verdict_bits || (!hash && !e)))) return FABGPU_EINVAL;
    if (n > 0x7FFFFFF0ull / 160) return FABGPU_ETOOBIG;
    if (n == 0) return FABGPU_OK;
    if (hash) {
        for (size_t i = 0; i < n; i++)
            if (off[i + 1] < off[i]) return FABGPU_EINVAL;
        if (!arena && off[n] != off[0]) return FABGPU_EINVAL;
    }
    std::lock_guard<std::mutex> lk(m->mu);
    const uint32_t G = (uint32_t)m->dev.size();
    const ShardPlan p = hash ? plan_by_bytes(n, off, G) : plan_by_count(n, G);
    const size_t wpr = p.words_per_rank;
    const int nf = hash ? 4 : 5;
    int rc = FABGPU_OK;
    // Whatever way this function is left, nothing it queued may still be in flight (the next call may free or regrow the staging
    // buffers a copy or a kernel still reads) and the caller's current device is what it was: every early return below runs this.
    struct Quiesce {
        fabgpu_multi* m;
        int prev = -1;
        bool clean = false;               // the normal path has synchronised every stream itself
        explicit Quiesce(fabgpu_multi* m_) : m(m_) { hipGetDevice(&prev); }
        ~Quiesce() {
            if (!clean)
                for (auto& d : m->dev) {
                    if (hipSetDevice(d.ordinal) == hipSuccess && d.stream) hipStreamSynchronize(d.stream);
                }
            if (prev >= 0) hipSetDevice(prev);
        }
    } quiesce(m);
    // 1a. per device: room (rare: only when a batch is larger than anything before)
    for (uint32_t g = 0; g < G; g++) {
        Dev& d = m->dev[g];
        const size_t cnt = p.hi[g] - p.lo[g], fb = cnt * 32;
        if (hipSetDevice(d.ordinal) != hipSuccess) return FABGPU_ENODEV;
        size_t msg_pad = 0, off_bytes = 0;
        if (hash && cnt) {
            msg_pad = round_up((size_t)off[p.hi[g]] - off[p.lo[g]], 4) + 128;
            off_bytes = round_up((cnt + 1) * 4, 64);
        }
        const size_t in_need = round_up((size_t)nf * fb, 64) + off_bytes + msg_pad + 64;
        if ((rc = grow(&d.h_in, &d.in_cap, in_need, true))) return rc;
        if ((rc = grow(&d.d_in, &d.din_cap, in_need, false))) return rc;
        if ((rc = grow(&d.d_words, &d.words_cap, wpr * 8 + 64, false)) || (rc = grow(&d.d_merged, &d.merged_cap, (size_t)G * wpr * 8 + 64, false)) ||
            (rc = grow(&d.h_out, &d.out_cap, (size_t)G * wpr * 8 + 64 + (status ? cnt : 0) + 64, true)) ||
            (status && (rc = grow(&d.d_status, &d.status_cap, cnt + 64, false))))
            return rc;
    }
    // 1b. per device, ALL DEVICES AT ONCE (one worker each: the host side's pool, or threads of its own when that is busy): stage the
    // shard into the device's pinned buffer, upload, launch - asynchronous on the device's own stream.  The staging is a memcpy of the
    // shard (a configs[3]-sized batch: 557 MB over the node) and was a loop on the calling thread in round 3 - the last of eight
    // devices got its first byte 50 ms after the first.  Inside a worker the copy and the upload overlap too: every STAGE_PIECE bytes
    // that are in the pinned buffer go up while the next are being copied.
    std::vector<int> rcs(G, FABGPU_OK);
    run_workers((int)G, [&](int gi) {
        const uint32_t g = (uint32_t)gi;
        Dev& d = m->dev[g];
        int& out = rcs[g];
        const size_t cnt = p.hi[g] - p.lo[g], fb = cnt * 32;
        if (hipSetDevice(d.ordinal) != hipSuccess) { out = FABGPU_ENODEV; return; }      // (the current device is per thread)
        size_t msg_bytes = 0, msg_pad = 0, off_bytes = 0;
        if (hash && cnt) {
            msg_bytes = (size_t)off[p.hi[g]] - off[p.lo[g]];
            msg_pad = round_up(msg_bytes, 4) + 128;
            off_bytes = round_up((cnt + 1) * 4, 64);
        }
        hipError_t err = hipMemsetAsync(d.d_words, 0, wpr * 8, d.stream);      // tail ranks contribute zero words
        if (err != hipSuccess) { out = hip_rc(err); return; }
        if (!cnt) return;
        uint8_t* h = (uint8_t*)d.h_in;
        uint8_t* dd = (uint8_t*)d.d_in;
        constexpr size_t STAGE_PIECE = (size_t)4 << 20;
        size_t sent = 0;
        auto staged_upto = [&](size_t end_, bool last) {                       // [sent, end_) of the pinned buffer is final: send what is worth a DMA
            if (err != hipSuccess || end_ <= sent || (!last && end_ - sent < STAGE_PIECE)) return;
            err = hipMemcpyAsync(dd + sent, h + sent, end_ - sent, hipMemcpyHostToDevice, d.stream);
            sent = end_;
        };
        auto stage = [&](size_t at, const uint8_t* from, size_t bytes) {       // h[at, at + bytes) <- from, in pieces
            for (size_t k = 0; k < bytes; k += STAGE_PIECE) {
                const size_t c = std::min(STAGE_PIECE, bytes - k);
                memcpy(h + at + k, from + k, c);
                staged_upto(at + k + c, false);
            }
        };
        const uint8_t* src[5] = {qx, qy, hash ? r : e, hash ? s : r, s};
        for (int f = 0; f < nf; f++) stage((size_t)f * fb, src[f] + 32 * p.lo[g], fb);
        const size_t o_off = round_up((size_t)nf * fb, 64), m_off = o_off + off_bytes;
        memset(h + (size_t)nf * fb, 0, o_off - (size_t)nf * fb);
        if (hash) {
            uint32_t* ho = (uint32_t*)(h + o_off);
            const uint32_t base = off[p.lo[g]];
            for (size_t i = 0; i <= cnt; i++) ho[i] = off[p.lo[g] + i] - base;
            memset(h + o_off + (cnt + 1) * 4, 0, off_bytes - (cnt + 1) * 4);
            staged_upto(m_off, false);
            if (msg_bytes) stage(m_off, arena + base, msg_bytes);
            memset(h + m_off + msg_bytes, 0, msg_pad - msg_bytes);
        }
        staged_upto(m_off + msg_pad, true);
        if (err != hipSuccess) { out = hip_rc(err); return; }
        int r2;
        if (hash)
            r2 = fabgpu_sha256_p256_verify_batch_dev(d.ctx, cnt, dd + m_off, msg_pad, dd + o_off, dd, dd + fb, dd + 2 * fb, dd + 3 * fb, d.d_words,
                                                     status ? d.d_status : nullptr, d.stream);
        else
            r2 = fabgpu_p256_verify_batch_dev(d.ctx, cnt, dd, dd + fb, dd + 2 * fb, dd + 3 * fb, dd + 4 * fb, d.d_words, status ? d.d_status : nullptr, d.stream);
        if (r2) { out = r2; return; }
        if (status) {
            err = hipMemcpyAsync((uint8_t*)d.h_out + round_up((size_t)G * wpr * 8, 64), d.d_status, cnt, hipMemcpyDeviceToHost, d.stream);
            if (err != hipSuccess) out = hip_rc(err);
        }
        if (out == FABGPU_OK && d.ev_verified && hipEventRecord(d.ev_verified, d.stream) != hipSuccess) out = FABGPU_ELAUNCH;
    });
    for (uint32_t g = 0; g < G; g++)
        if (rcs[g] != FABGPU_OK) return rcs[g];
    // 2. the verdict bitmaps: one all-gather over RCCL / xGMI (every device ends up with the merged bitmap), or G small D2H copies
    if (!m->host_merge) {
        bool ok = m->rccl.GroupStart() == 0;
        if (ok) {
            for (uint32_t g = 0; g < G && ok; g++) {
                Dev& d = m->dev[g];
                ok = m->rccl.AllGather(d.d_words, d.d_merged, wpr, Rccl::kUint64, m->comms[g], d.stream) == 0;
            }
            ok = (m->rccl.GroupEnd() == 0) && ok;
        }
        if (ok) {
            Dev& d0 = m->dev[0];
            hipSetDevice(d0.ordinal);
            hipError_t err = hipMemcpyAsync(d0.h_out, d0.d_merged, (size_t)G * wpr * 8, hipMemcpyDeviceToHost, d0.stream);
            if (err != hipSuccess) return hip_rc(err);
        } else {
            // this batch and every later one are merged by the host - on fresh streams: a group call that failed half-way may have queued
            // some ranks' share of the collective, and neither the copies below nor shutdown may wait behind those
            to_host_merge(m, "ncclAllGather returned an error on a batch: host merge from here on");
            abandon_collective(m, /*may_still_run=*/true);
            for (uint32_t g = 0; g < G; g++) {             // the fresh streams wait for the verify launches queued on the old ones - not for the collective
                Dev& d = m->dev[g];
                hipSetDevice(d.ordinal);
                if (!d.ev_verified) return FABGPU_ELAUNCH;
                hipError_t e2 = hipStreamWaitEvent(d.stream, d.ev_verified, 0);
                if (e2 != hipSuccess) return hip_rc(e2);
            }
        }
    }
    if (m->host_merge) {
        for (uint32_t g = 0; g < G; g++) {
            Dev& d = m->dev[g];
            hipSetDevice(d.ordinal);
            hipError_t err = hipMemcpyAsync(d.h_out, d.d_words, wpr * 8, hipMemcpyDeviceToHost, d.stream);
            if (err != hipSuccess) return hip_rc(err);
        }
    }
    // 3. wait, then lay the words out as ONE bitmap (count mode: the gathered buffer already is; bytes mode: shards are ragged)
    for (uint32_t g = 0; g < G; g++) {
        Dev& d = m->dev[g];
        hipSetDevice(d.ordinal);
        hipError_t err = hipStreamSynchronize(d.stream);
        if (err != hipSuccess) return hip_rc(err);
    }
    const size_t words = (n + 63) / 64;
    for (uint32_t g = 0; g < G; g++) {
        const size_t cnt = p.hi[g] - p.lo[g];
        if (!cnt) continue;
        const size_t w = (cnt + 63) / 64;
        const uint64_t* srcw = !m->host_merge ? (const uint64_t*)m->dev[0].h_out + (size_t)g * wpr : (const uint64_t*)m->dev[g].h_out;
        if (p.word_at[g] + w > words) return FABGPU_ELAUNCH;
        memcpy(verdict_bits + p.word_at[g], srcw, w * 8);
        if (status) memcpy(status + p.lo[g], (const uint8_t*)m->dev[g].h_out + round_up((size_t)G * wpr * 8, 64), cnt);
    }
    quiesce.clean = true;
    return FABGPU_OK;
}

int fabgpu_multi_p256_verify_batch(fabgpu_multi* m, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* e, const uint8_t* r,
                                   const uint8_t* s, uint64_t* verdict_bits, uint8_t* status) {
    return multi_verify(m, n, nullptr, nullptr, qx, qy, e, r, s, verdict_bits, status);
}

int fabgpu_multi_sha256_p256_verify_batch(fabgpu_multi* m, size_t n, const uint8_t* arena, const uint32_t* off, const uint8_t* qx, const uint8_t* qy,
                                          const uint8_t* r, const uint8_t* s, uint64_t* verdict_bits, uint8_t* status) {
    if (!off) return FABGPU_EINVAL;
    return multi_verify(m, n, arena, off, qx, qy, nullptr, r, s, verdict_bits, status);
}

// device pointer of the merged bitmap on device g after the last call (G x words_per_rank u64; count mode: the first ceil(n/64)
// words ARE the bitmap) - what an on-device policy evaluation would read.  NULL when the host merged.
const void* fabgpu_multi_merged_bitmap_dev(fabgpu_multi* m, int g) {
    if (!m || g < 0 || (size_t)g >= m->dev.size() || m->host_merge) return nullptr;
    return m->dev[(size_t)g].d_merged;
}

// How the shard bitmaps are merged: returns the number of RCCL ranks (G) when the merge is the ncclAllGather, 0 when the host merges
// (G small D2H copies); `why` (optional) receives the reason - which of the init-time checks chose the host merge, or that the
// collective passed its one-word self-check.
int fabgpu_multi_collective(fabgpu_multi* m, char* why, size_t why_cap) {
    if (!m) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(m->mu);
    if (why && why_cap) {
        strncpy(why, m->why.c_str(), why_cap - 1);
        why[why_cap - 1] = 0;
    }
    return m->host_merge ? 0 : (int)m->dev.size();
}

}  // extern "C"
