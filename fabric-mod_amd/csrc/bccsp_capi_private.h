// Private to the library and its test hooks: what a fabgpu_csp handle (include/fabgpu_bccsp.h) is made of.  Not an interface.
#pragma once
#include <chrono>
#include <memory>
#include <mutex>
#include <string>

#include "../../include/fabgpu_bccsp.h"
#include "bccsp_host.h"

struct fabgpu_csp {
    std::unique_ptr<fab::bccsp::GPUCSP> csp;
    // An upload whose pass ended with FABGPU_ETOOBIG (the caller's arrays were too small): kept for the retry - the same buffer, length,
    // block_seq AND content fingerprint, within a second - so that making room costs the caller no second upload.  One slot.
    // The retry contract (fabgpu_bccsp.h "FABGPU_ETOOBIG and the retry"):
    //  * the upload is JOINED before FABGPU_ETOOBIG is returned: fabgpu_arena_stage has finished reading the caller's buffer (it copies
    //    through staging memory the context owns and waits for its DMAs), so nothing reads the caller's memory after the call returned -
    //    the caller may free or reuse the buffer, and the cgo rule "C must not keep a Go pointer past the call" holds;
    //  * a parked upload only matches a call that presents the same pointer, length, block_seq and the same first / last KiB within one
    //    second; anything else starts a fresh upload, and a parked upload older than a second is dropped by the next pass or by
    //    fabgpu_csp_block_pass_abandon (its device counts as busy until then: GPUCSP::RouteBlock).
    std::mutex orphan_mu;
    std::unique_ptr<fab::bccsp::GPUCSP::BlockUpload> orphan;
    std::chrono::steady_clock::time_point orphan_at;
    uint64_t orphan_print = 0;
    // FNV-1a over the length, the first and last KiB and 64 samples in between: cheap (6 KiB), and enough to tell a different block that happens to sit at a
    // re-used address with the same length from the block that was uploaded (verdicts never depend on it being collision-free against
    // an adversary: a peer's caller retries with ITS OWN buffer; this guards against an honest caller's allocator)
    static uint64_t fingerprint(const uint8_t* block, size_t len) {
        uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)len;
        auto mix = [&](const uint8_t* p, size_t n) {
            for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
        };
        const size_t k = len < 1024 ? len : 1024;
        mix(block, k);
        mix(block + len - k, k);
        // ... and 64 bytes at each of 64 places spread over the rest (a buffer changed in its MIDDLE between the attempts
        // used to pass; the contract still says the caller must leave the buffer alone between them - this narrows what an honest
        // caller's reused allocation can get past, it is not a guarantee)
        if (len > 4096)
            for (size_t i = 1; i <= 64; i++) mix(block + (len - 64) / 65 * i, 64);
        return h;
    }
    std::unique_ptr<fab::bccsp::GPUCSP::BlockUpload> upload_for(const uint8_t* block, size_t len, uint64_t seq, bool keep_host_copy) {
        std::unique_ptr<fab::bccsp::GPUCSP::BlockUpload> up, stale;
        {
            std::lock_guard<std::mutex> lk(orphan_mu);
            if (orphan) {
                const bool fresh = std::chrono::steady_clock::now() - orphan_at <= std::chrono::seconds(1);
                if (fresh && orphan->block == block && orphan->len == len && orphan->seq == seq && orphan_print == fingerprint(block, len)) up = std::move(orphan);
                else if (!fresh) stale = std::move(orphan);
            }
        }
        stale.reset();                                       // (outside the lock)
        if (!up) {
            up.reset(new fab::bccsp::GPUCSP::BlockUpload);
            csp->StartBlockUpload(*up, block, len, seq, keep_host_copy);     // the block travels while it is walked
        }
        return up;
    }
    void park(std::unique_ptr<fab::bccsp::GPUCSP::BlockUpload> up) {
        up->join();                                          // nothing reads the caller's buffer once FABGPU_ETOOBIG has been returned
        const uint64_t print = up->block ? fingerprint(up->block, up->len) : 0;   // (still inside the call: the buffer is the caller's to lend)
        std::unique_ptr<fab::bccsp::GPUCSP::BlockUpload> old;
        {
            std::lock_guard<std::mutex> lk(orphan_mu);
            old = std::move(orphan);
            orphan = std::move(up);
            orphan_at = std::chrono::steady_clock::now();
            orphan_print = print;
        }
    }
    bool abandon() {
        std::unique_ptr<fab::bccsp::GPUCSP::BlockUpload> old;
        {
            std::lock_guard<std::mutex> lk(orphan_mu);
            old = std::move(orphan);
        }
        return old != nullptr;
    }
    // which way the block passes went (fabgpu_csp_pass_routes)
    std::mutex route_mu;
    uint64_t device_walks = 0, host_walks = 0;
    std::string last_decline;
    void note_route(bool on_device, const char* why) {
        std::lock_guard<std::mutex> lk(route_mu);
        if (on_device) device_walks++;
        else {
            host_walks++;
            last_decline = why ? why : "";
        }
    }
};

