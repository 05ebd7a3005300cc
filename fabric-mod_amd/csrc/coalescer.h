// Coalescing of one-signature calls into device batches.
//
// bccsp.Verify (bccsp/sw/impl.go:247-270) and identity.Verify (msp/identities.go:169-196) take ONE signature and block; a launch costs
// 0.7 ms whether it carries one signature or thirty thousand.  Where many callers are in those functions at the same moment - the
// orderer's Broadcast handlers behind SigFilter (orderer/common/msgprocessor/sigfilter.go:50-80: one goroutine per client stream), the
// validator pool on verdict-memo misses (core/committer/txvalidator/v20/validator.go:198-208) - their calls can share a launch.
//
// Leader / follower, no thread of its own: a caller queues its request; if nobody is leading it becomes the leader, gives company a
// short window to arrive when it is alone, takes what is queued (at most max_batch), runs it and marks the requests done.  Everybody
// who arrives during a launch queues up behind it and travels with the next one - under load the batch size regulates itself
// (arrival rate x launch time) and the window never waits.
// Built so that many callers do not meet on one lock:
//   * the queue is a lock-free stack (one compare-and-swap to push, one exchange for the leader to take everything) and the lead is an
//     atomic flag;
//   * every request sleeps on a mutex + condition variable of ITS OWN; whoever completes a request does so under that mutex, so it
//     never touches a request that is gone (the owner cannot see `done` without it);
//   * wake-ups fan out as a binary tree: the leader links the batch members (child[0], child[1]) and wakes only the root; every
//     member, once awake, wakes its two children before it returns;
//   * a finishing leader hands the lead to the owner of the oldest queued request before it wakes its own batch.
// Measured (tools/coalesce_harness.c, native threads through the C ABI, MI355X box): 1.4 k calls/s from one caller (one launch per
// call), 20 k from 16, 48 k from 64, 145 k from 256 (mean batch 127).  Beyond that the HOST decides, not this file: 62 k from 1 024
// callers and 35 k from 4 096 - the same figures with a fake device that merely sleeps (tools/coalesce_fake_stress.py) and with three
// simpler versions of this queue (one shared mutex and condition variable; per-request condition variables; serial wake-ups): the box
// gives a container 16 CPUs' worth of time on 256 CPUs, and with thousands of threads waking at once 98 % of it went to the kernel
// (cpu.stat: 145 s system against 2.4 s user, 88 of 165 periods throttled).
// Requests live on their callers' stacks: a caller cannot return before `done` is set.
#pragma once
#include <atomic>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <vector>

namespace fab {

struct CoalescedBase {                 // a request type derives from this
    std::mutex m;
    std::condition_variable cv;
    bool done = false, lead = false;                 // guarded by m
    CoalescedBase* next = nullptr;                   // queue link
    CoalescedBase* child[2] = {nullptr, nullptr};    // written by the leader before the tree's root is woken, read by the owner after
    static void complete(CoalescedBase* q) {
        std::lock_guard<std::mutex> l(q->m);
        q->done = true;
        q->cv.notify_one();
    }
    void wake_children() {                           // called by the owner once it has seen `done` (its request is still its own)
        CoalescedBase *a = child[0], *b = child[1];
        if (a) complete(a);
        if (b) complete(b);
    }
};

template <class Req>
class Coalescer {
   public:
    typedef std::function<void(std::vector<Req*>&)> Runner;      // must fill every request's answer; must not throw
    void submit(Req* r, const Runner& run) {
        calls_.fetch_add(1, std::memory_order_relaxed);
        r->done = r->lead = false;
        r->child[0] = r->child[1] = nullptr;
        CoalescedBase* h = head_.load(std::memory_order_relaxed);
        do {
            r->next = h;
        } while (!head_.compare_exchange_weak(h, r, std::memory_order_seq_cst, std::memory_order_relaxed));
        bool lead = !leader_.exchange(true);
        for (;;) {
            if (!lead) {
                std::unique_lock<std::mutex> l(r->m);
                r->cv.wait(l, [&] { return r->done || r->lead; });
                if (r->done) {
                    l.unlock();
                    r->wake_children();
                    return;
                }
                r->lead = false;
            }
            lead = false;
            // ---- leading: `waiting_` (oldest first) belongs to whoever leads ----
            drain();
            const uint32_t window = window_us_.load(std::memory_order_relaxed);
            if (window && waiting_.size() == 1) {                        // alone: give company a moment
                std::this_thread::sleep_for(std::chrono::microseconds(window));
                drain();
            }
            const size_t cap = max_batch_.load(std::memory_order_relaxed);
            std::vector<Req*> batch;
            while (!waiting_.empty() && batch.size() < cap) {
                batch.push_back(waiting_.front());
                waiting_.pop_front();
            }
            run(batch);
            launches_.fetch_add(1, std::memory_order_relaxed);
            uint64_t big = largest_.load(std::memory_order_relaxed);
            while (batch.size() > big && !largest_.compare_exchange_weak(big, batch.size(), std::memory_order_relaxed)) {}
            // who leads next: the owner of the oldest request still waiting; nobody waiting: the next caller elects itself
            Req* nx = nullptr;
            for (;;) {
                drain();
                if (!waiting_.empty()) {
                    nx = waiting_.front();
                    break;
                }
                leader_.store(false);
                if (head_.load() == nullptr) break;
                if (leader_.exchange(true)) break;                        // the newcomer saw the flag down and leads itself
            }                                                             // (else it saw the flag up and sleeps: lead on, for it)
            if (nx && nx != r) {                                          // last touch of waiting_ was above
                std::lock_guard<std::mutex> l(nx->m);
                nx->lead = true;
                nx->cv.notify_one();
            }
            bool mine = false;
            size_t k = 0;
            for (Req* q : batch) {                                        // the others, compacted to batch[0 .. k)
                if (q == r) mine = true;
                else batch[k++] = q;
            }
            for (size_t i = 0; i < k; i++) {
                batch[i]->child[0] = 2 * i + 1 < k ? batch[2 * i + 1] : nullptr;
                batch[i]->child[1] = 2 * i + 2 < k ? batch[2 * i + 2] : nullptr;
            }
            if (k) CoalescedBase::complete(batch[0]);                     // the root wakes the rest
            if (mine) return;                                             // (nobody else reads r->done)
            if (nx == r) lead = true;                                     // more than max_batch were ahead of this caller: lead again
        }
    }
    void configure(uint32_t window_us, uint32_t max_batch) {
        window_us_.store(window_us);
        max_batch_.store(max_batch ? max_batch : 1);
    }
    void stats(uint64_t* calls, uint64_t* launches, uint64_t* largest) {
        if (calls) *calls = calls_.load();
        if (launches) *launches = launches_.load();
        if (largest) *largest = largest_.load();
    }

   private:
    void drain() {                                   // leader only: everything queued so far goes behind waiting_, oldest first
        CoalescedBase* l = head_.exchange(nullptr);
        size_t n = 0;
        for (CoalescedBase* q = l; q; q = q->next) n++;
        const size_t at = waiting_.size();
        waiting_.resize(at + n);
        for (CoalescedBase* q = l; q; q = q->next) waiting_[at + --n] = static_cast<Req*>(q);
    }
    std::atomic<CoalescedBase*> head_{nullptr};
    std::atomic<bool> leader_{false};
    std::deque<Req*> waiting_;
    std::atomic<uint32_t> window_us_{50};
    std::atomic<size_t> max_batch_{32768};
    std::atomic<uint64_t> calls_{0}, launches_{0}, largest_{0};
};

}  // namespace fab
