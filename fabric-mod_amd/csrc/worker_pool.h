// A small persistent worker pool for the host side of the block pass (walk, gates, memo fill).
//
// Round 1 spawned its workers per pass (8 for the walk, 16 for the gates): on the GPU boxes of the pool (2 x EPYC 9575F, 256 CPUs
// visible, cgroup quota 16) creating and joining two dozen threads costs 0.5-1 ms per block - as much as the device call - and it
// showed most on small blocks (a 1 000-transaction block: walk 1.4 ms, of which the parse itself is 0.15 ms).  Here the threads are
// created once per process, park on a condition variable, and a pass wakes as many as it wants.
//
// run(k, fn) executes fn(0) .. fn(k-1), fn(0) on the CALLING thread (it usually has work of its own first: the walk's lister), and
// returns when all are done.  One job at a time: a second caller that finds the pool busy (two channels validating at once) WAITS for
// it - a stage lasts a millisecond or two, and two stages spinning side by side on a container's CPU quota cost more than taking
// turns (measured: two callers, 10 000-transaction blocks, 4.9 ms per block when the second spawned threads of its own against 3.7 ms
// for one caller alone).  Only a caller that waited 20 ms, or one that is itself a pool worker, gets false and runs on its own threads.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace fab {

class WorkerPool {
   public:
    static WorkerPool& instance() {
        static WorkerPool p(15);            // + the caller = 16
        return p;
    }
    // false: busy, nothing was run
    bool run(int k, const std::function<void(int)>& fn) {
        if (k <= 1) {
            fn(0);
            return true;
        }
        if (inside_job()) return false;                                   // a nested job would wait for itself
        std::unique_lock<std::timed_mutex> use(use_mu_, std::chrono::milliseconds(20));
        if (!use.owns_lock()) return false;
        const int helpers = k - 1 < (int)threads_.size() ? k - 1 : (int)threads_.size();
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &fn;
            want_ = helpers;
            total_ = k;
            next_index_.store(helpers + 1, std::memory_order_relaxed);   // indices beyond the helpers are drained by whoever is free
            remaining_.store(helpers, std::memory_order_relaxed);
            gen_++;
        }
        cv_.notify_all();
        inside_job() = true;
        fn(0);
        drain(fn);                                                        // k larger than the pool: the caller takes what is left
        inside_job() = false;
        // wait for the helpers (short: spin, then sleep)
        for (int spin = 0; remaining_.load(std::memory_order_acquire) != 0; spin++) {
            if (spin < 2000) std::this_thread::yield();
            else {
                std::unique_lock<std::mutex> lk(mu_);
                done_cv_.wait_for(lk, std::chrono::microseconds(200), [&] { return remaining_.load(std::memory_order_acquire) == 0; });
            }
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = nullptr;
        }
        return true;
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }

    // a pool of its own (a device context keeps four copier threads for its uploads: fabgpu_api.hip fabgpu_arena_stage)
    explicit WorkerPool(int n) {
        for (int i = 0; i < n; i++) threads_.emplace_back([this, i] { loop(i); });
    }

   private:
    void drain(const std::function<void(int)>& fn) {
        for (;;) {
            int idx = next_index_.fetch_add(1, std::memory_order_relaxed);
            if (idx >= total_) return;
            fn(idx);
        }
    }
    void loop(int me) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                if (me < want_) job = job_;
            }
            if (!job) continue;
            inside_job() = true;
            (*job)(me + 1);
            drain(*job);
            inside_job() = false;
            if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(mu_);
                done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    static bool& inside_job() {
        static thread_local bool f = false;
        return f;
    }
    std::timed_mutex use_mu_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)>* job_ = nullptr;
    int want_ = 0, total_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    std::atomic<int> remaining_{0}, next_index_{0};
};

// k workers (worker 0 = the caller) through the pool, or - pool busy - on threads of their own
inline void run_workers(int k, const std::function<void(int)>& fn, WorkerPool* pool = nullptr) {
    if ((pool ? *pool : WorkerPool::instance()).run(k, fn)) return;
    std::vector<std::thread> th;
    for (int w = 1; w < k; w++) th.emplace_back(fn, w);
    fn(0);
    for (auto& t : th) t.join();
}

}  // namespace fab
