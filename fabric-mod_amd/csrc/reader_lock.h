// A reader-writer lock whose READERS do not share a cache line.
//
// The verdict memo is read by every validator thread for every signature of a block (bccsp.Verify -> fabgpu_csp_memo_lookup: 40 000
// lookups per 10 000-transaction block from validatorPoolSize threads) and written a few times per block (publish, evict).  With a
// std::shared_timed_mutex every lookup is two atomic read-modify-writes of ONE word that sixteen cores on two sockets pass around:
// measured through tools/go_call_replay.c, 4-5 us per lookup for 0.1 us of hashing and probing - the validators' 15 ms per block were
// that, not their SHA-256.  Here a reader touches only the counter of its own shard (a cache line of its own, chosen per thread) and
// reads a writer flag that stays shared in every core's cache; a writer raises the flag and waits for every shard to drain.
// (The classic "big reader" lock; the handshake is Dekker's - reader: count up, then look at the flag; writer: raise the flag, then look
// at the counts - so both sides use sequentially consistent operations.)
//
// Interface of std::shared_timed_mutex as far as std::unique_lock / std::shared_lock need it.  Not fair: writers wait for readers that
// are already in, readers that find the flag up step back and wait.  lock_shared / unlock_shared must be called by the same thread.
#pragma once
#include <atomic>
#include <mutex>
#include <thread>

namespace fab {

class BigReaderLock {
   public:
    static constexpr unsigned kShards = 64;
    void lock_shared() {
        Shard& s = shard();
        for (;;) {
            s.readers.fetch_add(1, std::memory_order_seq_cst);
            if (!writer_.load(std::memory_order_seq_cst)) return;
            s.readers.fetch_sub(1, std::memory_order_seq_cst);           // a writer is in, or coming: out of its way
            for (unsigned spin = 0; writer_.load(std::memory_order_acquire); spin++) {
                if (spin < 200) pause();
                else std::this_thread::yield();
            }
        }
    }
    void unlock_shared() { shard().readers.fetch_sub(1, std::memory_order_release); }
    void lock() {
        wmu_.lock();                                                     // one writer at a time
        writer_.store(1, std::memory_order_seq_cst);
        for (unsigned i = 0; i < kShards; i++)
            for (unsigned spin = 0; shards_[i].readers.load(std::memory_order_seq_cst) != 0; spin++) {
                if (spin < 200) pause();
                else std::this_thread::yield();
            }
    }
    void unlock() {
        writer_.store(0, std::memory_order_release);
        wmu_.unlock();
    }

   private:
    struct alignas(64) Shard {
        std::atomic<uint32_t> readers{0};
        char pad[60];
    };
    static void pause() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    Shard& shard() {
        static std::atomic<unsigned> next{0};
        static thread_local unsigned mine = next.fetch_add(1, std::memory_order_relaxed) % kShards;   // threads take shards in turn
        return shards_[mine];
    }
    Shard shards_[kShards];
    alignas(64) std::atomic<uint32_t> writer_{0};
    std::mutex wmu_;
};

// a counter many threads add to and somebody reads now and then: one cache line per shard
class ShardedCounter {
   public:
    void add(uint64_t v) {
        static std::atomic<unsigned> next{0};
        static thread_local unsigned mine = next.fetch_add(1, std::memory_order_relaxed) % kShards;
        c_[mine].v.fetch_add(v, std::memory_order_relaxed);
    }
    uint64_t load() const {
        uint64_t s = 0;
        for (unsigned i = 0; i < kShards; i++) s += c_[i].v.load(std::memory_order_relaxed);
        return s;
    }

   private:
    static constexpr unsigned kShards = 64;
    struct alignas(64) Cell {
        std::atomic<uint64_t> v{0};
        char pad[56];
    };
    Cell c_[kShards];
};

}  // namespace fab
