// ThreadSanitizer stress of reader_lock.h: readers check an invariant the writers break and restore under the write lock.
#include <stdio.h>

#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <vector>

#include "reader_lock.h"

int main() {
    fab::BigReaderLock mu;
    fab::ShardedCounter hits;
    long a = 0, b = 0;                       // invariant under the lock: a + b == 0 (plain variables: TSan sees any unprotected access)
    std::atomic<bool> stop{false};
    std::atomic<long> bad{0}, reads{0}, writes{0};
    std::vector<std::thread> th;
    for (int r = 0; r < 12; r++)
        th.emplace_back([&] {
            while (!stop.load(std::memory_order_relaxed)) {
                std::shared_lock<fab::BigReaderLock> lk(mu);
                if (a + b != 0) bad.fetch_add(1);
                hits.add(1);
                reads.fetch_add(1, std::memory_order_relaxed);
            }
        });
    for (int w = 0; w < 3; w++)
        th.emplace_back([&, w] {
            for (int k = 0; k < 20000; k++) {
                {
                    std::unique_lock<fab::BigReaderLock> lk(mu);
                    a += w + 1;
                    b -= w + 1;
                }
                writes.fetch_add(1, std::memory_order_relaxed);
                if ((k & 63) == 0) std::this_thread::yield();
            }
        });
    for (size_t i = 12; i < th.size(); i++) th[i].join();
    stop.store(true);
    for (size_t i = 0; i < 12; i++) th[i].join();
    printf("tsan reader lock: %ld reads, %ld writes, invariant broken %ld times, counter %s\n", reads.load(), writes.load(), bad.load(),
           (long)hits.load() == reads.load() ? "ok" : "WRONG");
    return bad.load() != 0 || (long)hits.load() != reads.load();
}
