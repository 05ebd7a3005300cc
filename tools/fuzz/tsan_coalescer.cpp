// ThreadSanitizer run of csrc/coalescer.h with a fake device (g++ -fsanitize=thread): see tools/fuzz/run_tsan.sh
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "coalescer.h"
using namespace fab;
struct Req : CoalescedBase {
    long x, y;
};
static int run_case(int threads, int calls, unsigned launch_us, unsigned window_us, unsigned max_batch) {
    Coalescer<Req> co;
    co.configure(window_us, max_batch);
    std::atomic<int> wrong(0);
    auto runner = [&](std::vector<Req*>& batch) {
        if (launch_us) std::this_thread::sleep_for(std::chrono::microseconds(launch_us));
        for (Req* q : batch) q->y = 3 * q->x + 1;
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            for (int c = 0; c < calls; c++) {
                Req r;
                r.x = (long)t * 1000003 + c;
                r.y = 0;
                co.submit(&r, runner);
                if (r.y != 3 * r.x + 1) wrong++;
            }
        });
    for (auto& t : th) t.join();
    return wrong.load();
}
int main() {
    int bad = 0;
    bad += run_case(32, 200, 100, 20, 32768);
    bad += run_case(48, 150, 50, 0, 4);
    bad += run_case(3, 2000, 0, 5, 2);
    bad += run_case(64, 100, 0, 0, 32768);
    printf("tsan coalescer: wrong answers %d\n", bad);
    return bad != 0;
}
