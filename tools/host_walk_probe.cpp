// How fast does the HOST count a block's envelopes (walk::walk_envelope with the counting emitter), on k threads, while nothing else runs?
// g++ -O2 -std=c++17 -Ifabric-mod_amd/csrc -Iinclude tools/host_walk_probe.cpp fabric-mod_amd/csrc/block_prepass.cpp -o /tmp/hwp -lpthread
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "block_prepass.h"
#include "block_walk_core.h"
using namespace fab::bccsp;
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    size_t n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b(n);
    if (fread(b.data(), 1, n, f) != n) return 1;
    for (int rep = 0; rep < 2; rep++) {
        std::vector<uint8_t> c(b);   // a fresh copy, as a block arrives
        ParsedBlock pb;
        std::vector<uint32_t> es, ps;
        std::vector<BlockTuple> bs;
        auto t = std::chrono::steady_clock::now();
        OutlineBlock(c.data(), n, pb, es, bs, &ps);
        printf("outline %.3f ms (%zu envelopes)\n", ms(t), es.size() / 2);
        const uint32_t ne = (uint32_t)(es.size() / 2);
        for (int th : {1, 2, 4, 8, 12}) {
            std::vector<uint32_t> cnt(4 * (size_t)ne);
            std::atomic<uint32_t> next(0);
            auto work = [&] {
                for (;;) {
                    const uint32_t lo = next.fetch_add(64);
                    if (lo >= ne) return;
                    for (uint32_t e = lo; e < std::min(ne, lo + 64); e++) {
                        walk::CountEmitter em;
                        uint8_t ty = 255, und = 0;
                        walk::walk_envelope(c.data(), c.data() + es[2 * e], es[2 * e + 1], e, em, ty, und);
                        cnt[4 * (size_t)e] = em.nt;
                    }
                }
            };
            t = std::chrono::steady_clock::now();
            std::vector<std::thread> v;
            for (int k = 1; k < th; k++) v.emplace_back(work);
            const double spawned = ms(t);
            work();
            for (auto& x : v) x.join();
            printf("count walk, %2d threads: %.3f ms (spawning took %.3f)\n", th, ms(t), spawned);
        }
    }
}
