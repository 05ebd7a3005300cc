// Walker-only timing harness (no device): ParseBlock in a loop over a marshalled block file (tools/make_walk_block.py writes one).
//   g++ -O3 -std=c++17 -Ifabric-mod_amd/csrc tools/walk_harness.cpp fabric-mod_amd/csrc/block_prepass.cpp -o /tmp/walk -lpthread
//   /tmp/walk /tmp/blk.bin <threads>
#include "block_prepass.h"
#include <chrono>
#include <cstdio>
#include <vector>
using namespace fab::bccsp;
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> b(n); if (fread(b.data(), 1, n, f) != (size_t)n) return 1; fclose(f);
    int threads = argc > 2 ? atoi(argv[2]) : 16;
    ParsedBlock pb;
    for (int it = 0; it < 8; it++) {
        auto t0 = std::chrono::steady_clock::now();
        bool ok = ParseBlock(b.data(), b.size(), pb, threads);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("%d ok=%d n_tx=%u tuples=%zu %.2f ms\n", it, ok, pb.n_tx, pb.tuples.size(), ms);
    }
}
