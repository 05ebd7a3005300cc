#include <chrono>
#include <cstdio>
#include <vector>
#include "block_prepass.h"
using namespace fab::bccsp;
int main(int argc, char** argv) {
    FILE* f = fopen("/tmp/blk10k.bin", "rb");
    std::vector<uint8_t> b(60000000);
    size_t n = fread(b.data(), 1, b.size(), f);
    for (int threads : {1, 8}) {
        ParsedBlock pb;                                   // reused: storage survives from block to block
        for (int it = 0; it < 5; it++) {
            auto t0 = std::chrono::steady_clock::now();
            ParseBlock(b.data(), n, pb, threads);
            auto t1 = std::chrono::steady_clock::now();
            printf("threads %d: %.2f ms, %zu tuples %zu checks\n", threads, std::chrono::duration<double, std::milli>(t1 - t0).count(), pb.tuples.size(), pb.hash_checks.size());
        }
    }
}
