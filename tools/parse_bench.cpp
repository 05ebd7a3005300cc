// Host-only micro-benchmark of the block walker (fabric-mod_amd/csrc/block_prepass.cpp::ParseBlock) on /tmp/blk10k.bin, the
// 10 000-transaction block tools/mk_parse_block.py writes (fake signatures: the walker does not look at them).
//   python tools/mk_parse_block.py && g++ -O3 -std=c++17 -Ifabric-mod_amd/csrc tools/parse_bench.cpp fabric-mod_amd/csrc/block_prepass.cpp -o /tmp/pb -lpthread && /tmp/pb
// Used to find that the walk's serial part (locating the envelopes) dominates: DESIGN.md section 4.4.
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
