#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <string>
#include "block_prepass.h"
#include "idemix_host.h"
using namespace fab::bccsp;
int main() {
    FILE* f = fopen("/tmp/blk_small.bin", "rb");
    std::vector<uint8_t> base(8 << 20);
    size_t n = fread(base.data(), 1, base.size(), f);
    base.resize(n);
    std::mt19937_64 rng(7);
    size_t parsed = 0, tuples = 0, blocksigs = 0;
    for (int it = 0; it < 20000; it++) {
        std::vector<uint8_t> b = base;
        int k = 1 + rng() % 8;
        for (int j = 0; j < k; j++) {
            size_t pos = rng() % b.size();
            switch (rng() % 4) {
                case 0: b[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
                case 1: b[pos] = (uint8_t)rng(); break;
                case 2: b[pos] = 0xFF; break;                         // long varints
                default: if (b.size() > 16) b.resize(b.size() - rng() % 16); break;   // truncation
            }
        }
        // exact-size heap copy so that ASAN sees any read past the end
        uint8_t* heap = (uint8_t*)malloc(b.size());
        memcpy(heap, b.data(), b.size());
        ParsedBlock pb;
        if (ParseBlock(heap, b.size(), pb, it % 2 ? 4 : 1)) {
            parsed++;
            tuples += pb.tuples.size();
            if (pb.tail_base < b.size() || (pb.tail_base & 63u)) { printf("TAIL BASE\n"); return 1; }
            for (auto& t : pb.tuples) {
                // the message of an orderer block signature lives in the walker's tail (block_prepass.h), everything else in the block
                const bool in_tail = t.kind == TUPLE_BLOCK_SIG;
                if (in_tail) {
                    blocksigs++;
                    if (t.suffix.off < pb.tail_base || (size_t)t.suffix.off - pb.tail_base + t.suffix.len > pb.tail.size() || t.tx != BLOCK_LEVEL_TX) { printf("TAIL SPAN OUT OF RANGE\n"); return 1; }
                } else if ((size_t)t.suffix.off + t.suffix.len > b.size() || t.tx >= pb.n_tx) { printf("SPAN OUT OF RANGE\n"); return 1; }
                if ((size_t)t.identity.off + t.identity.len > b.size() || (size_t)t.sig.off + t.sig.len > b.size() ||
                    (size_t)t.prefix.off + t.prefix.len > b.size()) { printf("SPAN OUT OF RANGE\n"); return 1; }
                uint8_t qx[32], qy[32], nx[32], ny[32];
                std::string ms;
                IdentityToP256(heap + t.identity.off, t.identity.len, qx, qy);
                IdentityToIdemixNym(heap + t.identity.off, t.identity.len, ms, nx, ny);
                NymSignatureFields sf;
                UnmarshalNymSignature(heap + t.sig.off, t.sig.len, sf);
            }
            if ((size_t)pb.data.off + pb.data.len > b.size() || (size_t)pb.previous_hash.off + pb.previous_hash.len > b.size() ||
                (size_t)pb.data_hash.off + pb.data_hash.len > b.size()) { printf("HEADER SPAN OUT OF RANGE\n"); return 1; }
            for (auto& h : pb.hash_checks) {
                for (int p = 0; p < 3; p++) if ((size_t)h.piece[p].off + h.piece[p].len > b.size()) { printf("PIECE OUT OF RANGE\n"); return 1; }
                if ((size_t)h.expect.off + h.expect.len > b.size()) { printf("EXPECT OUT OF RANGE\n"); return 1; }
                uint8_t dg[32] = {0};
                HashCheckMatches(heap, h, dg);
            }
        }
        free(heap);
    }
    printf("fuzz ok: %zu of 20000 mutants parsed, %zu tuples (%zu orderer block signatures)\n", parsed, tuples, blocksigs);
}
