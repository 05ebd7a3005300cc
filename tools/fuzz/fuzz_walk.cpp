#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <string>
#include "block_prepass.h"
#include "idemix_host.h"
using namespace fab::bccsp;

// The device walk's two-run procedure (block_walk_core.h: CountEmitter, exclusive prefix sum, WriteEmitter) on the host, into arrays of
// EXACTLY the counted sizes (a record written outside its envelope's range is a heap overflow ASAN sees), compared with ParseBlock;
// the outline's payload spans against the creator messages; the device's signature gate on every signature of the mutant.
static bool two_run(const uint8_t* block, size_t len, const ParsedBlock& host) {
    ParsedBlock out;
    std::vector<uint32_t> env, pay;
    std::vector<BlockTuple> sigs;
    if (!OutlineBlock(block, len, out, env, sigs, &pay)) { printf("OUTLINE REFUSES WHAT PARSEBLOCK TAKES\n"); return false; }
    const uint32_t ne = (uint32_t)(env.size() / 2);
    struct C { uint32_t t, p, c; uint64_t g; };
    std::vector<C> cnt(ne), base(ne);
    std::vector<uint32_t> cb(ne);
    C run = {0, 0, 0, 0};
    uint32_t ncre = 0;
    for (uint32_t e = 0; e < ne; e++) {
        if ((size_t)env[2 * e] + env[2 * e + 1] > len) { printf("ENVELOPE SPAN OUT OF RANGE\n"); return false; }
        walk::CountEmitter em;
        uint8_t ty, un;
        walk::walk_envelope(block, block + env[2 * e], env[2 * e + 1], e, em, ty, un);
        if (ty != host.tx_type[e] || un != host.tx_understood[e]) { printf("COUNT RUN DISAGREES WITH PARSEBLOCK\n"); return false; }
        cnt[e] = {em.nt, em.np, em.nc, em.gb};
        base[e] = run;
        cb[e] = ncre;
        run.t += em.nt; run.p += em.np; run.c += em.nc; run.g += em.gb;
        if (em.nt) ncre++;
    }
    if (run.t + sigs.size() != host.tuples.size() || run.p != host.prefixes.size() || run.c != host.hash_checks.size()) { printf("TOTALS DIFFER\n"); return false; }
    // exact-size heap arrays (malloc, not vectors: no slack)
    BlockTuple* tuples = (BlockTuple*)malloc(sizeof(BlockTuple) * run.t + 1);
    uint32_t* pre = (uint32_t*)malloc(8 * (size_t)run.p + 1);
    BlockHashCheck* chk = (BlockHashCheck*)malloc(sizeof(BlockHashCheck) * run.c + 1);
    uint32_t* gsp = (uint32_t*)malloc(24 * (size_t)run.c + 1);
    uint32_t* gof = (uint32_t*)malloc(4 * (size_t)run.c + 4);
    uint32_t* csp = (uint32_t*)malloc(8 * (size_t)ncre + 1);
    bool ok = true;
    for (uint32_t e = 0; e < ne && ok; e++) {
        if (!cnt[e].t && !cnt[e].p && !cnt[e].c) continue;
        walk::WriteEmitter em{tuples, pre, chk, gsp, gof, base[e].t, base[e].p, base[e].c, (uint32_t)base[e].g, cnt[e].t, cnt[e].p, cnt[e].c, csp, cb[e]};
        uint8_t ty, un;
        walk::walk_envelope(block, block + env[2 * e], env[2 * e + 1], e, em, ty, un);
        if (em.nt != cnt[e].t || em.np != cnt[e].p || em.nc != cnt[e].c) { printf("THE TWO RUNS DISAGREE\n"); ok = false; }
    }
    for (size_t i = 0; ok && i < run.t; i++)
        if (memcmp(&tuples[i], &host.tuples[i], sizeof(BlockTuple)) != 0) { printf("TUPLE %zu DIFFERS\n", i); ok = false; }
    for (uint32_t e = 0, c = 0; ok && e < ne; e++) {
        if (!cnt[e].t) continue;
        if (pay[2 * e] != csp[2 * c] || pay[2 * e + 1] != csp[2 * c + 1]) { printf("PAYLOAD SPAN OF ENVELOPE %u DIFFERS\n", e); ok = false; }
        c++;
    }
    for (size_t i = 0; ok && i < host.tuples.size(); i++) {
        const BlockTuple& t = host.tuples[i];
        uint8_t r[32], s2[32];
        if ((size_t)t.sig.off + t.sig.len <= len) {
            uint8_t* sig = (uint8_t*)malloc(t.sig.len ? t.sig.len : 1);          // exact-size copy: the gate must not read past the signature
            memcpy(sig, block + t.sig.off, t.sig.len);
            (void)walk::gate_sig_fast(sig, t.sig.len, r, s2);
            // ... and the gate the device route applies to EVERY signature (round 3): the general parser for what the fast gate declines,
            // on the same exact-size copy; a submit must come with magnitudes inside the signature
            uint32_t pr, lr, ps, ls;
            const uint8_t g = walk::gate_sig_general(sig, t.sig.len, pr, lr, ps, ls);
            if (g == walk::GATE_SUBMIT && ((size_t)pr + lr > t.sig.len || (size_t)ps + ls > t.sig.len || lr > 32 || ls > 32 || !lr || !ls)) { printf("GENERAL GATE SPAN\n"); ok = false; }
            (void)walk::gate_sig_any(sig, t.sig.len, r, s2);
            // the idemix parsers the gate kernel shares with the host: same answer as the host's own
            {
                const uint8_t* sf4[4];
                const bool dev_ok = walk::unmarshal_nym_signature32(sig, t.sig.len, sf4);
                NymSignatureFields hf;
                const bool host_ok = UnmarshalNymSignature(sig, t.sig.len, hf) && hf.len[0] == 32 && hf.len[1] == 32 && hf.len[2] == 32 && hf.len[3] == 32;
                if (dev_ok != host_ok) { printf("NYM SIGNATURE PARSERS DISAGREE\n"); ok = false; }
                for (int q = 0; ok && dev_ok && q < 4; q++)
                    if (sf4[q] != hf.f[q]) { printf("NYM SIGNATURE FIELD %d DIFFERS\n", q); ok = false; }
            }
            free(sig);
        }
        (void)walk::id_hash_host(block + t.identity.off, t.identity.len);
        if ((size_t)t.identity.off + t.identity.len <= len) {
            uint8_t* idc = (uint8_t*)malloc(t.identity.len ? t.identity.len : 1);   // exact-size copy of the identity
            memcpy(idc, block + t.identity.off, t.identity.len);
            walk::IdemixNymRef ref;
            const bool is_nym = walk::identity_to_idemix_nym(idc, t.identity.len, ref);
            if (is_nym && (ref.mspid < idc || ref.mspid + ref.mspid_len > idc + t.identity.len || ref.nx < idc || ref.nx + 32 > idc + t.identity.len ||
                           ref.ny < idc || ref.ny + 32 > idc + t.identity.len)) { printf("IDEMIX IDENTITY SPAN\n"); ok = false; }
            (void)walk::id_hash_host(idc, t.identity.len, 0x1234567ull);
            free(idc);
        }
    }
    free(tuples); free(pre); free(chk); free(gsp); free(gof); free(csp);
    return ok;
}

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
            if (!two_run(heap, b.size(), pb)) return 1;
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
    printf("fuzz ok: %zu of 20000 mutants parsed, %zu tuples (%zu orderer block signatures); the device walk's two-run procedure, the outline and the\n"
           "signature gate ran on every parsed mutant with exact-size arrays and agreed with ParseBlock\n", parsed, tuples, blocksigs);
}
