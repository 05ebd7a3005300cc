// TEST HOOKS - built into libfabgpu_testhooks.so, never into the product library libfabgpu.so (the product's C ABI
// carries no probe / compare / generator entry).  They reach into the library through its C++ internals (bccsp_host.h, block_walk_dev.h,
// bccsp_capi_private.h), which libfabgpu.so exports as C++ symbols like any shared library does; tests, bench.py and the tools load this
// library NEXT TO the product's.  Declarations: fabgpu_testhooks.h.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "bccsp_capi_private.h"
#include "block_walk_dev.h"
#include "fabgpu_testhooks.h"
#include "idemix_host.h"

using namespace fab::bccsp;

namespace {
void put_err(char* dst, size_t cap, const std::string& s) {
    if (!dst || cap == 0) return;
    size_t k = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(dst, s.data(), k);
    dst[k] = 0;
}
}  // namespace

extern "C" {

// Duration in milliseconds of the most recent kernel launched through ctx (FABGPU_FLAG_TIME_KERNELS contexts only)
float fabgpu_last_kernel_ms(fabgpu_ctx* ctx) { return fab::ctx_last_kernel_ms(ctx); }
// a registered key's comb table as it lies on the device, and the host builder's for the same key (both KeyTab8::TABLE_WORDS = 163 840 words)
int fabgpu_test_key_table(fabgpu_ctx* ctx, uint32_t key_id, int32_t* out_words, size_t cap_words) {
    if (cap_words < fab::key_table_words()) return FABGPU_ETOOBIG;
    return fab::key_table_copy(ctx, key_id, out_words);
}
long long fabgpu_test_gtab_compare_with_host(fabgpu_ctx* ctx) { return (long long)fab::gtab_compare_with_host(ctx); }
long long fabgpu_test_key_tables16(fabgpu_ctx* ctx, uint32_t key_id) { return (long long)fab::key_tables16_check(ctx, key_id); }
int fabgpu_test_key_table_host(const uint8_t* qx32, const uint8_t* qy32, int32_t* out_words, size_t cap_words) {
    if (cap_words < fab::key_table_words()) return FABGPU_ETOOBIG;
    return fab::key_table_build(qx32, qy32, out_words) ? FABGPU_OK : FABGPU_EINVAL;
}
// the idemix four-lane form: order the side launch behind the commitment launch (idemix_kernels.hip: both halves of the fallback)
void fabgpu_test_nym_side_after(fabgpu_ctx* ctx, int on) { fab::ctx_test_nym_side_after(ctx, on != 0); }

// TEST HOOK: the device walker against the host walker on one block.  0: identical (or *declined = 1: the device declined, nothing
// compared); 1: they differ, `diff` says where.
int fabgpu_csp_block_walk_compare(fabgpu_csp* csp, const uint8_t* block, size_t len, int* declined, char* diff, size_t cap) {
    if (!csp || !block || !declined) return FABGPU_EINVAL;
    *declined = 0;
    put_err(diff, cap, "");
    ParsedBlock host, dev;
    const bool hok = ParseBlock(block, len, host, WalkThreads());
    const char* why = "";
    const int r = csp->csp->WalkBlockOnDevice(block, len, dev, &why);
    if (r == 1) {
        *declined = 1;
        put_err(diff, cap, why);
        return FABGPU_OK;
    }
    if (!hok || r == FABGPU_EINVAL) {
        if (hok != (r != FABGPU_EINVAL)) { put_err(diff, cap, "one walker refuses the framing, the other does not"); return 1; }
        return FABGPU_OK;
    }
    if (r < 0) return r;
    std::string d;
    auto span_eq = [](const Span& a, const Span& b) { return a.off == b.off && a.len == b.len; };
    if (host.n_tx != dev.n_tx) d = "n_tx";
    else if (host.tx_type != dev.tx_type) d = "tx_type";
    else if (host.tx_understood != dev.tx_understood) d = "tx_understood";
    else if (host.tuples.size() != dev.tuples.size()) d = "tuple count " + std::to_string(host.tuples.size()) + " vs " + std::to_string(dev.tuples.size());
    else if (host.prefixes.size() != dev.prefixes.size()) d = "prefix count";
    else if (host.hash_checks.size() != dev.hash_checks.size()) d = "hash check count";
    for (size_t i = 0; d.empty() && i < host.tuples.size(); i++) {
        const BlockTuple &a = host.tuples[i], &b = dev.tuples[i];
        if (a.tx != b.tx || a.kind != b.kind || a.prefix_index != b.prefix_index || !span_eq(a.identity, b.identity) || !span_eq(a.prefix, b.prefix) ||
            !span_eq(a.suffix, b.suffix) || !span_eq(a.sig, b.sig))
            d = "tuple " + std::to_string(i);
    }
    for (size_t i = 0; d.empty() && i < host.prefixes.size(); i++) {
        // (the device keeps (start, end) pairs and normalises an empty prefix to (0, 0))
        const Span a = host.prefixes[i], b = dev.prefixes[i];
        if (a.len != b.len || (a.len && a.off != b.off)) d = "prefix " + std::to_string(i);
    }
    for (size_t i = 0; d.empty() && i < host.hash_checks.size(); i++) {
        const BlockHashCheck &a = host.hash_checks[i], &b = dev.hash_checks[i];
        if (a.tx != b.tx || a.kind != b.kind || !span_eq(a.piece[0], b.piece[0]) || !span_eq(a.piece[1], b.piece[1]) || !span_eq(a.piece[2], b.piece[2]) ||
            !span_eq(a.expect, b.expect))
            d = "hash check " + std::to_string(i);
    }
    put_err(diff, cap, d);
    return d.empty() ? FABGPU_OK : 1;
}


// TEST HOOK (pure host): the device walk's two-run procedure - count per envelope, exclusive prefix sum, write at the assigned offsets
// (block_walk_core.h CountEmitter / WriteEmitter, the code block_walk_kernels.hip runs) - carried out serially on the host and
// compared, record for record, with ParseBlock.  0 identical, 1 different (`diff` says where), FABGPU_EINVAL: the framing is refused
// (by both).
int fabgpu_block_walk_twopass_compare(const uint8_t* block, size_t len, char* diff, size_t cap) {
    if (!block) return FABGPU_EINVAL;
    put_err(diff, cap, "");
    ParsedBlock host, out;
    std::vector<uint32_t> env, pay;
    std::vector<BlockTuple> sigs;
    const bool hok = ParseBlock(block, len, host, 1);
    const bool ook = OutlineBlock(block, len, out, env, sigs, &pay);
    if (hok != ook) { put_err(diff, cap, "framing verdicts differ"); return 1; }
    if (!hok) return FABGPU_EINVAL;
    const uint32_t ne = (uint32_t)(env.size() / 2);
    struct C { uint32_t t, p, c; uint64_t g; };
    std::vector<C> cnt(ne), base(ne);
    std::vector<uint8_t> type(ne), und(ne);
    // (as the kernels: the counting run keeps each envelope's records in its slot, the second run copies them - or, for an envelope with
    //  more records than a slot holds, walks again)
    std::vector<walk::EnvStash> stash(ne);
    for (uint32_t e = 0; e < ne; e++) {
        walk::StashEmitter em{&stash[e]};
        walk::walk_envelope(block, block + env[2 * e], env[2 * e + 1], e, em, type[e], und[e]);
        cnt[e] = {em.nt, em.np, em.nc, em.gb};
        stash[e].over = em.fits() ? 0u : 1u;
    }
    C run = {0, 0, 0, 0};
    for (uint32_t e = 0; e < ne; e++) {
        base[e] = run;
        run.t += cnt[e].t; run.p += cnt[e].p; run.c += cnt[e].c; run.g += cnt[e].g;
    }
    std::vector<BlockTuple> tuples(run.t + 1);
    std::vector<uint32_t> pre_off2(2 * (size_t)run.p + 2), gsp(6 * (size_t)run.c + 6), goff(run.c + 1);
    std::vector<BlockHashCheck> checks(run.c + 1);
    // canaries: a record written outside its envelope's range would land on one
    BlockTuple canary;
    canary.tx = 0xDEADBEEF;
    std::fill(tuples.begin(), tuples.end(), canary);
    std::vector<uint32_t> cspans(2 * (size_t)ne + 2, 0xDEADBEEFu);     // the creators' message spans, in creator order
    uint32_t ncre = 0;
    for (uint32_t e = 0; e < ne; e++) {
        if (cnt[e].t == 0 && cnt[e].p == 0 && cnt[e].c == 0) continue;
        walk::WriteEmitter em{tuples.data(), pre_off2.data(), checks.data(), gsp.data(), goff.data(), base[e].t, base[e].p, base[e].c, (uint32_t)base[e].g,
                              cnt[e].t, cnt[e].p, cnt[e].c, cspans.data(), ncre};
        if (cnt[e].t) ncre++;
        if (stash[e].over == 0) {
            for (uint32_t k = 0; k < cnt[e].p; k++) em.add_prefix(stash[e].p[k]);
            for (uint32_t k = 0; k < cnt[e].t; k++) {
                BlockTuple t = stash[e].t[k];
                if (t.prefix_index >= 0) t.prefix_index += (int32_t)base[e].p;
                em.add_tuple(t);
            }
            for (uint32_t k = 0; k < cnt[e].c; k++) em.add_check(stash[e].c[k]);
            continue;
        }
        uint8_t t2, u2;
        walk::walk_envelope(block, block + env[2 * e], env[2 * e + 1], e, em, t2, u2);
        if (t2 != type[e] || u2 != und[e] || em.nt != cnt[e].t || em.np != cnt[e].p || em.nc != cnt[e].c) { put_err(diff, cap, "the two runs disagree on envelope " + std::to_string(e)); return 1; }
    }
    std::string d;
    auto span_eq = [](const Span& a, const Span& b) { return a.off == b.off && a.len == b.len; };
    const size_t host_env_tuples = host.tuples.size() - host.n_block_sigs;
    if (host.n_tx != ne) d = "n_tx";
    else if (memcmp(host.tx_type.data(), type.data(), ne) != 0) d = "tx_type";
    else if (memcmp(host.tx_understood.data(), und.data(), ne) != 0) d = "tx_understood";
    else if (host_env_tuples != run.t) d = "tuple count";
    else if (host.prefixes.size() != run.p) d = "prefix count";
    else if (host.hash_checks.size() != run.c) d = "hash check count";
    else if (sigs.size() != host.n_block_sigs) d = "block signature count";
    for (size_t i = 0; d.empty() && i < host.tuples.size(); i++) {
        const BlockTuple& a = host.tuples[i];
        const BlockTuple& b = i < host_env_tuples ? tuples[i] : sigs[i - host_env_tuples];
        if (a.tx != b.tx || a.kind != b.kind || a.prefix_index != b.prefix_index || !span_eq(a.identity, b.identity) || !span_eq(a.prefix, b.prefix) ||
            !span_eq(a.suffix, b.suffix) || !span_eq(a.sig, b.sig))
            d = "tuple " + std::to_string(i);
    }
    if (d.empty() && tuples[run.t].tx != 0xDEADBEEF) d = "a tuple was written past the end";
    {   // creator spans: one per envelope that yields tuples, equal to the suffix of that envelope's first tuple
        uint32_t k = 0;
        for (size_t i = 0; d.empty() && i < host_env_tuples; i++) {
            if (host.tuples[i].kind != TUPLE_CREATOR) continue;
            const Span sx = host.tuples[i].suffix;
            if (cspans[2 * (size_t)k + 1] - cspans[2 * (size_t)k] != sx.len || (sx.len && cspans[2 * (size_t)k] != sx.off)) d = "creator span " + std::to_string(k);
            k++;
        }
        if (d.empty() && (k != ncre || cspans[2 * (size_t)k] != 0xDEADBEEFu)) d = "creator span count";
        // the outline's payload span of an envelope that yields tuples IS its creator's message (the device hashes it before it has
        // walked anything, and its gate kernel insists on exactly this equality)
        if (d.empty() && pay.size() != 2 * (size_t)ne) d = "payload span list";
        uint32_t c = 0;
        for (uint32_t e = 0; d.empty() && e < ne; e++) {
            if (!cnt[e].t) continue;
            if (pay[2 * (size_t)e] != cspans[2 * (size_t)c] || pay[2 * (size_t)e + 1] != cspans[2 * (size_t)c + 1]) d = "payload span of envelope " + std::to_string(e);
            c++;
        }
    }
    for (size_t i = 0; d.empty() && i < host.prefixes.size(); i++) {
        const Span a = host.prefixes[i];
        if (pre_off2[2 * i + 1] - pre_off2[2 * i] != a.len || (a.len && pre_off2[2 * i] != a.off)) d = "prefix " + std::to_string(i);
    }
    uint64_t g = 0;
    for (size_t i = 0; d.empty() && i < host.hash_checks.size(); i++) {
        const BlockHashCheck &a = host.hash_checks[i], &b = checks[i];
        if (a.tx != b.tx || a.kind != b.kind || !span_eq(a.piece[0], b.piece[0]) || !span_eq(a.piece[1], b.piece[1]) || !span_eq(a.piece[2], b.piece[2]) ||
            !span_eq(a.expect, b.expect))
            d = "hash check " + std::to_string(i);
        if (d.empty() && goff[i] != (uint32_t)g) d = "gather offset " + std::to_string(i);
        for (int p = 0; d.empty() && p < 3; p++) {
            if (gsp[6 * i + 2 * p + 1] - gsp[6 * i + 2 * p] != a.piece[p].len || (a.piece[p].len && gsp[6 * i + 2 * p] != a.piece[p].off)) d = "gather span " + std::to_string(i);
            g += a.piece[p].len;
        }
    }
    if (d.empty() && g != run.g) d = "gathered bytes";
    put_err(diff, cap, d);
    return d.empty() ? FABGPU_OK : 1;
}
// TEST HOOK (pure host): the device's signature gate (block_walk_core.h gate_sig_fast): 0 submit (r32 / s32 set), 1 high-S, 2 empty, 3 declined
int fabgpu_gate_sig_fast(const uint8_t* sig, size_t len, uint8_t* r32, uint8_t* s32) {
    uint8_t r[32], s[32];
    if (len > 0xFFFFFFFFull) return walk::GATE_DECLINED;
    const uint8_t g = walk::gate_sig_fast(sig, (uint32_t)len, r, s);
    if (g == walk::GATE_SUBMIT) {
        if (r32) memcpy(r32, r, 32);
        if (s32) memcpy(s32, s, 32);
    }
    return g;
}
// TEST HOOK (pure host): the gate the device route applies to EVERY signature (block_walk_core.h gate_sig_any = the fast gate, then the
// general parser for what it declines): 0 submit (r32 / s32 set), 1 high-S, 2 empty, 4 does not unmarshal / r, s <= 0, 5 r beyond 256 bits
int fabgpu_gate_sig_any(const uint8_t* sig, size_t len, uint8_t* r32, uint8_t* s32) {
    uint8_t r[32], s[32];
    if (len > 0xFFFFFFFFull) return walk::GATE_BAD_DER;
    const uint8_t g = walk::gate_sig_any(sig, (uint32_t)len, r, s);
    if (g == walk::GATE_SUBMIT) {
        if (r32) memcpy(r32, r, 32);
        if (s32) memcpy(s32, s, 32);
    }
    return g;
}

// TEST HOOK (device): the identity decoder of the device route (certificate -> key by one wavefront) over n identities
int fabgpu_csp_idfix_probe(fabgpu_csp* csp, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* key) {
    if (!csp) return FABGPU_EINVAL;
    return fab::walk_idfix_probe(csp->csp->ctx(), n, arena, arena_len, spans, code, key);
}
// TEST HOOK (device): the wavefront form of the same gate, as block_walk_kernels.hip runs it, over n signatures
int fabgpu_csp_gate_probe(fabgpu_csp* csp, uint32_t n, const uint8_t* arena, size_t arena_len, const uint32_t* spans, uint8_t* code, uint8_t* r, uint8_t* s) {
    if (!csp) return FABGPU_EINVAL;
    return fab::walk_gate_probe(csp->csp->ctx(), n, arena, arena_len, spans, code, r, s);
}
// TEST HOOK (pure host): the table hash of identity bytes (block_walk_core.h id_hash_host)
uint64_t fabgpu_identity_table_hash(const uint8_t* p, size_t len) { return walk::id_hash_host(p, (uint32_t)len); }


}  // extern "C"
