// Flat C wrappers (include/fabgpu_bccsp.h) over the C++ host mirror, for ctypes / other FFIs.
#include <stdio.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "block_walk_dev.h"

#include <chrono>

#include "../../include/fabgpu_bccsp.h"
#include "bccsp_host.h"
#include "bccsp_capi_private.h"
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

int fabgpu_csp_new(const fabgpu_cfg* cfg, fabgpu_csp** out, char* err, size_t errcap) {
    if (!out) return FABGPU_EINVAL;
    *out = nullptr;
    fabgpu_csp* h = new fabgpu_csp();
    Error e = GPUCSP::New(cfg, h->csp);
    if (!e.ok()) {
        put_err(err, errcap, e.msg);
        delete h;
        return FABGPU_ENODEV;
    }
    put_err(err, errcap, "");
    *out = h;
    return FABGPU_OK;
}
// One provider over several devices: what bccsp/factory builds from the `GPU:` section (go/bccsp/factory/gpufactory.go GPUOpts).
int fabgpu_csp_new2(const fabgpu_csp_opts* o, fabgpu_csp** out, char* err, size_t errcap) {
    if (!out) return FABGPU_EINVAL;
    *out = nullptr;
    ProviderOptions po;
    if (o) {
        if (o->size < sizeof(uint32_t) * 2 || o->n_devices < 0 || o->n_devices > kMaxProviderDevices) return FABGPU_EINVAL;
        fabgpu_csp_opts v;
        memset(&v, 0, sizeof(v));                                      // (0 = the default of every switch)
        memcpy(&v, o, o->size < sizeof(v) ? o->size : sizeof(v));     // (a caller built against an older, shorter struct)
        for (int i = 0; i < v.n_devices; i++) po.devices.push_back(v.devices ? v.devices[i] : i);
        po.ctx_flags = v.ctx_flags;
        po.concurrent_passes = v.concurrent_passes;
        po.expect_block_bytes = v.expect_block_bytes;
        po.expect_tuples = v.expect_tuples;
        po.pass_stage_min_bytes = v.pass_stage_min_bytes;
        po.pass_device_walk = v.pass_device_walk;
        po.pass_device_memo = v.pass_device_memo;
        po.pass_host_counts = v.pass_host_counts;
        po.pass_timing = v.pass_timing;
        po.pass_hash_memo = v.pass_hash_memo;
        po.hash_memo_blocks = v.hash_memo_blocks;
    }
    fabgpu_csp* h = new fabgpu_csp();
    Error e = GPUCSP::New(po, h->csp);
    if (!e.ok()) {
        put_err(err, errcap, e.msg);
        delete h;
        return FABGPU_ENODEV;
    }
    put_err(err, errcap, "");
    *out = h;
    return FABGPU_OK;
}
void fabgpu_csp_free(fabgpu_csp* csp) {
    if (csp) csp->orphan.reset();                            // (an upload refers to the provider: it goes first)
    delete csp;
}
fabgpu_ctx* fabgpu_csp_ctx(fabgpu_csp* csp) { return csp ? csp->csp->ctx() : nullptr; }
int fabgpu_csp_device_count(fabgpu_csp* csp) { return csp ? csp->csp->n_devices() : FABGPU_EINVAL; }
fabgpu_ctx* fabgpu_csp_ctx_of(fabgpu_csp* csp, int d) { return csp && d >= 0 && d < csp->csp->n_devices() ? csp->csp->ctx_of(d) : nullptr; }
int fabgpu_csp_passes_per_device(fabgpu_csp* csp, uint64_t* passes, int cap) {
    if (!csp || !passes || cap < csp->csp->n_devices()) return FABGPU_EINVAL;
    csp->csp->PassesPerDevice(passes);
    return csp->csp->n_devices();
}
int fabgpu_csp_route_block(fabgpu_csp* csp, uint64_t block_seq) { return csp ? csp->csp->RouteBlock(block_seq) : FABGPU_EINVAL; }
int fabgpu_csp_set_option(fabgpu_csp* csp, const char* name, int64_t value, int64_t* previous) {
    if (!csp || !name) return FABGPU_EINVAL;
    const int64_t prev = csp->csp->SetOption(name, value);
    if (prev == INT64_MIN) return FABGPU_EINVAL;
    if (previous) *previous = prev;
    return FABGPU_OK;
}
int fabgpu_csp_get_option(fabgpu_csp* csp, const char* name, int64_t* value) {
    if (!csp || !name || !value) return FABGPU_EINVAL;
    const int64_t v = csp->csp->GetOption(name);
    if (v == INT64_MIN) return FABGPU_EINVAL;
    *value = v;
    return FABGPU_OK;
}

int fabgpu_csp_key_import(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, int* on_curve, char* err, size_t errcap) {
    if (!csp) return FABGPU_EINVAL;
    ECDSAPublicKey k;
    Error e = csp->csp->KeyImport(qx32, qy32, k, true);
    put_err(err, errcap, e.ok() ? "" : e.msg);
    if (on_curve) *on_curve = k.on_curve ? 1 : 0;
    return FABGPU_OK;
}

// bccsp.Hash(msg, &bccsp.SHA256Opts{}) answered from the digest memo: 0 hit (digest32 filled), 1 miss (hash on the CPU)
int fabgpu_csp_hash_lookup(fabgpu_csp* csp, const uint8_t* msg, size_t len, uint8_t* digest32) {
    if (!csp || !msg || !digest32) return 1;
    return csp->csp->HashLookup(msg, len, digest32);
}
int fabgpu_csp_hash_memo_stats(fabgpu_csp* csp, uint64_t* hits, uint64_t* misses, uint64_t* blocks_held, uint64_t* bytes_held, uint64_t* refused) {
    if (!csp) return FABGPU_EINVAL;
    csp->csp->HashMemoStats(hits, misses, blocks_held, bytes_held, refused);
    return FABGPU_OK;
}
int fabgpu_csp_hash(fabgpu_csp* csp, const uint8_t* msg, size_t len, const char* alg, uint8_t* digest32, char* err, size_t errcap) {
    if (!csp || !digest32) return FABGPU_EINVAL;
    HashOpts o;
    if (alg) o.algorithm = alg;
    std::vector<uint8_t> d;
    Error e = csp->csp->Hash(msg, len, alg ? &o : nullptr, d);
    put_err(err, errcap, e.ok() ? "" : e.msg);
    if (e.ok()) memcpy(digest32, d.data(), 32);
    return FABGPU_OK;
}

int fabgpu_csp_verify(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen,
                      const uint8_t* digest, size_t dlen, int* valid, int* flags, char* err, size_t errcap) {
    if (!csp || !valid) return FABGPU_EINVAL;
    ECDSAPublicKey k;
    const ECDSAPublicKey* kp = nullptr;
    if (qx32 && qy32) {
        csp->csp->KeyImport(qx32, qy32, k);
        kp = &k;
    }
    VerifyResult r = csp->csp->Verify(kp, sig, siglen, digest, dlen);
    if (r.infrastructure) {
        put_err(err, errcap, r.err.msg);
        return FABGPU_ELAUNCH;
    }
    *valid = r.valid ? 1 : 0;
    if (flags) *flags = r.needs_sw ? 1 : 0;
    put_err(err, errcap, r.err.ok() ? "" : r.err.msg);
    return FABGPU_OK;
}

int fabgpu_csp_verify_coalesced(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen,
                                const uint8_t* digest, size_t dlen, int* valid, int* flags, char* err, size_t errcap) {
    if (!csp || !valid) return FABGPU_EINVAL;
    ECDSAPublicKey k;
    const ECDSAPublicKey* kp = nullptr;
    if (qx32 && qy32) {
        csp->csp->KeyImport(qx32, qy32, k);
        kp = &k;
    }
    VerifyResult r = csp->csp->VerifyCoalesced(kp, sig, siglen, digest, dlen);
    if (r.infrastructure) {
        put_err(err, errcap, r.err.msg);
        return FABGPU_ELAUNCH;
    }
    *valid = r.valid ? 1 : 0;
    if (flags) *flags = r.needs_sw ? 1 : 0;
    put_err(err, errcap, r.err.ok() ? "" : r.err.msg);
    return FABGPU_OK;
}

int fabgpu_csp_identity_verify_coalesced(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* msg, size_t msglen,
                                         const uint8_t* sig, size_t siglen, char* err, size_t errcap) {
    if (!csp || (msglen && !msg)) return FABGPU_EINVAL;
    ECDSAPublicKey k;
    const ECDSAPublicKey* kp = nullptr;
    if (qx32 && qy32) {
        csp->csp->KeyImport(qx32, qy32, k);
        kp = &k;
    }
    bool infra = false;
    std::string out = csp->csp->IdentityVerifyCoalesced(kp, msg, msglen, sig, siglen, &infra);
    put_err(err, errcap, out);
    return infra ? FABGPU_ELAUNCH : FABGPU_OK;
}

int fabgpu_csp_coalescer_configure(fabgpu_csp* csp, uint32_t window_us, uint32_t max_batch) {
    if (!csp) return FABGPU_EINVAL;
    csp->csp->CoalescerConfigure(window_us, max_batch);
    return FABGPU_OK;
}

int fabgpu_csp_coalescer_stats(fabgpu_csp* csp, uint64_t* calls, uint64_t* launches, uint64_t* largest_batch) {
    if (!csp) return FABGPU_EINVAL;
    csp->csp->CoalescerStats(calls, launches, largest_batch);
    return FABGPU_OK;
}

int fabgpu_csp_verify_batch(fabgpu_csp* csp, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* sig_arena,
                            const uint32_t* sig_off, const uint8_t* dig_arena, const uint32_t* dig_off, uint8_t* valid,
                            char* errs, size_t errstride) {
    if (!csp || (n && (!qx || !qy || !sig_off || !dig_off || !valid))) return FABGPU_EINVAL;
    std::vector<ECDSAPublicKey> keys(n);
    std::vector<VerifyItem> items(n);
    for (size_t i = 0; i < n; i++) {
        csp->csp->KeyImport(qx + 32 * i, qy + 32 * i, keys[i]);
        items[i] = {&keys[i], sig_arena + sig_off[i], sig_off[i + 1] - sig_off[i], dig_arena + dig_off[i], dig_off[i + 1] - dig_off[i]};
    }
    std::vector<VerifyResult> res;
    Error e = csp->csp->VerifyBatch(items, res);
    if (!e.ok()) return FABGPU_ELAUNCH;
    for (size_t i = 0; i < n; i++) {
        valid[i] = res[i].valid ? 1 : 0;
        if (errs && errstride) put_err(errs + i * errstride, errstride, res[i].err.ok() ? "" : res[i].err.msg);
    }
    return FABGPU_OK;
}

int fabgpu_csp_identity_verify_batch(fabgpu_csp* csp, size_t n, const uint8_t* qx, const uint8_t* qy, const uint8_t* msg_arena,
                                     const uint32_t* msg_off, const uint8_t* sig_arena, const uint32_t* sig_off, char* errs,
                                     size_t errstride) {
    if (!csp || (n && (!qx || !qy || !msg_off || !sig_off || !errs || !errstride))) return FABGPU_EINVAL;
    std::vector<ECDSAPublicKey> keys(n);
    std::vector<IdentityItem> items(n);
    for (size_t i = 0; i < n; i++) {
        csp->csp->KeyImport(qx + 32 * i, qy + 32 * i, keys[i]);
        items[i] = {&keys[i], msg_arena + msg_off[i], msg_off[i + 1] - msg_off[i], sig_arena + sig_off[i], sig_off[i + 1] - sig_off[i]};
    }
    std::vector<std::string> out;
    Error e = csp->csp->IdentityVerifyBatch(items, out);
    if (!e.ok()) return FABGPU_ELAUNCH;
    for (size_t i = 0; i < n; i++) put_err(errs + i * errstride, errstride, out[i]);
    return FABGPU_OK;
}

int fabgpu_csp_block_preverify(fabgpu_csp* csp, const uint8_t* block, size_t len, uint32_t* n_tx, uint8_t* tx_flags, uint8_t* tx_type,
                               uint32_t cap_tx, uint32_t* n_tuples, uint32_t* tuple_tx, uint8_t* tuple_kind, uint8_t* tuple_status,
                               uint32_t cap_tuples) {
    if (!csp || !block || !n_tx || !n_tuples) return FABGPU_EINVAL;
    const bool timing = csp->csp->GetOption("pass_timing") > 0;           // stage breakdown on stderr (tools/bench_block.py --timing)
    auto t0 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    std::unique_ptr<GPUCSP::BlockUpload> up_p = csp->upload_for(block, len, 0, false);   // the block travels while it is walked
    GPUCSP::BlockUpload& up = *up_p;
    const bool per_tuple = tuple_tx != nullptr || tuple_kind != nullptr || tuple_status != nullptr;
    static thread_local ParsedBlock pb;                     // storage reused from block to block (a few MB: no page faults per block)
    static thread_local BlockVerdicts v;                    // answer arrays keep their capacity from block to block
    bool done = false;
    {   // the walk on the device (block_walk_dev.h); a block it declines takes the host walk below
        const char* why = "";
        uint32_t ntup = 0;
        // (room for per-tuple answers only matters to a caller that asked for some)
        const int r = csp->csp->PreVerifyBlockOnDevice(block, len, pb, v, up, PassOptions(), tuple_tx != nullptr || tuple_kind != nullptr ? GPUCSP::WANT_TUPLES : 0u, cap_tx,
                                                       per_tuple ? cap_tuples : 0xFFFFFFFFu, &ntup, &why);
        if (r != FABGPU_ETOOBIG) csp->note_route(r == 0, why);
        if (r == FABGPU_ETOOBIG) {
            *n_tx = pb.n_tx;
            *n_tuples = ntup;
            csp->park(std::move(up_p));                      // the retry finds its upload again
            return FABGPU_ETOOBIG;
        }
        if (r < 0) return r == FABGPU_EINVAL || r == FABGPU_ENOMEM ? r : FABGPU_ELAUNCH;
        done = r == 0;
        if (timing && done)
            fprintf(stderr, "fabgpu pass (device walk): total %.2f ms (outline + identity table %.2f, wait for upload %.2f, device %.2f, bookkeeping %.2f)\n",
                    ms(t0, std::chrono::steady_clock::now()), v.ms_gates, v.ms_upload_wait, v.ms_device, v.ms_post);
        if (timing && !done) fprintf(stderr, "fabgpu pass: device walk declined (%s)\n", why);
    }
    if (done) {
        *n_tx = pb.n_tx;
        *n_tuples = (uint32_t)v.tuple_status.size();
    } else {
        if (!ParseBlock(block, len, pb, WalkThreads())) return FABGPU_EINVAL;
        auto t1 = std::chrono::steady_clock::now();
        *n_tx = pb.n_tx;
        *n_tuples = (uint32_t)pb.tuples.size();
        if (pb.n_tx > cap_tx || (per_tuple && pb.tuples.size() > cap_tuples)) {          // counts are set: retry with room (nothing was launched)
            csp->park(std::move(up_p));
            return FABGPU_ETOOBIG;
        }
        Error e = csp->csp->PreVerifyParsed(block, pb, v, &up);
        if (timing) {
            auto t2 = std::chrono::steady_clock::now();
            fprintf(stderr, "fabgpu pass: walk %.2f ms, gates + submission + flags %.2f ms (gates %.2f, wait for upload %.2f, device call %.2f, idemix creators %.2f, memo %.2f)\n", ms(t0, t1),
                    ms(t1, t2), v.ms_gates, v.ms_upload_wait, v.ms_device, v.ms_nym, v.ms_memo);
        }
        if (!e.ok()) return FABGPU_ELAUNCH;
    }
    if (tx_flags && v.n_tx) memcpy(tx_flags, v.tx_flags.data(), v.n_tx);
    if (tx_type && v.n_tx) memcpy(tx_type, v.tx_type.data(), v.n_tx);
    size_t nt = v.tuple_status.size();
    if (tuple_tx && nt) memcpy(tuple_tx, v.tuple_tx.data(), nt * 4);
    if (tuple_kind && nt) memcpy(tuple_kind, v.tuple_kind.data(), nt);
    if (tuple_status && nt) memcpy(tuple_status, v.tuple_status.data(), nt);
    return FABGPU_OK;
}

// A caller that gives up after FABGPU_ETOOBIG (no retry will come) drops the upload the library kept for it.  1: one was dropped, 0: none.
int fabgpu_csp_block_pass_abandon(fabgpu_csp* csp) { return csp ? (csp->abandon() ? 1 : 0) : FABGPU_EINVAL; }

int fabgpu_csp_block_preverify2(fabgpu_csp* csp, fabgpu_block_pass* ps) {
    if (!csp || !ps || !ps->block) return FABGPU_EINVAL;
    if (ps->flags & ~(uint32_t)(FABGPU_PASS_SEED_MEMO | FABGPU_PASS_NO_BLOCK_SIGS)) return FABGPU_EINVAL;
    auto t0 = std::chrono::steady_clock::now();
    const bool timing = csp->csp->GetOption("pass_timing") > 0;
    // (a memo-seeding pass keeps the block's bytes in host memory of the device context: the digest memo compares bccsp.Hash callers' bytes with them)
    std::unique_ptr<GPUCSP::BlockUpload> up_p = csp->upload_for(ps->block, ps->len, ps->block_seq, (ps->flags & FABGPU_PASS_SEED_MEMO) != 0);
    GPUCSP::BlockUpload& up = *up_p;
    // room for per-tuple answers only matters to a caller that asked for some (the Go binding asks for flags alone)
    const bool per_tuple = ps->tuple_tx || ps->tuple_kind || ps->tuple_status || ps->tuple_spans || ps->tuple_digest || ps->tuple_hashed || ps->tuple_qxy;
    const uint32_t cap_tuples = per_tuple ? ps->cap_tuples : 0xFFFFFFFFu;
    static thread_local ParsedBlock pb;
    static thread_local BlockVerdicts v;                    // answer arrays keep their capacity from block to block
    PassOptions opt;
    opt.seed_memo = (ps->flags & FABGPU_PASS_SEED_MEMO) != 0;
    opt.want_digests = ps->tuple_digest != nullptr;
    opt.block_sigs = !(ps->flags & FABGPU_PASS_NO_BLOCK_SIGS);
    opt.block_seq = ps->block_seq;
    if (ps->tail) opt.tail_cap = ps->tail_cap;
    ps->memo_seeded = 0;
    ps->n_keyed = 0;
    ps->n_device_decoded = 0;
    bool done = false;
    {   // the walk on the device (block_walk_dev.h); a block it declines takes the host walk below
        const char* why = "";
        uint32_t ntup = 0;
        const unsigned want = (ps->tuple_tx || ps->tuple_kind || ps->tuple_spans ? GPUCSP::WANT_TUPLES : 0u) | (ps->tuple_qxy ? GPUCSP::WANT_QXY : 0u);
        const int r = csp->csp->PreVerifyBlockOnDevice(ps->block, ps->len, pb, v, up, opt, want, ps->cap_tx, cap_tuples, &ntup, &why);
        if (r != FABGPU_ETOOBIG) csp->note_route(r == 0, why);
        if (r == 0 || r == FABGPU_ETOOBIG) {
            ps->n_tx = pb.n_tx;
            ps->n_tuples = r == 0 ? (uint32_t)v.tuple_status.size() : (ntup ? ntup : ps->cap_tuples);   // (0: not counted yet - the transactions or the tail did not fit)
            ps->n_block_sigs = pb.n_block_sigs;
            ps->tail_base = pb.tail_base;
            ps->tail_len = (uint32_t)pb.tail.size();
            ps->block_sigs_understood = pb.block_sigs_understood ? 1 : 0;
        }
        if (r == FABGPU_ETOOBIG) {
            csp->park(std::move(up_p));                      // the retry - same buffer, length and block_seq - finds its upload again
            return FABGPU_ETOOBIG;
        }
        if (r < 0) return r == FABGPU_EINVAL || r == FABGPU_ENOMEM ? r : FABGPU_ELAUNCH;
        done = r == 0;
        if (timing) {
            if (done)
                fprintf(stderr, "fabgpu pass2 (device walk): total %.2f ms (outline + identity table %.2f, wait for upload %.2f, device %.2f, memo %.2f)\n",
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), v.ms_gates, v.ms_upload_wait, v.ms_device, v.ms_memo);
            else
                fprintf(stderr, "fabgpu pass2: device walk declined (%s)\n", why);
        }
    }
    if (!done) {
        if (!ParseBlock(ps->block, ps->len, pb, WalkThreads())) return FABGPU_EINVAL;
        ps->n_tx = pb.n_tx;
        ps->n_tuples = (uint32_t)pb.tuples.size();
        ps->n_block_sigs = pb.n_block_sigs;
        ps->tail_base = pb.tail_base;
        ps->tail_len = (uint32_t)pb.tail.size();
        ps->block_sigs_understood = pb.block_sigs_understood ? 1 : 0;
        if (pb.n_tx > ps->cap_tx || pb.tuples.size() > cap_tuples || (ps->tail && pb.tail.size() > ps->tail_cap)) {
            csp->park(std::move(up_p));
            return FABGPU_ETOOBIG;
        }
        Error e = csp->csp->PreVerifyParsed(ps->block, pb, v, &up, opt);
        if (timing)
            fprintf(stderr, "fabgpu pass2: gates %.2f ms, wait for upload %.2f, device call %.2f, idemix creators %.2f, memo %.2f\n", v.ms_gates, v.ms_upload_wait,
                    v.ms_device, v.ms_nym, v.ms_memo);
        if (!e.ok()) return FABGPU_ELAUNCH;
    }
    const size_t nt = v.tuple_status.size();
    if (ps->tx_flags && v.n_tx) memcpy(ps->tx_flags, v.tx_flags.data(), v.n_tx);
    if (ps->tx_type && v.n_tx) memcpy(ps->tx_type, v.tx_type.data(), v.n_tx);
    if (ps->tuple_tx && nt) memcpy(ps->tuple_tx, v.tuple_tx.data(), nt * 4);
    if (ps->tuple_kind && nt) memcpy(ps->tuple_kind, v.tuple_kind.data(), nt);
    if (ps->tuple_status && nt) memcpy(ps->tuple_status, v.tuple_status.data(), nt);
    if (ps->tuple_hashed && nt) memcpy(ps->tuple_hashed, v.tuple_hashed.data(), nt);
    if (ps->tuple_qxy && nt) memcpy(ps->tuple_qxy, v.tuple_qxy.data(), nt * 64);
    if (ps->tuple_digest && nt) memcpy(ps->tuple_digest, v.tuple_digest.data(), nt * 32);
    if (ps->tuple_spans)
        for (size_t i = 0; i < nt; i++) {
            const BlockTuple& t = pb.tuples[i];
            const Span* sp[4] = {&t.identity, &t.prefix, &t.suffix, &t.sig};
            for (int k = 0; k < 4; k++) {
                ps->tuple_spans[8 * i + 2 * k] = sp[k]->off;
                ps->tuple_spans[8 * i + 2 * k + 1] = sp[k]->len;
            }
        }
    if (ps->tail && !pb.tail.empty()) memcpy(ps->tail, pb.tail.data(), pb.tail.size());
    ps->memo_seeded = v.memo_seeded;
    ps->n_keyed = (uint32_t)v.n_keyed;
    ps->n_device_decoded = done ? v.n_device_decoded : 0;
    ps->ms_stage[0] = (float)v.ms_gates;
    ps->ms_stage[1] = (float)v.ms_upload_wait;
    ps->ms_stage[2] = (float)v.ms_device;
    ps->ms_stage[3] = (float)(done ? v.ms_post : v.ms_memo);
    ps->device_context = up.dev;
    return FABGPU_OK;
}

// Which way the passes of this provider went: walked on the device / walked on the host, and why the last block was declined.
int fabgpu_csp_pass_routes(fabgpu_csp* csp, uint64_t* device_walks, uint64_t* host_walks, char* last_decline, size_t cap) {
    if (!csp) return FABGPU_EINVAL;
    std::lock_guard<std::mutex> lk(csp->route_mu);
    if (device_walks) *device_walks = csp->device_walks;
    if (host_walks) *host_walks = csp->host_walks;
    put_err(last_decline, cap, csp->last_decline);
    return FABGPU_OK;
}
// device-route statistics of this provider (GPUCSP::PassStats): out[0] relaunches, [1] tuples whose certificate the device decoded,
// [2] identities learned that way, [3] signatures that took the general DER parser
int fabgpu_csp_pass_stats(fabgpu_csp* csp, uint64_t* out4) {
    if (!csp || !out4) return FABGPU_EINVAL;
    csp->csp->PassStats(out4);
    return FABGPU_OK;
}
int fabgpu_csp_memo_lookup(fabgpu_csp* csp, const uint8_t* qx32, const uint8_t* qy32, const uint8_t* sig, size_t siglen, const uint8_t* digest,
                           size_t dlen, uint8_t* status) {
    if (!csp) return 1;
    return csp->csp->MemoLookup(qx32, qy32, sig, siglen, digest, dlen, status);
}
// the entry of an idemix pseudonym signature: bound to the issuer key (ipk.Hash, 32 bytes) the caller verifies under
int fabgpu_csp_memo_lookup_nym(fabgpu_csp* csp, const uint8_t* issuer_hash32, const uint8_t* nym_x32, const uint8_t* nym_y32, const uint8_t* sig, size_t siglen,
                               const uint8_t* digest, size_t dlen, uint8_t* status) {
    if (!csp || !issuer_hash32) return 1;
    return csp->csp->MemoLookup(nym_x32, nym_y32, sig, siglen, digest, dlen, status, issuer_hash32);
}
int fabgpu_csp_memo_has_block(fabgpu_csp* csp, uint64_t block_seq, uint64_t* entries) {
    if (!csp || !entries) return FABGPU_EINVAL;
    *entries = csp->csp->MemoHasBlock(block_seq);
    return FABGPU_OK;
}
int fabgpu_csp_memo_evict_block(fabgpu_csp* csp, uint64_t block_seq, uint64_t* evicted) {
    if (!csp) return FABGPU_EINVAL;
    size_t g = csp->csp->MemoEvictBlock(block_seq);
    if (evicted) *evicted = g;
    return FABGPU_OK;
}
int fabgpu_csp_memo_stats(fabgpu_csp* csp, uint64_t* entries, uint64_t* hits, uint64_t* misses, uint64_t* evicted) {
    if (!csp) return FABGPU_EINVAL;
    csp->csp->MemoStats(entries, hits, misses, evicted);
    return FABGPU_OK;
}
int fabgpu_csp_memo_set_capacity(fabgpu_csp* csp, uint64_t max_entries) {
    if (!csp) return FABGPU_EINVAL;
    csp->csp->MemoSetCapacity((size_t)max_entries);
    return FABGPU_OK;
}
int fabgpu_csp_identity_cache_limits(fabgpu_csp* csp, uint64_t max_identities, uint64_t max_registered_keys, uint32_t register_after_hits) {
    if (!csp) return FABGPU_EINVAL;
    csp->csp->SetIdentityCacheLimits((size_t)max_identities, (size_t)max_registered_keys, register_after_hits);
    return FABGPU_OK;
}
int fabgpu_csp_identity_cache_size(fabgpu_csp* csp, uint64_t* identities) {
    if (!csp || !identities) return FABGPU_EINVAL;
    *identities = csp->csp->IdentityCacheSize();
    return FABGPU_OK;
}

// pure host: structure of a marshalled block as the pre-verify pass sees it
int fabgpu_block_parse(const uint8_t* block, size_t len, uint32_t* n_tx, uint32_t* n_tuples, uint32_t* n_prefixes, uint8_t* tx_type, uint32_t cap_tx,
                       char* channel_id, size_t channel_cap) {
    if (!block || !n_tx || !n_tuples) return FABGPU_EINVAL;
    ParsedBlock pb;
    if (!ParseBlock(block, len, pb)) return FABGPU_EINVAL;
    *n_tx = pb.n_tx;
    *n_tuples = (uint32_t)pb.tuples.size();
    if (n_prefixes) *n_prefixes = (uint32_t)pb.prefixes.size();
    if (tx_type)
        for (uint32_t t = 0; t < pb.n_tx && t < cap_tx; t++) tx_type[t] = pb.tx_type[t];
    put_err(channel_id, channel_cap, pb.first_channel_id);
    return FABGPU_OK;
}

// pure host: P-256 public key of a PEM (is_pem != 0) or DER x509 certificate; 0 ok, 1 not a P-256 certificate
int fabgpu_x509_p256_pubkey(const uint8_t* cert, size_t len, int is_pem, uint8_t* qx32, uint8_t* qy32) {
    if (!cert || !qx32 || !qy32) return FABGPU_EINVAL;
    std::vector<uint8_t> der;
    if (is_pem) {
        if (!PemToDer(cert, len, der)) return 1;
        cert = der.data();
        len = der.size();
    }
    return CertDerToP256(cert, len, qx32, qy32) ? 0 : 1;
}

// pure host: the TxID / proposal-hash checks the pass derives from a marshalled block (for tests that recompute them with a CPU hash)
int fabgpu_block_hash_checks(const uint8_t* block, size_t len, uint32_t cap, uint32_t* n_checks, uint32_t* tx, uint8_t* kind, uint32_t* spans6,
                             uint32_t* expect2) {
    if (!block || !n_checks) return FABGPU_EINVAL;
    ParsedBlock pb;
    if (!ParseBlock(block, len, pb)) return FABGPU_EINVAL;
    *n_checks = (uint32_t)pb.hash_checks.size();
    if (pb.hash_checks.size() > cap) return FABGPU_ETOOBIG;
    for (size_t j = 0; j < pb.hash_checks.size(); j++) {
        const BlockHashCheck& hc = pb.hash_checks[j];
        if (tx) tx[j] = hc.tx;
        if (kind) kind[j] = hc.kind;
        if (spans6)
            for (int p = 0; p < 3; p++) {
                spans6[6 * j + 2 * p] = hc.piece[p].off;
                spans6[6 * j + 2 * p + 1] = hc.piece[p].off + hc.piece[p].len;
            }
        if (expect2) {
            expect2[2 * j] = hc.expect.off;
            expect2[2 * j + 1] = hc.expect.off + hc.expect.len;
        }
    }
    return FABGPU_OK;
}

// pure host: every (identity, message, signature) tuple the pass derives from a marshalled block, as spans (start, length) x 4 =
// identity, prefix, suffix, sig into the VIRTUAL arena  block || zero padding up to *tail_base || tail  (block_prepass.h: the
// orderer block-signature messages live in the tail).  message = prefix || suffix.
int fabgpu_block_tuples(const uint8_t* block, size_t len, uint32_t cap, uint32_t* n_tuples, uint32_t* tx, uint8_t* kind, uint32_t* spans8,
                        uint8_t* tail, uint32_t tail_cap, uint32_t* tail_len, uint32_t* tail_base) {
    if (!block || !n_tuples) return FABGPU_EINVAL;
    ParsedBlock pb;
    if (!ParseBlock(block, len, pb)) return FABGPU_EINVAL;
    *n_tuples = (uint32_t)pb.tuples.size();
    if (tail_len) *tail_len = (uint32_t)pb.tail.size();
    if (tail_base) *tail_base = pb.tail_base;
    if (pb.tuples.size() > cap || (tail && pb.tail.size() > tail_cap)) return FABGPU_ETOOBIG;
    for (size_t i = 0; i < pb.tuples.size(); i++) {
        const BlockTuple& t = pb.tuples[i];
        if (tx) tx[i] = t.tx;
        if (kind) kind[i] = t.kind;
        if (spans8) {
            const Span* sp[4] = {&t.identity, &t.prefix, &t.suffix, &t.sig};
            for (int k = 0; k < 4; k++) {
                spans8[8 * i + 2 * k] = sp[k]->off;
                spans8[8 * i + 2 * k + 1] = sp[k]->len;
            }
        }
    }
    if (tail && !pb.tail.empty()) memcpy(tail, pb.tail.data(), pb.tail.size());
    return FABGPU_OK;
}

// (pure host) what the host route makes of an identity: 0 = a PEM x509 certificate with an on-curve P-256 key (qxy set),
// 1 = anything else (identity.Verify needs bccsp/sw)
int fabgpu_identity_to_p256(const uint8_t* ident, size_t len, uint8_t* qxy64) {
    uint8_t qx[32], qy[32];
    if (!ident || !IdentityToP256(ident, len, qx, qy) || !PublicKeyOnCurve(qx, qy)) return 1;
    if (qxy64) {
        memcpy(qxy64, qx, 32);
        memcpy(qxy64 + 32, qy, 32);
    }
    return 0;
}
// ---- idemix (idemix_host.h) ----
int fabgpu_csp_idemix_msp_register(fabgpu_csp* csp, const char* mspid, const uint8_t* ipk_raw, size_t len, int64_t* issuer_id) {
    return fabgpu_csp_idemix_msp_register2(csp, "", mspid, ipk_raw, len, issuer_id);
}
int fabgpu_csp_idemix_msp_register2(fabgpu_csp* csp, const char* channel, const char* mspid, const uint8_t* ipk_raw, size_t len, int64_t* issuer_id) {
    if (!csp || !channel || !mspid || !ipk_raw || !issuer_id) return FABGPU_EINVAL;
    *issuer_id = csp->csp->RegisterIdemixMSP(mspid, ipk_raw, len, channel);
    return FABGPU_OK;
}

int fabgpu_idemix_issuer_key_is_canonical(const uint8_t* ipk_raw, size_t len) { return IdemixCSP::IssuerKeyEncodingIsCanonical(ipk_raw, len) ? 1 : 0; }
int fabgpu_csp_idemix_issuer_import(fabgpu_csp* csp, const uint8_t* ipk_raw, size_t len, int64_t* issuer_id, char* err, size_t errcap) {
    if (!csp || !issuer_id) return FABGPU_EINVAL;
    // (every device of the provider gets the issuer's tables, under one id)
    std::string msg;
    *issuer_id = csp->csp->ImportIdemixIssuer(ipk_raw, len, &msg);
    put_err(err, errcap, msg);
    return FABGPU_OK;
}

// n (nym key, signature, message) triples under ONE issuer key (what an idemix MSP verifies: msp/idemixmsp.go:584-599).
// nym keys and signatures are ragged byte strings exactly as the reference's KeyImport / Verify receive them.
// valid[i] 0/1; flags[i] bit0: bccsp/idemix must decide this tuple; errs[i] = Go error text or "".
int fabgpu_csp_idemix_nym_verify_batch(fabgpu_csp* csp, int64_t issuer_id, size_t n, const uint8_t* nym_arena, const uint32_t* nym_off,
                                       const uint8_t* sig_arena, const uint32_t* sig_off, const uint8_t* msg_arena, const uint32_t* msg_off,
                                       uint8_t* valid, uint8_t* flags, char* errs, size_t errstride) {
    if (!csp || (n && (!nym_off || !sig_off || !msg_off || !valid || !flags || !errs || !errstride))) return FABGPU_EINVAL;
    IdemixCSP ic(csp->csp->flat_ctx());
    IdemixIssuerPublicKey ipk;
    ipk.issuer_id = issuer_id;
    std::vector<NymPublicKey> keys(n);
    std::vector<uint8_t> key_ok(n);
    std::vector<NymVerifyItem> items(n);
    std::vector<std::string> import_err(n);
    for (size_t i = 0; i < n; i++) {
        Error e = ic.NymKeyImport(nym_arena + nym_off[i], nym_off[i + 1] - nym_off[i], keys[i]);
        key_ok[i] = e.ok();
        if (!e.ok()) import_err[i] = e.msg;
        items[i] = {e.ok() ? &keys[i] : nullptr, &ipk, sig_arena + sig_off[i], sig_off[i + 1] - sig_off[i], msg_arena + msg_off[i],
                    msg_off[i + 1] - msg_off[i]};
    }
    std::vector<VerifyResult> res;
    Error e = ic.NymVerifyBatch(items, res);
    if (!e.ok()) return FABGPU_ELAUNCH;
    for (size_t i = 0; i < n; i++) {
        valid[i] = res[i].valid ? 1 : 0;
        flags[i] = res[i].needs_sw ? 1 : 0;
        put_err(errs + i * errstride, errstride, key_ok[i] ? (res[i].err.ok() ? "" : res[i].err.msg) : import_err[i]);
    }
    return FABGPU_OK;
}
}  // extern "C"
