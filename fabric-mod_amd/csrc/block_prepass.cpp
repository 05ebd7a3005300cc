// Block walker + identity decoding of the pre-verify pass (block_prepass.h).  Host only, no device code, no crypto:
// protobuf wire format, PEM/base64, and just enough DER to reach SubjectPublicKeyInfo.
#include "block_prepass.h"
#include "worker_pool.h"

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>

namespace fab {
namespace bccsp {

namespace {

// ---- protobuf wire format: block_walk_core.h (shared with the device walker) ----------------------------------------------------
using walk::PbField;
using walk::PbReader;
using walk::Pick;
using walk::pb_pick;
using walk::span_of;
// the one wanted field: 1 present once, 0 absent, -1 ambiguous / malformed
int pb_one(const uint8_t* b, size_t n, uint32_t num, const uint8_t*& out, size_t& outlen) {
    Pick w(num);
    if (!pb_pick(b, n, &w, 1)) return -1;
    out = w.p;
    outlen = w.len;
    return w.seen;
}
// compatibility form for callers that only ask "is there exactly one": absent and ambiguous both answer false
bool pb_bytes(const uint8_t* b, size_t n, uint32_t num, const uint8_t*& out, size_t& outlen) { return pb_one(b, n, num, out, outlen) == 1; }

// ---- DER: block_walk_core.h (DerCursor, cert_der_p256_key_offset, pem_char_class - shared with the device's identity decoder) ----
typedef walk::DerCursor Der;

}  // namespace

bool PemToDer(const uint8_t* pem, size_t len, std::vector<uint8_t>& der) {
    static const char BEGIN[] = "-----BEGIN CERTIFICATE-----";
    static const char END[] = "-----END CERTIFICATE-----";
    const size_t bl = sizeof(BEGIN) - 1, el = sizeof(END) - 1;
    size_t i = 0;
    while (i + bl <= len && memcmp(pem + i, BEGIN, bl) != 0) i++;
    if (i + bl > len) return false;
    i += bl;
    der.clear();
    uint32_t acc = 0;
    int bits = 0;
    for (; i < len; i++) {
        const int v = walk::pem_char_class(pem[i]);
        if (v == walk::PEM_DASH) break;
        if (v == walk::PEM_SKIP) continue;
        if (v == walk::PEM_INVALID) return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            der.push_back((uint8_t)(acc >> bits));
        }
    }
    if (i + el > len || memcmp(pem + i, END, el) != 0) return false;
    return !der.empty();
}

bool CertDerToP256(const uint8_t* der, size_t len, uint8_t qx[32], uint8_t qy[32]) {
    const int32_t at = walk::cert_der_p256_key_offset(der, len);
    if (at < 0) return false;
    memcpy(qx, der + at, 32);
    memcpy(qy, der + at + 32, 32);
    return true;
}

bool IdentityToP256(const uint8_t* ident, size_t len, uint8_t qx[32], uint8_t qy[32]) {
    const uint8_t* idb;
    size_t idl;
    if (!pb_bytes(ident, len, 2, idb, idl)) return false;             // msp.SerializedIdentity{1 mspid, 2 id_bytes}
    std::vector<uint8_t> der;
    if (!PemToDer(idb, idl, der)) return false;
    return CertDerToP256(der.data(), der.size(), qx, qy);
}

namespace {
void der_len(std::vector<uint8_t>& o, size_t n) {
    if (n < 128) { o.push_back((uint8_t)n); return; }
    uint8_t tmp[8];
    int k = 0;
    while (n) { tmp[k++] = (uint8_t)(n & 0xFF); n >>= 8; }
    o.push_back((uint8_t)(0x80 | k));
    while (k) o.push_back(tmp[--k]);
}
void der_octets(std::vector<uint8_t>& o, const uint8_t* p, size_t n) {
    o.push_back(0x04);
    der_len(o, n);
    o.insert(o.end(), p, p + n);
}
}  // namespace

void BlockHeaderBytes(uint64_t number, const uint8_t* prev, size_t prev_len, const uint8_t* data_hash, size_t dh_len, std::vector<uint8_t>& out) {
    std::vector<uint8_t> body;
    // INTEGER: minimal big-endian two's complement of a non-negative value (a leading 0x00 when the top bit is set)
    uint8_t be[9];
    int k = 0;
    be[k++] = 0;
    for (int i = 7; i >= 0; i--) be[k++] = (uint8_t)(number >> (8 * i));
    int first = 0;
    while (first < 8 && be[first] == 0 && !(be[first + 1] & 0x80)) first++;
    body.push_back(0x02);
    der_len(body, (size_t)(9 - first));
    body.insert(body.end(), be + first, be + 9);
    der_octets(body, prev, prev_len);
    der_octets(body, data_hash, dh_len);
    out.clear();
    out.push_back(0x30);
    der_len(out, body.size());
    out.insert(out.end(), body.begin(), body.end());
}

bool HashCheckMatches(const uint8_t* block, const BlockHashCheck& hc, const uint8_t digest[32]) {
    const uint8_t* e = block + hc.expect.off;
    if (hc.kind == HASH_PROPOSAL) return hc.expect.len == 32 && memcmp(e, digest, 32) == 0;
    if (hc.expect.len != 64) return false;
    static const char HEX[] = "0123456789abcdef";                 // hex.EncodeToString: lowercase
    for (int i = 0; i < 32; i++)
        if (e[2 * i] != (uint8_t)HEX[digest[i] >> 4] || e[2 * i + 1] != (uint8_t)HEX[digest[i] & 15]) return false;
    return true;
}

bool IdentityToIdemixNym(const uint8_t* ident, size_t len, std::string& mspid, uint8_t nx[32], uint8_t ny[32]) {
    walk::IdemixNymRef ref;
    if (!walk::identity_to_idemix_nym(ident, len, ref)) return false;       // (block_walk_core.h: shared with the gate kernel)
    memcpy(nx, ref.nx, 32);
    memcpy(ny, ref.ny, 32);
    mspid.assign((const char*)ref.mspid, ref.mspid_len);
    return true;
}

namespace {
// the host's emitter of walk::walk_envelope: appends to the vectors of a ParsedBlock (prefix indices local to it)
struct VectorEmitter {
    ParsedBlock& out;
    uint32_t tx;
    size_t first_tuple = 0, first_check = 0, first_prefix = 0;
    void mark() {
        first_tuple = out.tuples.size();
        first_check = out.hash_checks.size();
        first_prefix = out.prefixes.size();
    }
    void rollback() {
        out.tuples.resize(first_tuple);
        out.hash_checks.resize(first_check);
        out.prefixes.resize(first_prefix);
    }
    void add_tuple(const BlockTuple& t) { out.tuples.push_back(t); }
    int32_t add_prefix(const Span& s) {
        out.prefixes.push_back(s);
        return (int32_t)out.prefixes.size() - 1;
    }
    void add_check(const BlockHashCheck& c) { out.hash_checks.push_back(c); }
    void channel_id(const uint8_t* p, size_t n) {
        if (tx == 0) out.first_channel_id.assign((const char*)p, n);
    }
};
// one envelope -> its tuples (prefix indices local to `out`)
void parse_envelope(const uint8_t* block, const uint8_t* env, size_t env_len, uint32_t tx, ParsedBlock& out, uint8_t& tx_type, uint8_t& understood) {
    VectorEmitter em{out, tx};
    walk::walk_envelope(block, env, env_len, tx, em, tx_type, understood);
}
}  // namespace

// The walk has one inherently serial part - finding where each envelope starts (envelope i + 1 begins where envelope i ends: a
// chain of dependent cache misses through the BlockData, ~1 ms for 10 000 envelopes of a 50 MB block) - and one parallel part
// (the envelopes themselves).  They overlap: the calling thread lists envelopes in chunks of WALK_CHUNK and publishes each
// chunk as soon as it is complete; worker threads claim chunks in order and parse them into per-chunk parts, which are
// merged in order at the end (prefix indices rebased).
namespace {
constexpr uint32_t WALK_CHUNK = 256;
struct EnvChunk {
    std::pair<const uint8_t*, size_t> env[WALK_CHUNK];
    uint32_t count = 0;
};
}  // namespace

int WalkThreads() {
    // 8, not 16: the workers spin while the lister publishes chunks, and a peer's container is typically granted fewer CPUs than it
    // sees (the GPU boxes here: 256 visible, cgroup quota 16) - measured on such a box with the upload thread running beside
    // (round 2, 10 000-tx block): 16 workers 10.7 ms, 8 workers 1.9-2.2 ms, 4 workers 2.2-2.7 ms, 1 worker 5.9 ms.  A constant since
    // round 4 (it was FABGPU_PASS_WALK_THREADS while it was being measured).
    return 8;
}

namespace {
// block level: header fields and the orderer signatures over the block (TUPLE_BLOCK_SIG); `top` = the picked fields of common.Block
void parse_block_level(const uint8_t* block, size_t len, const Pick* top, ParsedBlock& out, std::vector<BlockTuple>& block_sigs) {
    const uint8_t* data = top[1].p;
    const size_t dlen = top[1].len;
    out.data = span_of(block, data, dlen);
    out.tail_base = (uint32_t)((len + 63) / 64 * 64);
    block_sigs.clear();
    if (top[0].seen) {
        // common.BlockHeader{1 number (varint), 2 previous_hash, 3 data_hash}
        PbReader r(top[0].p, top[0].len);
        PbField f;
        int c1 = 0, c2 = 0, c3 = 0;
        bool ok = true;
        while (r.next(f)) {
            if (f.num == 0) ok = false;
            if (f.num == 1) { if (f.wt != 0 || c1++) ok = false; else out.number = f.varint; }
            if (f.num == 2) { if (f.wt != 2 || c2++) ok = false; else out.previous_hash = span_of(block, f.data, f.len); }
            if (f.num == 3) { if (f.wt != 2 || c3++) ok = false; else out.data_hash = span_of(block, f.data, f.len); }
        }
        out.has_header = ok && r.ok;
    }
    if (out.has_header && top[2].seen && len <= 0xFFFF0000ull) {       // (the tail must be addressable behind the block with 32-bit offsets)
        // common.BlockMetadata{1 repeated bytes metadata}; entry [BlockMetadataIndex_SIGNATURES = 0] is a marshalled
        // common.Metadata{1 value, 2 repeated MetadataSignature{1 signature_header, 2 signature}}
        PbReader r(top[2].p, top[2].len);
        PbField f;
        const uint8_t* m0 = nullptr;
        size_t m0_l = 0;
        bool have = false, ok = true;
        while (r.next(f)) {
            if (f.num != 1) continue;
            if (f.wt != 2) { ok = false; break; }
            if (!have) { m0 = f.data; m0_l = f.len; have = true; }
        }
        ok = ok && r.ok && have;
        Pick val(1);
        if (ok && !pb_pick(m0, m0_l, &val, 1)) ok = false;
        if (ok) {
            std::vector<uint8_t> hb;
            BlockHeaderBytes(out.number, block + out.previous_hash.off, out.previous_hash.len, block + out.data_hash.off, out.data_hash.len, hb);
            PbReader sr(m0, m0_l);
            PbField g;
            while (sr.next(g)) {
                if (g.num != 2) continue;
                if (g.wt != 2) { ok = false; break; }
                Pick ms[2] = {Pick(1), Pick(2)};
                if (!pb_pick(g.data, g.len, ms, 2)) { ok = false; break; }
                // protoutil.UnmarshalSignatureHeader(signature_header).Creator; an absent header unmarshals to an empty creator
                Pick cr(1);
                if (ms[0].seen && !pb_pick(ms[0].p, ms[0].len, &cr, 1)) { ok = false; break; }
                BlockTuple bt;
                bt.tx = BLOCK_LEVEL_TX;
                bt.kind = TUPLE_BLOCK_SIG;
                bt.identity = cr.seen ? span_of(block, cr.p, cr.len) : span_of(block, block, 0);
                bt.sig = ms[1].seen ? span_of(block, ms[1].p, ms[1].len) : span_of(block, block, 0);
                const size_t at = out.tail.size();
                if (val.seen) out.tail.insert(out.tail.end(), val.p, val.p + val.len);
                if (ms[0].seen) out.tail.insert(out.tail.end(), ms[0].p, ms[0].p + ms[0].len);
                out.tail.insert(out.tail.end(), hb.begin(), hb.end());
                if ((uint64_t)out.tail_base + out.tail.size() > 0xFFFFFFF0ull) { ok = false; break; }
                bt.suffix.off = out.tail_base + (uint32_t)at;
                bt.suffix.len = (uint32_t)(out.tail.size() - at);
                block_sigs.push_back(bt);
            }
            if (!sr.ok) ok = false;
        }
        if (!ok) {
            block_sigs.clear();
            out.tail.clear();
        }
        out.block_sigs_understood = ok;
    }
    out.n_block_sigs = (uint32_t)block_sigs.size();
}
}  // namespace

// The host's share of a DEVICE-side walk (block_walk_kernels.hip): the outer framing, where each envelope starts (a serial chain by
// nature: envelope i + 1 begins where envelope i ends), the header fields and the orderers' block signatures with their tail.
bool OutlineBlock(const uint8_t* block, size_t len, ParsedBlock& out, std::vector<uint32_t>& env_spans, std::vector<BlockTuple>& block_sigs,
                  std::vector<uint32_t>* payload_spans) {
    out.reset();
    env_spans.clear();
    block_sigs.clear();
    if (payload_spans) payload_spans->clear();
    if (len > 0xFFFFFFF0ull) return false;
    Pick top[3] = {Pick(1), Pick(2), Pick(3)};                       // common.Block{1 header, 2 data, 3 metadata}
    if (!pb_pick(block, len, top, 3) || !top[1].seen) return false;
    PbReader r(top[1].p, top[1].len);
    PbField f;
    while (r.next(f)) {
        if (f.num != 1 || f.wt != 2) continue;                        // common.BlockData{1 repeated bytes data}
        // This loop is a chain of dependent cache misses (where envelope i + 1 starts is written at the start of envelope i), and a
        // block that has just arrived is in nobody's cache: ~100 ns per envelope, 1 ms for 10 000.  Transactions of one block tend to
        // be of similar size, so the lines where the NEXT few envelopes would start if they were as long as this one are requested
        // now (a wrong guess costs nothing but the request): 0.75-1.3 ms -> the chain runs out of the cache.
        {
            const uint8_t* guess = f.data + f.len;
            const size_t step = f.len + 3;
            for (int k = 0; k < 6; k++, guess += step)
                if (guess + 128 < block + len) {
                    __builtin_prefetch(guess);
                    __builtin_prefetch(guess + 64);
                }
        }
        env_spans.push_back((uint32_t)(f.data - block));
        env_spans.push_back((uint32_t)f.len);
        if (payload_spans) {
            // common.Envelope{1 payload, 2 signature}, by the walker's own rule (walk_envelope's first step): what a creator signed
            Pick e_[2] = {Pick(1), Pick(2)};
            const bool ok = pb_pick(f.data, f.len, e_, 2) && e_[0].seen && e_[0].len != 0;
            payload_spans->push_back(ok ? (uint32_t)(e_[0].p - block) : 0u);
            payload_spans->push_back(ok ? (uint32_t)(e_[0].p - block + e_[0].len) : 0u);
        }
    }
    if (!r.ok) {
        env_spans.clear();
        if (payload_spans) payload_spans->clear();
        return false;
    }
    out.n_tx = (uint32_t)(env_spans.size() / 2);
    parse_block_level(block, len, top, out, block_sigs);
    return true;
}

bool ParseBlock(const uint8_t* block, size_t len, ParsedBlock& out, int max_threads) {
    out.reset();
    if (len > 0xFFFFFFF0ull) return false;
    Pick top[3] = {Pick(1), Pick(2), Pick(3)};                       // common.Block{1 header, 2 data, 3 metadata}
    if (!pb_pick(block, len, top, 3) || !top[1].seen) return false;    // (a repeated embedded message would MERGE in Go: refused)
    const uint8_t* data = top[1].p;
    const size_t dlen = top[1].len;
    int nt = max_threads > 16 ? 16 : max_threads;
    if (dlen < ((size_t)1 << 20) || nt < 2) nt = 0;                   // small blocks: everything on the calling thread
    // chunk table: an envelope is at least 2 bytes, so dlen / (2 WALK_CHUNK) + 1 chunks is an upper bound nobody reaches;
    // one pointer per possible chunk is cheap next to the block itself
    const size_t max_chunks = dlen / (2 * (size_t)WALK_CHUNK) + 2;
    std::vector<std::unique_ptr<EnvChunk>> chunks(max_chunks);
    std::atomic<uint32_t> ready(0), next(0);
    std::atomic<bool> listing_done(false), broken(false);
    std::vector<std::unique_ptr<ParsedBlock>>& part = out.parts;      // one per chunk, storage reused from block to block
    std::mutex grow_mu;                                               // the pointer table grows under this lock; the parts never move
    auto chunk_part = [&](uint32_t ci) {
        std::lock_guard<std::mutex> lk(grow_mu);
        if (part.size() <= ci) part.resize((size_t)ci + 16);
        if (!part[ci]) part[ci].reset(new ParsedBlock);
        return part[ci].get();
    };
    auto parse_chunk = [&](uint32_t ci) {
        ParsedBlock* p = chunk_part(ci);
        const EnvChunk& ch = *chunks[ci];
        p->reset();
        p->tuples.reserve((size_t)ch.count * 5);
        p->hash_checks.reserve((size_t)ch.count * 2);
        p->prefixes.reserve(ch.count);
        p->tx_type.assign(ch.count, 255);                             // chunk-local: indexed by position inside the chunk
        p->tx_understood.assign(ch.count, 0);
        for (uint32_t k = 0; k < ch.count; k++)
            parse_envelope(block, ch.env[k].first, ch.env[k].second, ci * WALK_CHUNK + k, *p, p->tx_type[k], p->tx_understood[k]);
    };
    auto worker = [&] {
        for (;;) {
            uint32_t ci = next.fetch_add(1, std::memory_order_relaxed);
            while (ready.load(std::memory_order_acquire) <= ci) {
                if (listing_done.load(std::memory_order_acquire) && ready.load(std::memory_order_acquire) <= ci) return;
                std::this_thread::yield();
            }
            parse_chunk(ci);
        }
    };
    // listing (this thread) while the pool's workers parse the chunks it publishes
    uint32_t n = 0, nchunks = 0;
    auto lister = [&] {
        PbReader r(data, dlen);
        PbField f;
        std::unique_ptr<EnvChunk> cur(new EnvChunk);
        while (r.next(f)) {
            if (f.num != 1 || f.wt != 2) continue;                    // common.BlockData{1 repeated bytes data}
            cur->env[cur->count++] = std::make_pair(f.data, f.len);
            n++;
            if (cur->count == WALK_CHUNK) {
                chunks[nchunks] = std::move(cur);
                nchunks++;
                ready.store(nchunks, std::memory_order_release);
                cur.reset(new EnvChunk);
            }
        }
        if (!r.ok) broken.store(true);
        if (cur->count) {
            chunks[nchunks] = std::move(cur);
            nchunks++;
            ready.store(nchunks, std::memory_order_release);
        }
        listing_done.store(true, std::memory_order_release);
    };
    if (nt == 0) {
        lister();
        for (uint32_t ci = 0; ci < nchunks; ci++) parse_chunk(ci);
    } else {
        run_workers(nt + 1, [&](int w) {
            if (w == 0) lister();                                     // ... and then helps with what is left
            worker();
        });
    }
    if (broken.load()) {
        out.reset();
        return false;
    }
    std::vector<BlockTuple> block_sigs;
    parse_block_level(block, len, top, out, block_sigs);
    // merge in chunk order
    out.n_tx = n;
    out.tx_type.resize(n);
    out.tx_understood.resize(n);
    size_t ntup = 0, npre = 0, nchk = 0;
    for (uint32_t ci = 0; ci < nchunks; ci++) { ntup += part[ci]->tuples.size(); npre += part[ci]->prefixes.size(); nchk += part[ci]->hash_checks.size(); }
    out.tuples.reserve(ntup + block_sigs.size());
    out.prefixes.reserve(npre);
    out.hash_checks.reserve(nchk);
    for (uint32_t ci = 0; ci < nchunks; ci++) {
        ParsedBlock& p = *part[ci];
        if (ci == 0) out.first_channel_id = p.first_channel_id;
        const uint32_t cnt = chunks[ci]->count;
        memcpy(out.tx_type.data() + (size_t)ci * WALK_CHUNK, p.tx_type.data(), cnt);
        memcpy(out.tx_understood.data() + (size_t)ci * WALK_CHUNK, p.tx_understood.data(), cnt);
        int32_t base = (int32_t)out.prefixes.size();
        out.prefixes.insert(out.prefixes.end(), p.prefixes.begin(), p.prefixes.end());
        size_t first = out.tuples.size();
        out.tuples.insert(out.tuples.end(), p.tuples.begin(), p.tuples.end());
        if (base)
            for (size_t i = first; i < out.tuples.size(); i++)
                if (out.tuples[i].prefix_index >= 0) out.tuples[i].prefix_index += base;
        out.hash_checks.insert(out.hash_checks.end(), p.hash_checks.begin(), p.hash_checks.end());
    }
    out.tuples.insert(out.tuples.end(), block_sigs.begin(), block_sigs.end());   // block-level tuples come last
    return true;
}

}  // namespace bccsp
}  // namespace fab
