// The envelope walker of the pre-verify pass as ONE body of code for the host and for the device (block_prepass.h describes what it
// extracts and which reference functions those tuples stand for).  block_prepass.cpp instantiates walk_envelope with an emitter that
// appends to vectors; block_walk_kernels.hip instantiates the same template once with a counting emitter and once with an emitter that
// writes at the offsets a prefix sum over those counts assigned - so the strictness rules (a repeated singular field, a wanted field with
// another wire type, a tag 0 ... make the transaction "not understood") hold on the device because they are the same lines, and the CPU
// tests of the host walker (ledger goldens, mutation fuzz under ASAN) cover the logic the kernels run.
//
// Also here, for the same reason: the signature gate of the common DER shape (gate_sig_fast) that the device applies per tuple; the
// general parser with Go's error texts (bccsp_host.cpp, bccsp/utils/ecdsa.go:43-67) stays on the host and decides whatever this one
// declines to.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define WALK_HD __host__ __device__
#define WALK_FORCEINLINE __forceinline__
#else
#define WALK_HD
#define WALK_FORCEINLINE inline __attribute__((always_inline))
#endif

namespace fab {
namespace bccsp {

struct Span {
    uint32_t off = 0, len = 0;   // into the block buffer
};

enum : uint8_t { TUPLE_CREATOR = 0, TUPLE_ENDORSEMENT = 1, TUPLE_BLOCK_SIG = 2 };
constexpr uint32_t BLOCK_LEVEL_TX = 0xFFFFFFFFu;
enum : uint8_t { HASH_TXID = 0, HASH_PROPOSAL = 1 };

struct BlockHashCheck {
    uint32_t tx = 0;
    uint8_t kind = HASH_TXID;
    Span piece[3];                  // the message is their concatenation (unused pieces have len 0)
    Span expect;                    // HASH_TXID: 64 hex characters; HASH_PROPOSAL: 32 raw bytes; anything else cannot match
};
struct BlockTuple {
    uint32_t tx = 0;
    uint8_t kind = TUPLE_CREATOR;
    Span identity, prefix, suffix, sig;   // prefix.len == 0 for creator tuples
    int32_t prefix_index = -1;
};
static_assert(sizeof(BlockTuple) == 44 && sizeof(BlockHashCheck) == 40 && sizeof(Span) == 8, "records travel between host and device as raw bytes");

namespace walk {

// ---- protobuf wire format ---------------------------------------------------------------------------------------
struct PbField {
    uint32_t num = 0, wt = 0;
    uint64_t varint = 0;
    const uint8_t* data = nullptr;   // wire type 2
    size_t len = 0;
};
// eight bytes at p, little-endian, any alignment (the caller has checked that they exist)
WALK_HD WALK_FORCEINLINE uint64_t load_le64(const uint8_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
    return *reinterpret_cast<const u64_unaligned*>(p);
#else
    uint64_t v;
    memcpy(&v, p, 8);
    return v;                                                          // (the hosts this is built for are little-endian)
#endif
}
struct PbReader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    WALK_HD PbReader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    WALK_HD bool varint(uint64_t& v) {
        v = 0;
        for (int shift = 0; shift < 64; shift += 7) {
            if (p >= end) return false;
            uint8_t c = *p++;
            v |= (uint64_t)(c & 0x7F) << shift;
            if (!(c & 0x80)) return true;
        }
        return false;
    }
    // next field; false at the end of the buffer or on malformed input (then ok == false)
    WALK_HD bool next(PbField& f) {
        if (p >= end) return false;
        if ((size_t)(end - p) >= 8) {                                  // the common shapes from ONE load (see pb_pick_n)
            const uint64_t w = load_le64(p);
            const uint32_t b0 = (uint32_t)w & 0xFFu, b1 = (uint32_t)(w >> 8) & 0xFFu, b2 = (uint32_t)(w >> 16) & 0xFFu;
            if (!(b0 & 0x80) && (b0 & 7) == 2 && (!(b1 & 0x80) || !(b2 & 0x80))) {
                const uint32_t hdr = (b1 & 0x80) ? 3u : 2u;
                const uint64_t n = (b1 & 0x80) ? ((uint64_t)(b1 & 0x7F) | ((uint64_t)b2 << 7)) : (uint64_t)b1;
                f.num = b0 >> 3;
                f.wt = 2;
                p += hdr;
                if (n > (uint64_t)(end - p)) return ok = false;
                f.data = p;
                f.len = (size_t)n;
                p += n;
                return true;
            }
            if (!(b0 & 0x80) && (b0 & 7) == 0 && !(b1 & 0x80)) {
                f.num = b0 >> 3;
                f.wt = 0;
                f.varint = b1;
                f.data = nullptr;
                f.len = 0;
                p += 2;
                return true;
            }
        }
        uint64_t key;
        if (!varint(key)) return ok = false;
        f.num = (uint32_t)(key >> 3);
        f.wt = (uint32_t)(key & 7);
        f.data = nullptr;
        f.len = 0;
        switch (f.wt) {
            case 0: return varint(f.varint) ? true : (ok = false);
            case 1: if (end - p < 8) return ok = false; p += 8; return true;
            case 5: if (end - p < 4) return ok = false; p += 4; return true;
            case 2: {
                uint64_t n;
                if (!varint(n) || n > (uint64_t)(end - p)) return ok = false;
                f.data = p;
                f.len = (size_t)n;
                p += n;
                return true;
            }
            default: return ok = false;
        }
    }
};

// Singular length-delimited fields.  golang/protobuf's proto.Unmarshal takes the LAST occurrence of a repeated singular bytes
// field and MERGES repeated embedded messages; no marshaller ever writes a singular field twice.  A walker that picked "an"
// occurrence could verify other bytes than the Go validators later see, so this one refuses the ambiguity instead of
// resolving it: every message is scanned to its end, and a wanted field that repeats (or arrives with another wire type, which
// Go rejects) makes the whole message "not understood" - the transaction then stays with the Go validators.
struct Pick {
    uint32_t num;
    const uint8_t* p = nullptr;
    size_t len = 0;
    int seen = 0;
    WALK_HD explicit Pick(uint32_t n) : num(n) {}
};
// false: malformed wire format, or one of the wanted fields repeated / not length-delimited.
// Hand-rolled scan (this is the walker's inner loop: ~25 messages per transaction): one-byte keys and one- or two-byte lengths - what
// every field of these messages has - take the fast path; anything else goes through the general varint decoder.
// K wanted fields with DISTINCT numbers.  Every access to want[] has a compile-time index (the loops over K are unrolled and the hit is
// applied per index, not through a computed one): on the device the picks then live in registers - with `want[hit]` they lived in
// scratch memory, and the count kernel spent its time waiting for its own spills (walk_count_kernel: 496 bytes of scratch per lane,
// 40 us for 100 envelopes).
template <int K>
WALK_HD WALK_FORCEINLINE bool pb_pick_n(const uint8_t* b, size_t n, Pick* want) {
    const uint8_t* p = b;
    const uint8_t* const end = b + n;
    while (p < end) {
        uint32_t num = 0, wt = 0;
        uint64_t len = 0;
        bool decoded = false;
        // The shape nearly every field of these messages has - a one-byte key, length-delimited, a length of one or two bytes - is
        // decoded from ONE eight-byte load: a lane walking an envelope waits for every load it depends on (~100 of them per envelope,
        // byte by byte; the wait, not the arithmetic, is what the count kernel's time is made of).
        if ((size_t)(end - p) >= 8) {
            const uint64_t w = load_le64(p);
            const uint32_t b0 = (uint32_t)w & 0xFFu, b1 = (uint32_t)(w >> 8) & 0xFFu, b2 = (uint32_t)(w >> 16) & 0xFFu;
            if (!(b0 & 0x80) && (b0 & 7) == 2) {
                if (!(b1 & 0x80)) {
                    num = b0 >> 3; wt = 2; len = b1; p += 2; decoded = true;
                } else if (!(b2 & 0x80)) {
                    num = b0 >> 3; wt = 2; len = (uint64_t)(b1 & 0x7F) | ((uint64_t)b2 << 7); p += 3; decoded = true;
                }
            }
        }
        if (!decoded) {
            uint64_t key = *p++;
            if (key & 0x80) {                                          // multi-byte key: field numbers >= 16
                key &= 0x7F;
                int shift = 7;
                for (;;) {
                    if (p >= end || shift > 63) return false;
                    const uint8_t c = *p++;
                    key |= (uint64_t)(c & 0x7F) << shift;
                    if (!(c & 0x80)) break;
                    shift += 7;
                }
            }
            num = (uint32_t)(key >> 3);
            wt = (uint32_t)(key & 7);
            if (wt == 2) {
                if (p >= end) return false;
                len = *p++;
                if (len & 0x80) {
                    len &= 0x7F;
                    int shift = 7;
                    for (;;) {
                        if (p >= end || shift > 63) return false;
                        const uint8_t c = *p++;
                        len |= (uint64_t)(c & 0x7F) << shift;
                        if (!(c & 0x80)) break;
                        shift += 7;
                    }
                }
            }
        }
        if (num == 0) return false;                                    // "illegal tag 0" in Go
        bool hit = false;
#pragma unroll
        for (int i = 0; i < K; i++) hit = hit || want[i].num == num;
        if (wt == 2) {
            if (len > (uint64_t)(end - p)) return false;
            bool again = false;
#pragma unroll
            for (int i = 0; i < K; i++) {
                if (want[i].num == num) {
                    again = again || want[i].seen != 0;
                    want[i].seen = 1;
                    want[i].p = p;
                    want[i].len = (size_t)len;
                }
            }
            if (again) return false;
            p += len;
            continue;
        }
        if (hit) return false;                                         // a wanted field with another wire type: Go rejects the message
        if (wt == 0) {
            int cnt = 0;
            for (;;) {
                if (p >= end || ++cnt > 10) return false;
                if (!(*p++ & 0x80)) break;
            }
        } else if (wt == 1) {
            if (end - p < 8) return false;
            p += 8;
        } else if (wt == 5) {
            if (end - p < 4) return false;
            p += 4;
        } else {
            return false;
        }
    }
    return true;
}
WALK_HD WALK_FORCEINLINE bool pb_pick(const uint8_t* b, size_t n, Pick* want, int k) {
    switch (k) {                                                       // (k is a literal at every call site: the switch folds away)
        case 1: return pb_pick_n<1>(b, n, want);
        case 2: return pb_pick_n<2>(b, n, want);
        case 3: return pb_pick_n<3>(b, n, want);
        default: return false;
    }
}

WALK_HD inline Span span_of(const uint8_t* base, const uint8_t* p, size_t n) {
    Span s;
    s.off = (uint32_t)(p - base);
    s.len = (uint32_t)n;
    return s;
}

// One envelope -> its tuples, shared prefixes and hash checks, through an emitter:
//     void    mark();                              the transaction starts here
//     void    rollback();                          forget everything since mark() (the transaction is left to the Go validators)
//     void    add_tuple(const BlockTuple&);
//     int32_t add_prefix(const Span&);             returns the index tuples of this action carry as prefix_index
//     void    add_check(const BlockHashCheck&);
//     void    channel_id(const uint8_t*, size_t);  ChannelHeader.channel_id of the envelope (fixture pin; the device ignores it)
template <class Em>
WALK_HD inline void walk_envelope(const uint8_t* block, const uint8_t* env, size_t env_len, uint32_t tx, Em& em, uint8_t& tx_type, uint8_t& understood) {
    tx_type = 255;
    understood = 0;
    // common.Envelope{1 payload, 2 signature}
    Pick e_[2] = {Pick(1), Pick(2)};
    if (!pb_pick(env, env_len, e_, 2) || !e_[0].seen) return;
    const uint8_t* payload = e_[0].p;
    const size_t payload_l = e_[0].len;
    const uint8_t* sig = e_[1].seen ? e_[1].p : payload;
    const size_t sig_l = e_[1].seen ? e_[1].len : 0;
    // common.Payload{1 header, 2 data}; common.Header{1 channel_header, 2 signature_header}
    Pick p_[2] = {Pick(1), Pick(2)};
    if (!pb_pick(payload, payload_l, p_, 2) || !p_[0].seen) return;
    Pick h_[2] = {Pick(1), Pick(2)};
    if (!pb_pick(p_[0].p, p_[0].len, h_, 2) || !h_[0].seen || !h_[1].seen) return;
    const uint8_t *chdr = h_[0].p, *shdr = h_[1].p;
    const size_t chdr_l = h_[0].len, shdr_l = h_[1].len;
    // common.ChannelHeader{1 type (varint), ..., 4 channel_id, 5 tx_id}
    uint8_t type = 0;   // proto3 default: MESSAGE
    Span txid_span;     // ChannelHeader.tx_id (field 5)
    {
        PbReader r(chdr, chdr_l);
        PbField g;
        int n_type = 0, n_chan = 0, n_txid = 0;
        while (r.next(g)) {
            if (g.num == 0) return;
            if (g.num == 1) {
                if (g.wt != 0 || n_type++) return;
                if (g.varint > 254) return;                            // no HeaderType is that large: leave it to Go
                type = (uint8_t)g.varint;
            }
            if (g.num == 4) {
                if (g.wt != 2 || n_chan++) return;
                em.channel_id(g.data, g.len);
            }
            if (g.num == 5) {
                if (g.wt != 2 || n_txid++) return;
                txid_span = span_of(block, g.data, g.len);
            }
        }
        if (!r.ok) return;
    }
    tx_type = type;
    // common.SignatureHeader{1 creator, 2 nonce}
    Pick s_[2] = {Pick(1), Pick(2)};
    if (!pb_pick(shdr, shdr_l, s_, 2) || !s_[0].seen) return;
    BlockTuple ct;
    ct.tx = tx;
    ct.kind = TUPLE_CREATOR;
    ct.identity = span_of(block, s_[0].p, s_[0].len);
    ct.suffix = span_of(block, payload, payload_l);
    ct.sig = span_of(block, sig, sig_l);
    em.mark();
    em.add_tuple(ct);
    if (type != 3) {                                               // only ENDORSER_TRANSACTION carries endorsements
        understood = 1;
        return;
    }
    {   // CheckTxID: endorser transactions only (msgvalidation.go:283-296)
        BlockHashCheck hc;
        hc.tx = tx;
        hc.kind = HASH_TXID;
        if (s_[1].seen) hc.piece[0] = span_of(block, s_[1].p, s_[1].len);
        hc.piece[1] = ct.identity;
        hc.expect = txid_span;
        em.add_check(hc);
    }
    bool good = p_[1].seen != 0;
    // peer.Transaction{1 repeated actions}; TransactionAction{1 header, 2 payload}
    PbReader acts(good ? p_[1].p : payload, good ? p_[1].len : 0);
    PbField a;
    while (good && acts.next(a)) {
        if (a.num == 0) { good = false; break; }
        if (a.num != 1) continue;
        if (a.wt != 2) { good = false; break; }
        // ChaincodeActionPayload{1 chaincode_proposal_payload, 2 action}; ChaincodeEndorsedAction{1 proposal_response_payload, 2 endorsements}
        Pick ta_[2] = {Pick(1), Pick(2)};
        if (!pb_pick(a.data, a.len, ta_, 2) || !ta_[1].seen) { good = false; break; }
        Pick cap_[2] = {Pick(1), Pick(2)};
        if (!pb_pick(ta_[1].p, ta_[1].len, cap_, 2) || !cap_[1].seen) { good = false; break; }
        const uint8_t* cea = cap_[1].p;
        const size_t cea_l = cap_[1].len;
        Pick prp_(1);
        if (!pb_pick(cea, cea_l, &prp_, 1) || !prp_.seen) { good = false; break; }
        const uint8_t* prp = prp_.p;
        const size_t prp_l = prp_.len;
        const int32_t pidx = em.add_prefix(span_of(block, prp, prp_l));
        {   // GetProposalHash2 of this action
            BlockHashCheck hc;
            hc.tx = tx;
            hc.kind = HASH_PROPOSAL;
            hc.piece[0] = span_of(block, chdr, chdr_l);
            if (ta_[0].seen) hc.piece[1] = span_of(block, ta_[0].p, ta_[0].len);
            if (cap_[0].seen) hc.piece[2] = span_of(block, cap_[0].p, cap_[0].len);
            Pick ph_(1);                                               // ProposalResponsePayload{1 proposal_hash, 2 extension}
            if (!pb_pick(prp, prp_l, &ph_, 1)) { good = false; break; }
            if (ph_.seen) hc.expect = span_of(block, ph_.p, ph_.len);
            em.add_check(hc);
        }
        PbReader ends(cea, cea_l);
        PbField e;
        while (ends.next(e)) {
            if (e.num != 2) continue;
            if (e.wt != 2) { good = false; break; }
            // peer.Endorsement{1 endorser, 2 signature}
            Pick en_[2] = {Pick(1), Pick(2)};
            if (!pb_pick(e.data, e.len, en_, 2) || !en_[0].seen) { good = false; break; }
            BlockTuple et;
            et.tx = tx;
            et.kind = TUPLE_ENDORSEMENT;
            et.identity = span_of(block, en_[0].p, en_[0].len);
            et.prefix = span_of(block, prp, prp_l);
            et.prefix_index = pidx;
            et.suffix = et.identity;                               // message = prp || endorser
            et.sig = en_[1].seen ? span_of(block, en_[1].p, en_[1].len) : span_of(block, en_[0].p, 0);
            em.add_tuple(et);
        }
        if (!ends.ok) good = false;
    }
    if (!acts.ok) good = false;
    if (!good) {
        em.rollback();                                             // leave the whole transaction to the Go validators
        return;
    }
    understood = 1;
}

// ---- the two emitters of the device walk (block_walk_kernels.hip; the host runs them in tests) ---------------------------------
// First run: count.  Then an exclusive prefix sum over the envelopes assigns every envelope its ranges.  Second run: write.
struct CountEmitter {
    uint32_t nt = 0, np = 0, nc = 0;
    uint64_t gb = 0;
    uint32_t m_nt = 0, m_np = 0, m_nc = 0;
    uint64_t m_gb = 0;
    WALK_HD void mark() { m_nt = nt; m_np = np; m_nc = nc; m_gb = gb; }
    WALK_HD void rollback() { nt = m_nt; np = m_np; nc = m_nc; gb = m_gb; }
    WALK_HD void add_tuple(const BlockTuple&) { nt++; }
    WALK_HD int32_t add_prefix(const Span&) { return (int32_t)np++; }
    WALK_HD void add_check(const BlockHashCheck& c) {
        nc++;
        gb += (uint64_t)c.piece[0].len + c.piece[1].len + c.piece[2].len;
    }
    WALK_HD void channel_id(const uint8_t*, size_t) {}
};

// What the counting run found, KEPT: an envelope's records in a fixed-size slot of its own (indices local to the envelope), so that
// the second run is a copy to the places the scan assigned instead of a second walk (the walk is ~35 us per envelope-lane; an
// endorser transaction is 4 tuples + 1 prefix + 2 hash checks).  An envelope with more records than the slot holds sets `over`: the
// second run walks it again, as before.
constexpr uint32_t STASH_TUPLES = 8, STASH_PREFIXES = 2, STASH_CHECKS = 4;
struct EnvStash {
    uint32_t over = 0, pad = 0;
    BlockTuple t[STASH_TUPLES];
    Span p[STASH_PREFIXES];
    BlockHashCheck c[STASH_CHECKS];
};
static_assert(sizeof(EnvStash) == 8 + 44 * STASH_TUPLES + 8 * STASH_PREFIXES + 40 * STASH_CHECKS, "no padding: one slot per envelope, raw bytes");
// CountEmitter that also fills a slot (same counts, same mark / rollback)
struct StashEmitter {
    EnvStash* slot;
    uint32_t nt = 0, np = 0, nc = 0;
    uint64_t gb = 0;
    uint32_t m_nt = 0, m_np = 0, m_nc = 0;
    uint64_t m_gb = 0;
    WALK_HD void mark() { m_nt = nt; m_np = np; m_nc = nc; m_gb = gb; }
    WALK_HD void rollback() { nt = m_nt; np = m_np; nc = m_nc; gb = m_gb; }
    WALK_HD void add_tuple(const BlockTuple& t) {
        if (nt < STASH_TUPLES) slot->t[nt] = t;
        nt++;
    }
    WALK_HD int32_t add_prefix(const Span& s) {
        if (np < STASH_PREFIXES) slot->p[np] = s;
        return (int32_t)np++;                                        // local: the copy adds the envelope's prefix base
    }
    WALK_HD void add_check(const BlockHashCheck& c) {
        if (nc < STASH_CHECKS) slot->c[nc] = c;
        nc++;
        gb += (uint64_t)c.piece[0].len + c.piece[1].len + c.piece[2].len;
    }
    WALK_HD void channel_id(const uint8_t*, size_t) {}
    WALK_HD bool fits() const { return nt <= STASH_TUPLES && np <= STASH_PREFIXES && nc <= STASH_CHECKS; }
};

// Writes at base + k while k is below what the counting run reserved for this envelope: a transaction that is rolled back reserved
// nothing, so its transient records never land in a neighbour's range.
struct WriteEmitter {
    BlockTuple* tuples;
    uint32_t* pre_off2;
    BlockHashCheck* checks;
    uint32_t* gather_spans;
    uint32_t* gather_off;
    uint32_t base_t, base_p, base_c, base_g;
    uint32_t lim_t, lim_p, lim_c;
    uint32_t* creator_spans;          // optional: (start, end) of the creator tuple's message at index creator_index (its hash can start early)
    uint32_t creator_index;
    uint32_t nt = 0, np = 0, nc = 0, g = 0;
    WALK_HD void mark() {}
    WALK_HD void rollback() { nt = np = nc = g = 0; }    // (mark() precedes the first record of an envelope)
    WALK_HD void add_tuple(const BlockTuple& t) {
        if (nt < lim_t) {
            tuples[base_t + nt] = t;
            if (nt == 0 && creator_spans) {                          // an envelope's first tuple is its creator's
                creator_spans[2 * (size_t)creator_index] = t.suffix.len ? t.suffix.off : 0;
                creator_spans[2 * (size_t)creator_index + 1] = t.suffix.len ? t.suffix.off + t.suffix.len : 0;
            }
        }
        nt++;
    }
    WALK_HD int32_t add_prefix(const Span& s) {
        if (np < lim_p) {
            pre_off2[2 * (size_t)(base_p + np)] = s.len ? s.off : 0;
            pre_off2[2 * (size_t)(base_p + np) + 1] = s.len ? s.off + s.len : 0;
        }
        return (int32_t)(base_p + np++);
    }
    WALK_HD void add_check(const BlockHashCheck& c) {
        if (nc < lim_c) {
            const size_t j = base_c + nc;
            checks[j] = c;
            gather_off[j] = base_g + g;
            for (int p = 0; p < 3; p++) {
                gather_spans[6 * j + 2 * p] = c.piece[p].len ? c.piece[p].off : 0;
                gather_spans[6 * j + 2 * p + 1] = c.piece[p].len ? c.piece[p].off + c.piece[p].len : 0;
            }
        }
        g += c.piece[0].len + c.piece[1].len + c.piece[2].len;
        nc++;
    }
    WALK_HD void channel_id(const uint8_t*, size_t) {}
};

// ---- the signature gate of the common DER shape ----------------------------------------------------------------------------
// bccsp/sw/ecdsa.go:41-57 before any arithmetic: UnmarshalECDSASignature (bccsp/utils/ecdsa.go:43-67: asn1.Unmarshal into
// {R, S *big.Int}, R and S > 0) and IsLowS (bccsp/utils/ecdsa.go:84-92).  This decides the one shape every signer produces -
// a minimal DER SEQUENCE { INTEGER r, INTEGER s }, short-form lengths, nothing behind it, r and s positive and below 2^256 - exactly
// as the general parser would: GATE_SUBMIT (s <= n/2: r32 / s32 are set, the device decides, r >= n included) or GATE_HIGH_S.
// Everything else (long-form lengths, trailing bytes, negative / zero / oversize integers, truncation ...) is GATE_DECLINED: the
// caller takes the general parser and its error texts.
enum : uint8_t { GATE_SUBMIT = 0, GATE_HIGH_S = 1, GATE_EMPTY = 2, GATE_DECLINED = 3 };
WALK_HD inline bool der_minimal_positive(const uint8_t* p, uint32_t l) {
    if (p[0] & 0x80) return false;                                  // negative
    if (p[0] == 0) return l > 1 && (p[1] & 0x80) != 0 && l <= 33;   // a leading zero must be needed (also excludes zero)
    return l <= 32;
}
WALK_HD inline uint8_t gate_sig_fast(const uint8_t* sig, uint32_t siglen, uint8_t* r32, uint8_t* s32) {
    // n/2 of P-256 (bccsp/utils/ecdsa.go:30-37 curveHalfOrders)
    const uint8_t HALF_N[32] = {0x7f, 0xff, 0xff, 0xff, 0x80, 0x00, 0x00, 0x00, 0x7f, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
                                0xde, 0x73, 0x7d, 0x56, 0xd3, 0x8b, 0xcf, 0x42, 0x79, 0xdc, 0xe5, 0x61, 0x7e, 0x31, 0x92, 0xa8};
    if (siglen == 0) return GATE_EMPTY;
    if (siglen < 8 || siglen > 72 || sig[0] != 0x30 || sig[1] != siglen - 2 || sig[2] != 0x02) return GATE_DECLINED;
    uint32_t lr = sig[3];
    if (lr < 1 || lr > 33 || 4 + lr + 2 > siglen || sig[4 + lr] != 0x02) return GATE_DECLINED;
    uint32_t ls = sig[5 + lr];
    const uint8_t* pr = sig + 4;
    const uint8_t* ps = sig + 6 + lr;
    if (ls < 1 || ls > 33 || 6 + lr + ls != siglen || !der_minimal_positive(pr, lr) || !der_minimal_positive(ps, ls)) return GATE_DECLINED;
    if (pr[0] == 0) { pr++; lr--; }
    if (ps[0] == 0) { ps++; ls--; }
    for (uint32_t k = 0; k < 32; k++) {
        r32[k] = k + lr >= 32 ? pr[k + lr - 32] : 0;
        s32[k] = k + ls >= 32 ? ps[k + ls - 32] : 0;
    }
    for (int k = 0; k < 32; k++) {
        if (s32[k] < HALF_N[k]) return GATE_SUBMIT;
        if (s32[k] > HALF_N[k]) return GATE_HIGH_S;
    }
    return GATE_SUBMIT;                                             // s == n/2 is low
}

// ---- the signature gate in general ---------------------------------------------------------------------------------------------
// Everything gate_sig_fast declines, decided the way the reference decides it - so that no signature, however it is encoded, takes a
// block off the device route (one crafted signature used to send a 10 000-transaction block to the host walk).  Restates, outcome
// for outcome, bccsp/sw/ecdsa.go:41-57 up to the arithmetic:
//   utils.UnmarshalECDSASignature (bccsp/utils/ecdsa.go:43-67): Go's asn1.Unmarshal into struct{R, S *big.Int} - identifier octet
//     0x30 (a high-tag-number form never matches), DER length (short form, or long form that is minimal, without leading zero, below
//     2^23 at every step, and present), content inside the input; two INTEGERs (identifier 0x02, same length rules, not empty,
//     minimally encoded); bytes behind the second INTEGER inside the SEQUENCE and bytes behind the SEQUENCE are ignored (asn1's struct
//     parser / the discarded `rest`); then R > 0 and S > 0                                              -> GATE_BAD_DER otherwise
//   utils.IsLowS (bccsp/utils/ecdsa.go:84-92): S <= n/2                                                  -> GATE_HIGH_S otherwise
//   ecdsa.Verify's range check: an R of more than 256 bits is >= n, (false, nil)                         -> GATE_RANGE
//   else GATE_SUBMIT: R and S are the magnitudes at sig[pr, pr + lr) and sig[ps, ps + ls), lr, ls <= 32 (R < n is the device's check).
// The host's general parser with Go's error TEXTS (bccsp_host.cpp UnmarshalECDSASignature) stays what single-signature callers get;
// tests hold the two against each other on every shape (tests/test_device_walk.py).
enum : uint8_t { GATE_BAD_DER = 4, GATE_RANGE = 5 };
struct DerTL {
    uint32_t len;
    bool ok;
};
// Go asn1.go parseTagAndLength + parseField's identifier comparison for one expected identifier octet; off moves past the header
WALK_HD inline DerTL gate_parse_tl(const uint8_t* b, uint32_t n, uint32_t& off, uint8_t want) {
    DerTL r{0, false};
    if (off >= n) return r;
    const uint8_t id = b[off++];
    if ((id & 0x1F) == 0x1F) return r;
    if (off >= n) return r;
    const uint8_t l0 = b[off++];
    uint32_t L = l0;
    if (l0 & 0x80) {
        const uint32_t nb = l0 & 0x7F;
        if (nb == 0) return r;                                       // indefinite length
        L = 0;
        for (uint32_t i = 0; i < nb; i++) {
            if (off >= n) return r;
            if (L >= (1u << 23)) return r;                           // "length too large"
            L = (L << 8) | b[off++];
            if (L == 0) return r;                                    // "superfluous leading zeros in length"
        }
        if (L < 0x80) return r;                                      // "non-minimal length"
    }
    if (id != want) return r;
    if (L > n - off) return r;
    r.len = L;
    r.ok = true;
    return r;
}
// one INTEGER of the SEQUENCE content b[0, n): sign (-1, 0, 1) and where its magnitude lies (positive values only)
WALK_HD inline bool gate_parse_int(const uint8_t* b, uint32_t n, uint32_t& off, int& sign, uint32_t& mag_off, uint32_t& mag_len) {
    if (off == n) return false;                                      // "sequence truncated"
    const DerTL tl = gate_parse_tl(b, n, off, 0x02);
    if (!tl.ok) return false;
    const uint32_t p = off, L = tl.len;
    off += L;
    if (L == 0) return false;                                        // "empty integer"
    if (L > 1 && ((b[p] == 0x00 && !(b[p + 1] & 0x80)) || (b[p] == 0xFF && (b[p + 1] & 0x80)))) return false;   // "not minimally-encoded"
    if (b[p] & 0x80) {
        sign = -1;
        return true;
    }
    uint32_t z = 0;                                                  // a minimal non-negative INTEGER has at most one leading zero octet
    while (z < L && b[p + z] == 0) z++;
    sign = z == L ? 0 : 1;
    mag_off = p + z;
    mag_len = L - z;
    return true;
}
WALK_HD inline uint8_t gate_sig_general(const uint8_t* sig, uint32_t siglen, uint32_t& pr, uint32_t& lr, uint32_t& ps, uint32_t& ls) {
    const uint8_t HALF_N[32] = {0x7f, 0xff, 0xff, 0xff, 0x80, 0x00, 0x00, 0x00, 0x7f, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
                                0xde, 0x73, 0x7d, 0x56, 0xd3, 0x8b, 0xcf, 0x42, 0x79, 0xdc, 0xe5, 0x61, 0x7e, 0x31, 0x92, 0xa8};
    pr = lr = ps = ls = 0;
    if (siglen == 0) return GATE_EMPTY;
    uint32_t off = 0;
    const DerTL seq = gate_parse_tl(sig, siglen, off, 0x30);
    if (!seq.ok) return GATE_BAD_DER;
    const uint8_t* in = sig + off;
    uint32_t io = 0, ro = 0, rl = 0, so = 0, sl = 0;
    int rsign = 0, ssign = 0;
    if (!gate_parse_int(in, seq.len, io, rsign, ro, rl)) return GATE_BAD_DER;
    if (!gate_parse_int(in, seq.len, io, ssign, so, sl)) return GATE_BAD_DER;
    if (rsign != 1 || ssign != 1) return GATE_BAD_DER;               // "R / S must be larger than zero"
    if (sl > 32) return GATE_HIGH_S;
    for (uint32_t k = 0; k < 32; k++) {
        const uint8_t v = k + sl >= 32 ? in[so + k + sl - 32] : 0;
        if (v < HALF_N[k]) break;
        if (v > HALF_N[k]) return GATE_HIGH_S;
    }
    if (rl > 32) return GATE_RANGE;
    pr = off + ro; lr = rl; ps = off + so; ls = sl;
    return GATE_SUBMIT;
}
// both gates as one answer (never GATE_DECLINED): what the device route's gate kernel computes per tuple, in lane form
WALK_HD inline uint8_t gate_sig_any(const uint8_t* sig, uint32_t siglen, uint8_t* r32, uint8_t* s32) {
    uint8_t g = gate_sig_fast(sig, siglen, r32, s32);
    if (g != GATE_DECLINED) return g;
    uint32_t pr, lr, ps, ls;
    g = gate_sig_general(sig, siglen, pr, lr, ps, ls);
    if (g == GATE_SUBMIT)
        for (uint32_t k = 0; k < 32; k++) {
            r32[k] = k + lr >= 32 ? sig[pr + k + lr - 32] : 0;
            s32[k] = k + ls >= 32 ? sig[ps + k + ls - 32] : 0;
        }
    return g;
}

// ---- x509 certificate (DER) -> where its P-256 public key lies ---------------------------------------------------------------
// Just enough DER to reach SubjectPublicKeyInfo (what msp/mspimpl.go:408-421 takes from x509.ParseCertificate): Certificate ->
// TBSCertificate -> [0] version (optional), serialNumber, signature, issuer, validity, subject, subjectPublicKeyInfo{algorithm{
// id-ecPublicKey, prime256v1}, BIT STRING 00 04 X Y}.  Returns the offset of X (Y follows) or -1.  One body of code for the host's
// identity decoder (block_prepass.cpp CertDerToP256) and the device's (block_walk_kernels.hip, over the bytes a wavefront decoded
// from the PEM into LDS).
struct DerCursor {
    const uint8_t* p;
    const uint8_t* end;
    // A WINDOW on a longer encoding (the device's certificate decoder keeps the first 3 KiB of a certificate in LDS, whatever its
    // length): bytes from `limit` on are not there.  Lengths are still checked against `end` - the real extent - and a header that
    // would have to be READ beyond the window sets *beyond and fails.  limit == nullptr: everything up to `end` is there.
    const uint8_t* limit = nullptr;
    bool* beyond = nullptr;
    WALK_HD bool missing(const uint8_t* upto) const {      // true: bytes below `upto` lie outside the window
        if (limit && upto > limit) {
            if (beyond) *beyond = true;
            return true;
        }
        return false;
    }
    // reads one TLV header; on success tag / content / len describe it and p is advanced past the whole element
    WALK_HD bool tlv(uint8_t& tag, const uint8_t*& content, size_t& len) {
        if (end - p < 2) return false;
        if (missing(p + 2)) return false;
        tag = *p++;
        size_t l = *p++;
        if (l & 0x80) {
            const int nb = (int)(l & 0x7F);
            if (nb == 0 || nb > 4 || end - p < nb) return false;
            if (missing(p + nb)) return false;
            l = 0;
            for (int i = 0; i < nb; i++) l = (l << 8) | *p++;
        }
        if ((size_t)(end - p) < l) return false;
        content = p;
        len = l;
        p += l;
        return true;
    }
};
WALK_HD inline bool der_bytes_equal(const uint8_t* a, const uint8_t* b, size_t n) {
    bool same = true;
    for (size_t i = 0; i < n; i++) same = same && a[i] == b[i];
    return same;
}
// `avail` <= len: how many of the certificate's `len` bytes are at `der` (a window, see DerCursor).  Returns the offset of X, -1
// (not a certificate with a P-256 key), or -2: the answer depends on bytes beyond the window.
WALK_HD inline int32_t cert_der_p256_key_offset_window(const uint8_t* der, size_t avail, size_t len) {
    const uint8_t OID_EC_PUBLIC_KEY[7] = {0x2A, 0x86, 0x48, 0xCE, 0x3D, 0x02, 0x01};          // 1.2.840.10045.2.1
    const uint8_t OID_PRIME256V1[8] = {0x2A, 0x86, 0x48, 0xCE, 0x3D, 0x03, 0x01, 0x07};       // 1.2.840.10045.3.1.7
    bool beyond = false;
    const uint8_t* const lim = avail < len ? der + avail : nullptr;
#define FAB_DER_FAIL return beyond ? -2 : -1
    DerCursor top{der, der + len, lim, &beyond};
    uint8_t tag;
    const uint8_t* c;
    size_t l;
    if (!top.tlv(tag, c, l) || tag != 0x30) FAB_DER_FAIL;          // Certificate
    DerCursor cert{c, c + l, lim, &beyond};
    if (!cert.tlv(tag, c, l) || tag != 0x30) FAB_DER_FAIL;         // TBSCertificate
    DerCursor tbs{c, c + l, lim, &beyond};
    if (!tbs.tlv(tag, c, l)) FAB_DER_FAIL;
    if (tag == 0xA0) {                                             // [0] version (absent in v1 certificates)
        if (!tbs.tlv(tag, c, l)) FAB_DER_FAIL;
    }
    if (tag != 0x02) FAB_DER_FAIL;                                 // serialNumber
    for (int k = 0; k < 4; k++)                                    // signature, issuer, validity, subject
        if (!tbs.tlv(tag, c, l) || tag != 0x30) FAB_DER_FAIL;
    if (!tbs.tlv(tag, c, l) || tag != 0x30) FAB_DER_FAIL;          // subjectPublicKeyInfo
    DerCursor spki{c, c + l, lim, &beyond};
    if (!spki.tlv(tag, c, l) || tag != 0x30) FAB_DER_FAIL;         // AlgorithmIdentifier
    DerCursor alg{c, c + l, lim, &beyond};
    if (!alg.tlv(tag, c, l) || tag != 0x06 || l != 7) FAB_DER_FAIL;
    if (alg.missing(c + 7)) return -2;
    if (!der_bytes_equal(c, OID_EC_PUBLIC_KEY, 7)) return -1;
    if (!alg.tlv(tag, c, l) || tag != 0x06 || l != 8) FAB_DER_FAIL;
    if (alg.missing(c + 8)) return -2;
    if (!der_bytes_equal(c, OID_PRIME256V1, 8)) return -1;
    if (!spki.tlv(tag, c, l) || tag != 0x03) FAB_DER_FAIL;         // BIT STRING: 00 04 X Y
    if (l != 66) return -1;
    if (spki.missing(c + 66)) return -2;
    if (c[0] != 0x00 || c[1] != 0x04) return -1;
    return (int32_t)(c + 2 - der);
#undef FAB_DER_FAIL
}
WALK_HD inline int32_t cert_der_p256_key_offset(const uint8_t* der, size_t len) { return cert_der_p256_key_offset_window(der, len, len); }
// PEM text -> the class of one character, as PemToDer (block_prepass.cpp) reads the body of a certificate block: a base64 digit (its
// value), something it skips ('=', line ends, blanks), the dash that ends the body, or a character that makes the block invalid
enum : int { PEM_SKIP = 64, PEM_DASH = 65, PEM_INVALID = 66 };
WALK_HD inline int pem_char_class(uint8_t c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    if (c == '-') return PEM_DASH;
    if (c == '=' || c == '\n' || c == '\r' || c == ' ' || c == '\t') return PEM_SKIP;
    return PEM_INVALID;
}

// ---- idemix creators ---------------------------------------------------------------------------------------------------------------
// msp.SerializedIdentity{1 mspid, 2 id_bytes = msp.SerializedIdemixIdentity{1 nym_x, 2 nym_y, 3 ou, 4 role, 5 proof}} (what
// idemixidentity.Serialize writes, msp/idemixmsp.go:605-640) -> where the MSP id and the two 32-byte pseudonym coordinates lie.
// false: not such an identity (or coordinates of another size: those stay with bccsp/idemix).  One body of code for the host's
// IdentityToIdemixNym (block_prepass.cpp) and the gate kernel.
struct IdemixNymRef {
    const uint8_t* mspid = nullptr;
    uint32_t mspid_len = 0;
    const uint8_t* nx = nullptr;
    const uint8_t* ny = nullptr;
};
WALK_HD inline bool identity_to_idemix_nym(const uint8_t* ident, size_t len, IdemixNymRef& out) {
    Pick idb(2), ms(1);
    if (!pb_pick(ident, len, &idb, 1) || idb.seen != 1) return false;
    if (!pb_pick(ident, len, &ms, 1) || ms.seen != 1) return false;
    PbReader r(idb.p, idb.len);
    PbField f;
    bool proof = false;
    out.nx = out.ny = nullptr;
    while (r.next(f)) {
        if (f.wt != 2) continue;
        if (f.num == 1 && f.len == 32) out.nx = f.data;
        if (f.num == 2 && f.len == 32) out.ny = f.data;
        if (f.num == 5) proof = true;
    }
    if (!r.ok || !out.nx || !out.ny || !proof) return false;
    out.mspid = ms.p;
    out.mspid_len = (uint32_t)ms.len;
    return true;
}
// idemix.NymSignature{1 proof_c, 2 proof_s_sk, 3 proof_s_r_nym, 4 nonce} (last occurrence wins, as proto.Unmarshal): true when the bytes
// are a protobuf message AND all four fields are 32 bytes (what the nym kernels take; anything else stays with bccsp/idemix)
WALK_HD inline bool unmarshal_nym_signature32(const uint8_t* raw, size_t len, const uint8_t* (&field)[4]) {
    size_t flen[4] = {0, 0, 0, 0};
    field[0] = field[1] = field[2] = field[3] = nullptr;
    PbReader r(raw, len);
    PbField f;
    while (r.next(f)) {
        if (f.num == 0) return false;                                  // "illegal tag 0"
        if (f.wt == 2 && f.num >= 1 && f.num <= 4) {
            field[f.num - 1] = f.data;
            flen[f.num - 1] = f.len;
        }
    }
    return r.ok && flen[0] == 32 && flen[1] == 32 && flen[2] == 32 && flen[3] == 32;
}

// ---- identity bytes -> 64-bit table hash ---------------------------------------------------------------------------------
// The device looks identities up in a table of the ones the provider has met (block_walk_kernels.hip); the hash only picks the slot,
// equality is always decided on ALL the bytes.  It covers the length and two rows of 64 bytes that a wavefront loads with one
// instruction each - the LAST 64 bytes (for a certificate the end of its signature) and 64 bytes SPREAD over the whole string (byte
// l * len / 64 for lane l: the subject, the public key ...) - folded per lane, mixed with per-lane odd constants and summed.  A
// per-provider random seed enters every lane, and a lookup gives up after WALK_ID_PROBE_MAX probes (the table builder never places an
// entry further from its home slot): identities crafted to collide cost a bounded number of comparisons and are then simply treated
// as unknown - the device decodes their certificates itself, which is always correct.  (Round 2's hash covered the last 64 bytes
// only: ten thousand certificates that shared them - re-keyed copies of one certificate - made every lookup walk 4 096 entries.)
constexpr uint32_t WALK_ID_PROBE_MAX = 16;
WALK_HD inline uint64_t id_stream_const(uint32_t l) { return (0x9E3779B97F4A7C15ull * (uint64_t)(2 * l + 1)) | 1ull; }
WALK_HD inline uint64_t id_stream_fold(uint64_t h, uint8_t b) { return (h ^ b) * 0x100000001B3ull; }
WALK_HD inline uint64_t id_stream_basis(uint64_t seed) { return 0xCBF29CE484222325ull ^ seed; }
WALK_HD inline uint32_t id_spread_pos(uint32_t l, uint32_t len) { return (uint32_t)(((uint64_t)l * len) >> 6); }   // < len for len > 0
WALK_HD inline uint64_t id_hash_finish(uint64_t sum, uint32_t len) {
    uint64_t h = sum ^ ((uint64_t)len * 0xD6E8FEB86659FD93ull);
    h ^= h >> 32;
    h *= 0xD6E8FEB86659FD93ull;
    h ^= h >> 29;
    return h;
}
// one lane's term of the sum (lane l of 64)
WALK_HD inline uint64_t id_lane_term(const uint8_t* p, uint32_t len, uint32_t l, uint64_t seed) {
    const uint32_t m = len < 64 ? len : 64;
    uint64_t h = id_stream_basis(seed);
    if (l < m) h = id_stream_fold(h, p[len - m + l]);
    if (len) h = id_stream_fold(h, p[id_spread_pos(l, len)]);
    // (xor-shifts around the per-lane multiplication: without them the sum is LINEAR in the small differences two bytes make - the
    // per-lane constants are multiples of one number - and strings that differ in a few sampled bytes collide in droves)
    h ^= h >> 31;
    h *= id_stream_const(l);
    h ^= h >> 29;
    return h;
}
inline uint64_t id_hash_host(const uint8_t* p, uint32_t len, uint64_t seed = 0) {
    uint64_t sum = 0;
    for (uint32_t l = 0; l < 64; l++) sum += id_lane_term(p, len, l, seed);
    return id_hash_finish(sum, len);
}

// ---- the block's DIGEST memo (bccsp_host.h BlockMemo "digest memo"; bccsp.Hash of bytes a pass has already hashed) ----------------
// Slot choice only - a hit is decided by comparing EVERY byte of the caller's message with the block's: the length and eight 8-byte
// samples spread evenly from the first to the last byte of the message a || b (b may be empty).  Endorsement messages of one
// transaction share their first kilobyte (prp) and their last bytes (the PEM trailer of the endorser's certificate), which is why
// first-and-last-bytes alone would not do; samples in between land in the certificates' bodies.  The same lines run on the device
// (walk_memo_index_kernel, over the two spans of a tuple) and on the host (GPUCSP::HashLookup, over the caller's contiguous bytes).
// (in two pieces so that the device can gather a message's 64 sampled bytes with one load per lane of a wavefront - walk_memo_write_kernel -
//  and still run THESE lines over them: where sample k starts in a message of n >= 8 bytes, and the mixing of the eight samples)
WALK_HD WALK_FORCEINLINE uint64_t msg_fingerprint_sample_pos(uint64_t n, uint32_t k) { return (n - 8) * k / 7; }
WALK_HD WALK_FORCEINLINE uint64_t msg_fingerprint_mix(uint64_t n, const uint64_t (&w)[8]) {
    uint64_t h = (n + 1) * 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < 8; k++) {
        h = (h ^ w[k]) * 0xD6E8FEB86659FD93ull;
        h ^= h >> 29;
    }
    return h ^ (h >> 32);
}
WALK_HD inline uint64_t msg_fingerprint(const uint8_t* a, uint32_t alen, const uint8_t* b, uint32_t blen) {
    const uint64_t n = (uint64_t)alen + blen;
    auto at = [&](uint64_t pos) -> uint64_t { return pos < alen ? a[pos] : b[pos - alen]; };
    if (n < 8) {
        uint64_t h = (n + 1) * 0x9E3779B97F4A7C15ull;
        for (uint64_t j = 0; j < n; j++) h = (h ^ at(j)) * 0x100000001B3ull;
        return h ^ (h >> 32);
    }
    uint64_t w[8];
    for (uint32_t k = 0; k < 8; k++) {
        const uint64_t p = msg_fingerprint_sample_pos(n, k);
        if (p + 8 <= alen) {
            w[k] = load_le64(a + p);                                       // a sample inside one span: ONE (unaligned) eight-byte load ...
        } else if (p >= alen) {
            w[k] = load_le64(b + (p - alen));
        } else {                                                           // ... the one that straddles the two spans: byte by byte
            w[k] = 0;
            for (uint64_t j = 0; j < 8; j++) w[k] |= at(p + j) << (8 * j);
        }
    }
    return msg_fingerprint_mix(n, w);
}
// messages shorter than this are not worth a lookup (one SHA-256 block costs less than the call); the C ABI answers "miss"
constexpr uint32_t HASH_MEMO_MIN_LEN = 64;
// entries with one fingerprint a lookup compares before it gives up (a sender can craft equal fingerprints; it cannot make a lookup expensive)
constexpr uint32_t HASH_MEMO_MAX_PROBES = 16;

}  // namespace walk
}  // namespace bccsp
}  // namespace fab
