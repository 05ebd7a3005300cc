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

// ---- protobuf wire format ---------------------------------------------------------------------------------------
struct PbField {
    uint32_t num = 0, wt = 0;
    uint64_t varint = 0;
    const uint8_t* data = nullptr;   // wire type 2
    size_t len = 0;
};
struct PbReader {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    PbReader(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    bool varint(uint64_t& v) {
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
    bool next(PbField& f) {
        if (p >= end) return false;
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
    explicit Pick(uint32_t n) : num(n) {}
};
// false: malformed wire format, or one of the wanted fields repeated / not length-delimited.
// Hand-rolled scan (this is the walker's inner loop: ~25 messages per transaction): one-byte keys and one- or two-byte lengths - what
// every field of these messages has - take the fast path; anything else goes through the general varint decoder.
inline bool pb_pick(const uint8_t* b, size_t n, Pick* want, int k) {
    const uint8_t* p = b;
    const uint8_t* const end = b + n;
    while (p < end) {
        uint64_t key = *p++;
        if (key & 0x80) {                                              // multi-byte key: field numbers >= 16
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
        const uint32_t num = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (num == 0) return false;                                    // "illegal tag 0" in Go
        int hit = -1;
        for (int i = 0; i < k; i++)
            if (want[i].num == num) hit = i;
        if (wt == 2) {
            if (p >= end) return false;
            uint64_t len = *p++;
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
            if (len > (uint64_t)(end - p)) return false;
            if (hit >= 0) {
                if (want[hit].seen) return false;
                want[hit].seen = 1;
                want[hit].p = p;
                want[hit].len = (size_t)len;
            }
            p += len;
            continue;
        }
        if (hit >= 0) return false;                                    // a wanted field with another wire type: Go rejects the message
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

Span span_of(const uint8_t* base, const uint8_t* p, size_t n) {
    Span s;
    s.off = (uint32_t)(p - base);
    s.len = (uint32_t)n;
    return s;
}

// ---- DER ------------------------------------------------------------------------------------------------------------------
struct Der {
    const uint8_t* p;
    const uint8_t* end;
    // reads one TLV header; on success tag / content / len describe it and p is advanced past the whole element
    bool tlv(uint8_t& tag, const uint8_t*& content, size_t& len) {
        if (end - p < 2) return false;
        tag = *p++;
        size_t l = *p++;
        if (l & 0x80) {
            int nb = (int)(l & 0x7F);
            if (nb == 0 || nb > 4 || end - p < nb) return false;
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

const uint8_t OID_EC_PUBLIC_KEY[] = {0x2A, 0x86, 0x48, 0xCE, 0x3D, 0x02, 0x01};          // 1.2.840.10045.2.1
const uint8_t OID_PRIME256V1[] = {0x2A, 0x86, 0x48, 0xCE, 0x3D, 0x03, 0x01, 0x07};       // 1.2.840.10045.3.1.7

int b64val(uint8_t c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
}

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
        uint8_t c = pem[i];
        if (c == '-') break;
        if (c == '=' || c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;
        int v = b64val(c);
        if (v < 0) return false;
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
    Der top{der, der + len};
    uint8_t tag;
    const uint8_t* c;
    size_t l;
    if (!top.tlv(tag, c, l) || tag != 0x30) return false;             // Certificate
    Der cert{c, c + l};
    if (!cert.tlv(tag, c, l) || tag != 0x30) return false;            // TBSCertificate
    Der tbs{c, c + l};
    if (!tbs.tlv(tag, c, l)) return false;
    if (tag == 0xA0) {                                                // [0] version (absent in v1 certificates)
        if (!tbs.tlv(tag, c, l)) return false;
    }
    if (tag != 0x02) return false;                                    // serialNumber
    for (int k = 0; k < 4; k++)                                       // signature, issuer, validity, subject
        if (!tbs.tlv(tag, c, l) || tag != 0x30) return false;
    if (!tbs.tlv(tag, c, l) || tag != 0x30) return false;             // subjectPublicKeyInfo
    Der spki{c, c + l};
    if (!spki.tlv(tag, c, l) || tag != 0x30) return false;            // AlgorithmIdentifier
    Der alg{c, c + l};
    if (!alg.tlv(tag, c, l) || tag != 0x06 || l != sizeof(OID_EC_PUBLIC_KEY) || memcmp(c, OID_EC_PUBLIC_KEY, l) != 0) return false;
    if (!alg.tlv(tag, c, l) || tag != 0x06 || l != sizeof(OID_PRIME256V1) || memcmp(c, OID_PRIME256V1, l) != 0) return false;
    if (!spki.tlv(tag, c, l) || tag != 0x03) return false;            // BIT STRING: 00 04 X Y
    if (l != 66 || c[0] != 0x00 || c[1] != 0x04) return false;
    memcpy(qx, c + 2, 32);
    memcpy(qy, c + 34, 32);
    return true;
}

const uint8_t OID_ECDSA_WITH_SHA256[] = {0x2A, 0x86, 0x48, 0xCE, 0x3D, 0x04, 0x03, 0x02};   // 1.2.840.10045.4.3.2

// Certificate ::= SEQUENCE { tbsCertificate, signatureAlgorithm, signatureValue BIT STRING }: the raw TBS TLV (what is hashed,
// crypto/x509 Certificate.RawTBSCertificate), whether the OUTER algorithm is ecdsa-with-SHA256, and the DER signature inside the
// BIT STRING.  false: not a certificate this walker understands.
bool CertDerSignatureParts(const uint8_t* der, size_t len, Span& tbs, Span& sig, bool& ecdsa_sha256) {
    Der top{der, der + len};
    uint8_t tag;
    const uint8_t* c;
    size_t l;
    if (!top.tlv(tag, c, l) || tag != 0x30) return false;
    Der cert{c, c + l};
    const uint8_t* tbs_start = cert.p;
    if (!cert.tlv(tag, c, l) || tag != 0x30) return false;
    tbs.off = (uint32_t)(tbs_start - der);
    tbs.len = (uint32_t)(cert.p - tbs_start);
    if (!cert.tlv(tag, c, l) || tag != 0x30) return false;            // AlgorithmIdentifier
    Der alg{c, c + l};
    const uint8_t* oc;
    size_t ol;
    if (!alg.tlv(tag, oc, ol) || tag != 0x06) return false;
    ecdsa_sha256 = ol == sizeof(OID_ECDSA_WITH_SHA256) && memcmp(oc, OID_ECDSA_WITH_SHA256, ol) == 0;
    if (!cert.tlv(tag, c, l) || tag != 0x03 || l < 1 || c[0] != 0x00) return false;   // BIT STRING, no unused bits
    sig.off = (uint32_t)(c + 1 - der);
    sig.len = (uint32_t)(l - 1);
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
    const uint8_t *idb, *ms;
    size_t idl, msl;
    if (!pb_bytes(ident, len, 2, idb, idl) || !pb_bytes(ident, len, 1, ms, msl)) return false;
    PbReader r(idb, idl);
    PbField f;
    bool hx = false, hy = false, proof = false;
    while (r.next(f)) {
        if (f.wt != 2) continue;
        if (f.num == 1 && f.len == 32) { memcpy(nx, f.data, 32); hx = true; }
        if (f.num == 2 && f.len == 32) { memcpy(ny, f.data, 32); hy = true; }
        if (f.num == 5) proof = true;
    }
    if (!r.ok || !hx || !hy || !proof) return false;
    mspid.assign((const char*)ms, msl);
    return true;
}

namespace {
// one envelope -> its tuples (prefix indices local to `out`)
void parse_envelope(const uint8_t* block, const uint8_t* env, size_t env_len, uint32_t tx, ParsedBlock& out, uint8_t& tx_type, uint8_t& understood) {
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
                if (tx == 0) out.first_channel_id.assign((const char*)g.data, g.len);
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
    size_t first_tuple = out.tuples.size(), first_check = out.hash_checks.size(), first_prefix = out.prefixes.size();
    out.tuples.push_back(ct);
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
        out.hash_checks.push_back(hc);
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
        int32_t pidx = (int32_t)out.prefixes.size();
        out.prefixes.push_back(span_of(block, prp, prp_l));
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
            out.hash_checks.push_back(hc);
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
            out.tuples.push_back(et);
        }
        if (!ends.ok) good = false;
    }
    if (!acts.ok) good = false;
    if (!good) {
        out.tuples.resize(first_tuple);                            // leave the whole transaction to the Go validators
        out.hash_checks.resize(first_check);
        out.prefixes.resize(first_prefix);
        return;
    }
    understood = 1;
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
    static const int n = [] {
        const char* e = getenv("FABGPU_PASS_WALK_THREADS");
        // 8, not 16: the workers spin while the lister publishes chunks, and a peer's container is typically granted fewer CPUs than it
        // sees (the GPU boxes here: 256 visible, cgroup quota 16) - measured on such a box with the upload thread running beside
        // (tools/gpu_pass_probe.sh, 10 000-tx block): 16 workers 10.7 ms, 8 workers 1.9-2.2 ms, 4 workers 2.2-2.7 ms, 1 worker 5.9 ms
        int v = e ? atoi(e) : 8;
        return v < 1 ? 1 : (v > 16 ? 16 : v);
    }();
    return n;
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
    // block level: header fields and the orderer signatures over the block (TUPLE_BLOCK_SIG)
    out.data = span_of(block, data, dlen);
    out.tail_base = (uint32_t)((len + 63) / 64 * 64);
    std::vector<BlockTuple> block_sigs;
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
