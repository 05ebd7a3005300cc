// Host mirror of the idemix nym-signature verbs (idemix_host.h).  Pure plumbing: protobuf field extraction, the argument
// checks of the reference's handlers, SoA marshalling, status -> Go error text.
#include "idemix_host.h"

#include <string.h>

#include "bn29_consts.h"
#include "fp256.h"

namespace fab {
namespace bccsp {

namespace {

struct Field {
    uint32_t num = 0, wt = 0;
    const uint8_t* data = nullptr;
    size_t len = 0;
};
// minimal protobuf wire walker (same rules as block_prepass.cpp's)
struct Walker {
    const uint8_t *p, *end;
    bool ok = true;
    Walker(const uint8_t* b, size_t n) : p(b), end(b + n) {}
    bool varint(uint64_t& v) {
        v = 0;
        for (int s = 0; s < 64; s += 7) {
            if (p >= end) return false;
            uint8_t c = *p++;
            v |= (uint64_t)(c & 0x7F) << s;
            if (!(c & 0x80)) return true;
        }
        return false;
    }
    bool next(Field& f) {
        if (p >= end) return false;
        uint64_t key, v;
        if (!varint(key)) return ok = false;
        f.num = (uint32_t)(key >> 3);
        f.wt = (uint32_t)(key & 7);
        f.data = nullptr;
        f.len = 0;
        if (f.num == 0) return ok = false;
        switch (f.wt) {
            case 0: return varint(v) ? true : (ok = false);
            case 1: if (end - p < 8) return ok = false; p += 8; return true;
            case 5: if (end - p < 4) return ok = false; p += 4; return true;
            case 2:
                if (!varint(v) || v > (uint64_t)(end - p)) return ok = false;
                f.data = p;
                f.len = (size_t)v;
                p += v;
                return true;
            default: return ok = false;
        }
    }
};

// idemix.ECP{x = 1, y = 2} with both coordinates exactly 32 bytes
bool ecp32(const uint8_t* b, size_t n, uint8_t* x, uint8_t* y) {
    Walker w(b, n);
    Field f;
    bool hx = false, hy = false;
    while (w.next(f)) {
        if (f.wt != 2) continue;
        if (f.num == 1 && f.len == 32) { memcpy(x, f.data, 32); hx = true; }
        if (f.num == 2 && f.len == 32) { memcpy(y, f.data, 32); hy = true; }
    }
    return w.ok && hx && hy;
}

}  // namespace

bool UnmarshalNymSignature(const uint8_t* raw, size_t len, NymSignatureFields& out) {
    Walker w(raw, len);
    Field f;
    while (w.next(f)) {
        if (f.wt == 2 && f.num >= 1 && f.num <= 4) {      // last occurrence wins, as proto.Unmarshal
            out.f[f.num - 1] = f.data;
            out.len[f.num - 1] = f.len;
        }
    }
    return w.ok;
}

// 0: the bytes do not parse; 1: parsed, but a field the device needs is missing or has another size; 2: HSk, HRand, Hash in place
static int issuer_key_fields(const uint8_t* raw, size_t len, IdemixIssuerPublicKey& out) {
    Walker w(raw, len);
    Field f;
    bool hsk = false, hrand = false, hash = false;
    while (w.next(f)) {
        if (f.wt != 2) continue;
        if (f.num == 2) hsk = ecp32(f.data, f.len, out.hsk_x, out.hsk_y);
        if (f.num == 3) hrand = ecp32(f.data, f.len, out.hrand_x, out.hrand_y);
        if (f.num == 10 && f.len == 32) { memcpy(out.hash, f.data, 32); hash = true; }
    }
    if (!w.ok) return 0;
    return hsk && hrand && hash ? 2 : 1;
}
// Is `raw` exactly what golang/protobuf would produce when it marshals the IssuerPublicKey it unmarshalled from `raw` (idemix.proto:18-48)?
// SetHash (idemix/issuerkey.go:171-182) hashes the RE-MARSHALLED key with Hash cleared; the library hashes the bytes it was given with
// field 10 cut out.  The two agree exactly when the encoding is canonical: known fields only, all length-delimited, in ascending field
// order (repeated only where the message repeats: 1 attribute_names, 4 h_attrs), single-byte tags, minimal length varints, no singular
// bytes field present-but-empty (proto3 drops those when marshalling), and the same for the nested ECP / ECP2 messages.  Anything else
// (out-of-order, duplicate or unknown fields, padded varints) is left to bccsp/idemix.
static bool canonical_message(const uint8_t* b, size_t n, int max_field, uint32_t repeated_mask, uint32_t message_mask, uint32_t ecp2_mask, int depth) {
    const uint8_t *p = b, *end = b + n;
    int prev = 0;
    while (p < end) {
        const uint8_t tag = *p++;
        if (tag & 0x80) return false;                                   // fields 1..15 have one-byte tags
        const int num = tag >> 3;
        if ((tag & 7) != 2 || num < 1 || num > max_field) return false;
        if (num < prev || (num == prev && !((repeated_mask >> num) & 1))) return false;
        prev = num;
        uint64_t v = 0;
        int k = 0;
        for (;; k++) {
            if (p >= end || k >= 5) return false;
            const uint8_t c = *p++;
            v |= (uint64_t)(c & 0x7F) << (7 * k);
            if (!(c & 0x80)) {
                if (k > 0 && c == 0) return false;                      // a padded (non-minimal) length
                break;
            }
        }
        if (v > (uint64_t)(end - p)) return false;
        const bool is_msg = (message_mask >> num) & 1;
        if (is_msg) {
            if (depth >= 1) return false;
            const bool ecp2 = (ecp2_mask >> num) & 1;
            if (!canonical_message(p, (size_t)v, ecp2 ? 4 : 2, 0, 0, 0, depth + 1)) return false;
        } else if (v == 0 && !((repeated_mask >> num) & 1)) {
            return false;                                               // a singular bytes field that is empty is not marshalled at all
        }
        p += v;
    }
    return true;
}
static bool canonical_issuer_key(const uint8_t* raw, size_t len) {
    // repeated: 1, 4; messages: 2 3 4 6 7 (ECP) and 5 (ECP2)
    return canonical_message(raw, len, 10, (1u << 1) | (1u << 4), (1u << 2) | (1u << 3) | (1u << 4) | (1u << 5) | (1u << 6) | (1u << 7), 1u << 5, 0);
}
bool IdemixCSP::IssuerKeyEncodingIsCanonical(const uint8_t* raw, size_t len) { return raw && len && canonical_issuer_key(raw, len); }

bool IdemixCSP::IssuerKeyFields(const uint8_t* raw, size_t len, IdemixIssuerPublicKey& out) {
    out.issuer_id = -1;
    return raw && len && issuer_key_fields(raw, len, out) == 2;
}

Error IdemixCSP::IssuerKeyImport(const uint8_t* raw, size_t len, IdemixIssuerPublicKey& out) const {
    if (!raw || len == 0) return Error("invalid raw, it must not be nil");                      // handlers/issuer.go KeyImport
    const int got = issuer_key_fields(raw, len, out);
    if (got == 0) return Error("failed to unmarshal issuer public key");
    out.issuer_id = -1;
    if (got != 2) return Error();   // a key the device cannot take (odd field sizes): valid for bccsp/idemix, not accelerated
    if (!canonical_issuer_key(raw, len)) return Error();   // Go would hash other bytes than these (see canonical_message): not accelerated
    // The reference never trusts field 10: IssuerPublicKey.Check ends in SetHash (idemix/issuerkey.go:171-182) - Hash = HashModOrder of the
    // key marshalled with Hash cleared - and the Go side's lookups carry THAT value.  The challenge of every pseudonym signature and
    // the memo's issuer binding hang on it, so the same is recomputed here (the marshalled key minus its field 10; SHA-256 on the device,
    // once per registration) and a key whose field 10 says something else is not accelerated.
    {
        std::vector<uint8_t> cleared;
        cleared.reserve(len);
        Walker w(raw, len);
        Field f;
        const uint8_t* at = raw;
        while (w.next(f)) {
            if (f.num != 10) cleared.insert(cleared.end(), at, w.p);
            at = w.p;
        }
        const uint32_t off[2] = {0u, (uint32_t)cleared.size()};
        uint8_t dig[32];
        if (cleared.empty() || fabgpu_sha256_batch(ctx_, 1, cleared.data(), off, dig) != FABGPU_OK) return Error();
        // HashModOrder (idemix/util.go:46-51): the digest as a big-endian number, mod the group order (r > 2^255: one subtraction)
        const u256 R = FAB_BN_R;
        u256 x, t, red;
        from_be32(x, dig);
        const uint32_t borrow = sub256(t, x, R);
        sel256(red, borrow == 0, t, x);
        uint8_t want[32];
        to_be32(want, red);
        if (memcmp(want, out.hash, 32) != 0) return Error();
    }
    uint32_t id = 0;
    int rc = fabgpu_idemix_issuer_register(ctx_, out.hsk_x, out.hsk_y, out.hrand_x, out.hrand_y, out.hash, &id);
    if (rc == FABGPU_OK) out.issuer_id = id;
    return Error();                                // registration failure (off-curve base, table limit): not accelerated, not an error
}

Error IdemixCSP::NymKeyImport(const uint8_t* raw, size_t len, NymPublicKey& out) const {
    if (!raw || len == 0) return Error("invalid raw, it must not be nil");                      // handlers/nym.go:157-159
    out.halves_are_32 = (len == 64);
    if (out.halves_are_32) {
        memcpy(out.x, raw, 32);
        memcpy(out.y, raw + 32, 32);
    }
    return Error();
}

Error IdemixCSP::NymVerifyBatch(const std::vector<NymVerifyItem>& items, std::vector<VerifyResult>& results) const {
    const size_t n = items.size();
    results.assign(n, VerifyResult());
    std::vector<uint32_t> idx;                      // items that go to the device
    std::vector<uint8_t> cols[6];
    std::vector<uint32_t> issuer, off(1, 0);
    std::vector<uint8_t> arena;
    for (size_t i = 0; i < n; i++) {
        const NymVerifyItem& it = items[i];
        VerifyResult& r = results[i];
        // the checks of NymVerifier.Verify, in its order (handlers/nymsigner.go:63-84)
        if (!it.key) { r.err = Error("invalid key, expected *nymPublicKey"); continue; }
        if (!it.ipk) { r.err = Error("invalid options, missing issuer public key"); continue; }
        if (!it.sig || it.siglen == 0) { r.err = Error("invalid signature, it must not be empty"); continue; }
        NymSignatureFields sf;
        if (!UnmarshalNymSignature(it.sig, it.siglen, sf)) {
            // Bytes this walker does not accept.  golang/protobuf's Unmarshal (bridge/nymsignaturescheme.go:83-86) is more
            // permissive in places (it skips unknown groups, for one), and a verdict is consensus-relevant: no error text is
            // invented here - bccsp/idemix decides, and words its own error.
            r.needs_sw = true;
            continue;
        }
        // FP256BN.FromBytes reads exactly 32 bytes: other sizes panic or truncate inside amcl -> bccsp/sw decides
        bool sizes = sf.len[0] == 32 && sf.len[1] == 32 && sf.len[2] == 32 && sf.len[3] == 32;
        if (!sizes || !it.key->halves_are_32 || it.ipk->issuer_id < 0 || it.dlen > 0x7FFFFFFFu - arena.size()) {
            r.needs_sw = true;
            continue;
        }
        idx.push_back((uint32_t)i);
        cols[0].insert(cols[0].end(), it.key->x, it.key->x + 32);
        cols[1].insert(cols[1].end(), it.key->y, it.key->y + 32);
        for (int k = 0; k < 4; k++) cols[2 + k].insert(cols[2 + k].end(), sf.f[k], sf.f[k] + 32);
        issuer.push_back((uint32_t)it.ipk->issuer_id);
        if (it.dlen) arena.insert(arena.end(), it.digest, it.digest + it.dlen);
        off.push_back((uint32_t)arena.size());
    }
    const size_t m = idx.size();
    if (m == 0) return Error();
    std::vector<uint64_t> bits((m + 63) / 64);
    std::vector<uint8_t> st(m);
    if (arena.empty()) arena.push_back(0);
    int rc = fabgpu_idemix_nym_verify_batch(ctx_, m, arena.data(), off.data(), issuer.data(), cols[0].data(), cols[1].data(), cols[2].data(),
                                            cols[3].data(), cols[4].data(), cols[5].data(), bits.data(), st.data());
    if (rc != FABGPU_OK) {
        for (size_t j = 0; j < m; j++) results[idx[j]].infrastructure = true;
        return Error(std::string("fabgpu: ") + fabgpu_strerror(rc));
    }
    for (size_t j = 0; j < m; j++) {
        VerifyResult& r = results[idx[j]];
        switch (st[j]) {
            case FABGPU_NYM_VALID: r.valid = true; break;
            case FABGPU_NYM_BAD_PROOF: r.err = Error("pseudonym signature invalid: zero-knowledge proof is invalid"); break;   // idemix/nymsignature.go:105
            default: r.needs_sw = true;
        }
    }
    return Error();
}

}  // namespace bccsp
}  // namespace fab
