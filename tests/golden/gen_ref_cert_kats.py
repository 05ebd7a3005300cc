#!/usr/bin/env python3
"""Extract ECDSA-P256/SHA-256 known-answer vectors from the reference's own X.509 fixtures.

Every ECDSA-signed certificate under /root/reference is a positive KAT
(Q_issuer, SHA-256(TBSCertificate), r, s) produced by the reference's own tooling
(SURVEY.md section 8(c)).  This script walks the tree, parses the PEM certificates with a
small DER reader, pairs each certificate with its issuer's public key (issuer DN == subject DN
of another fixture, or itself when self-signed) and writes the tuples to
tests/golden/ref_cert_kats.json.  It runs only in the build container (the reference tree does
not exist on the GPU box); the JSON it produced is committed.

The verdict recorded in the JSON is NOT computed here: each vector is "a signature the
reference's fixtures assert to be valid" (msp setup validates these chains,
msp/mspimplvalidate.go:21-52); `low_s` records whether bccsp/sw would accept it as-is or only
after msp/cert.go:76-116 sanitizeECDSASignedCert normalised it.
"""
import base64
import hashlib
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_cert_kats.json")

OID_EC_PUBKEY = bytes.fromhex("2a8648ce3d0201")
OID_P256 = bytes.fromhex("2a8648ce3d030107")
OID_ECDSA_SHA256 = bytes.fromhex("2a8648ce3d040302")
OID_SKI = bytes.fromhex("551d0e")
OID_AKI = bytes.fromhex("551d23")
HALF_N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551 >> 1


def tlv(buf, off):
    """Lenient DER TLV reader: returns (tag_byte, header_len, length)."""
    tag = buf[off]
    b = buf[off + 1]
    if b < 0x80:
        return tag, 2, b
    nb = b & 0x7F
    return tag, 2 + nb, int.from_bytes(buf[off + 2:off + 2 + nb], "big")


def children(buf, off, end):
    out = []
    while off < end:
        tag, hl, ln = tlv(buf, off)
        out.append((tag, off, hl, ln))
        off += hl + ln
    return out


def parse_cert(der):
    tag, hl, ln = tlv(der, 0)
    assert tag == 0x30
    top = children(der, hl, hl + ln)
    tbs_tag, tbs_off, tbs_hl, tbs_ln = top[0]
    tbs_raw = der[tbs_off:tbs_off + tbs_hl + tbs_ln]
    _, a_off, a_hl, a_ln = top[1]
    alg = children(der, a_off + a_hl, a_off + a_hl + a_ln)
    alg_oid = der[alg[0][1] + alg[0][2]:alg[0][1] + alg[0][2] + alg[0][3]]
    _, s_off, s_hl, s_ln = top[2]
    sig_der = der[s_off + s_hl + 1:s_off + s_hl + s_ln]  # skip unused-bits octet
    f = children(der, tbs_off + tbs_hl, tbs_off + tbs_hl + tbs_ln)
    i = 1 if f[0][0] == 0xA0 else 0  # [0] version
    # serial, sigalg, issuer, validity, subject, spki
    issuer = der[f[i + 2][1]:f[i + 2][1] + f[i + 2][2] + f[i + 2][3]]
    subject = der[f[i + 4][1]:f[i + 4][1] + f[i + 4][2] + f[i + 4][3]]
    _, k_off, k_hl, k_ln = f[i + 5]
    spki = children(der, k_off + k_hl, k_off + k_hl + k_ln)
    kalg = children(der, spki[0][1] + spki[0][2], spki[0][1] + spki[0][2] + spki[0][3])
    k_oid = der[kalg[0][1] + kalg[0][2]:kalg[0][1] + kalg[0][2] + kalg[0][3]]
    curve = der[kalg[1][1] + kalg[1][2]:kalg[1][1] + kalg[1][2] + kalg[1][3]] if len(kalg) > 1 else b""
    bits = der[spki[1][1] + spki[1][2] + 1:spki[1][1] + spki[1][2] + spki[1][3]]
    pub = None
    if k_oid == OID_EC_PUBKEY and curve == OID_P256 and len(bits) == 65 and bits[0] == 4:
        pub = (bits[1:33].hex(), bits[33:65].hex())
    ski = aki = None
    for tag, off, hl, ln in f[i + 6:]:
        if tag != 0xA3:  # [3] extensions
            continue
        seq = children(der, off + hl, off + hl + ln)[0]
        for _, e_off, e_hl, e_ln in children(der, seq[1] + seq[2], seq[1] + seq[2] + seq[3]):
            parts = children(der, e_off + e_hl, e_off + e_hl + e_ln)
            oid = der[parts[0][1] + parts[0][2]:parts[0][1] + parts[0][2] + parts[0][3]]
            val = parts[-1]
            body = der[val[1] + val[2]:val[1] + val[2] + val[3]]
            if oid == OID_SKI:
                _, h2, l2 = tlv(body, 0)
                ski = body[h2:h2 + l2].hex()
            elif oid == OID_AKI:
                _, h2, l2 = tlv(body, 0)
                for t3, o3, h3, l3 in children(body, h2, h2 + l2):
                    if t3 == 0x80:  # [0] keyIdentifier
                        aki = body[o3 + h3:o3 + h3 + l3].hex()
    return dict(tbs=tbs_raw, alg=alg_oid, sig=sig_der, issuer=issuer, subject=subject, pub=pub,
                ski=ski, aki=aki)


def openssl_says(qx_hex, qy_hex, e_hex, r, s):
    """Verdict of OpenSSL libcrypto (ECDSA_do_verify) on the raw tuple."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
    import ossl_check
    return ossl_check.verify_raw(int(qx_hex, 16), int(qy_hex, 16), bytes.fromhex(e_hex), r, s)


def sig_rs(sig_der):
    tag, hl, ln = tlv(sig_der, 0)
    ch = children(sig_der, hl, hl + ln)
    r = int.from_bytes(sig_der[ch[0][1] + ch[0][2]:ch[0][1] + ch[0][2] + ch[0][3]], "big", signed=True)
    s = int.from_bytes(sig_der[ch[1][1] + ch[1][2]:ch[1][1] + ch[1][2] + ch[1][3]], "big", signed=True)
    return r, s


def main():
    certs = {}
    for root, _, files in os.walk(REF):
        for fn in sorted(files):
            if not fn.endswith(".pem"):
                continue
            path = os.path.join(root, fn)
            try:
                txt = open(path, "r", errors="replace").read()
            except OSError:
                continue
            for blk in txt.split("-----BEGIN CERTIFICATE-----")[1:]:
                b64 = blk.split("-----END CERTIFICATE-----")[0]
                try:
                    der = base64.b64decode("".join(b64.split()))
                    c = parse_cert(der)
                except Exception:
                    continue
                c["path"] = os.path.relpath(path, REF)
                certs.setdefault(hashlib.sha256(der).hexdigest(), c)
    by_subject = {}
    for c in certs.values():
        if c["pub"]:
            by_subject.setdefault(c["subject"], []).append(c)
    vectors = []
    for h, c in sorted(certs.items(), key=lambda kv: kv[1]["path"]):
        if c["alg"] != OID_ECDSA_SHA256:
            continue
        cands = by_subject.get(c["issuer"], [])
        r, s = sig_rs(c["sig"])
        # Issuer selection for the PINNED positives never uses signature arithmetic (that would
        # make the KAT circular): AuthorityKeyIdentifier == issuer SubjectKeyIdentifier where the
        # SKI is a real key hash (>= 20 bytes; some fixtures carry the dummy id 01:02:03:04), or a
        # self-signed certificate without AKI.  Every other DN-matched (cert, candidate key) pair
        # is kept as an UNPINNED cross-implementation vector whose expected verdict comes from the
        # OpenSSL libcrypto ECDSA_do_verify (an implementation independent of both oracle and GPU code); those
        # include genuine wrong-key negatives (same DN, regenerated key).
        seen = set()
        for ca in cands:
            if ca["pub"] in seen:
                continue
            seen.add(ca["pub"])
            if c["aki"] is not None and len(c["aki"]) >= 40:
                if ca["ski"] != c["aki"]:
                    continue
                pinned, expect = "aki==ski", True
            elif c["aki"] is None and c["issuer"] == c["subject"] and ca["pub"] == c["pub"]:
                pinned, expect = "self-signed", True
            else:
                pinned, expect = None, openssl_says(ca["pub"][0], ca["pub"][1], hashlib.sha256(c["tbs"]).hexdigest(), r, s)
            vectors.append(dict(
                source=c["path"], issuer_source=ca["path"],
                qx=ca["pub"][0], qy=ca["pub"][1],
                e=hashlib.sha256(c["tbs"]).hexdigest(), tbs_len=len(c["tbs"]),
                r="%064x" % r, s="%064x" % s, sig_der=c["sig"].hex(),
                low_s=bool(s <= HALF_N), pinned_by=pinned, expect_valid=expect))
    json.dump(dict(generator="tests/golden/gen_ref_cert_kats.py", reference_root=REF,
                   n_certs=len(certs), vectors=vectors), open(OUT, "w"), indent=0)
    print("certs", len(certs), "vectors", len(vectors), "->", OUT)


if __name__ == "__main__":
    main()
