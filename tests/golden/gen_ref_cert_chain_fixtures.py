#!/usr/bin/env python3
"""Certificates of the reference tree as golden vectors for the x509 batch check (test infrastructure).

    python3 tests/golden/gen_ref_cert_chain_fixtures.py [/root/reference]  ->  tests/golden/ref_cert_chains.json

Same selection rules as gen_ref_cert_kats.py (whose parser this script imports): every ecdsa-with-SHA256 certificate under the tree,
paired with every DN-matched candidate issuer certificate; a pair is PINNED valid without any signature arithmetic when the child's
AuthorityKeyIdentifier equals the issuer's SubjectKeyIdentifier (real key hashes only) or the certificate is self-signed; every
other pair's expected verdict is OpenSSL's ECDSA_do_verify (valid chains and genuine wrong-key negatives: same DN, regenerated key).
Unlike ref_cert_kats.json (flattened (Q, e, r, s) tuples) this file carries the DER CERTIFICATES themselves: the x509 entry point
parses them (TBS span, signature algorithm, BIT STRING) and hashes the TBS on the device.  Note: 69 of the signatures are high-S -
valid for crypto/x509, which is why msp sanitizes certificates (msp/cert.go:76-116) and why this path differs from bccsp/sw."""
import base64
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_ref_cert_kats as g   # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    certs = {}
    for root, _, files in os.walk(ref):
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
                    c = g.parse_cert(der)
                except Exception:
                    continue
                c["path"] = os.path.relpath(path, ref)
                c["der"] = der
                certs.setdefault(hashlib.sha256(der).hexdigest(), c)
    by_subject = {}
    for h, c in certs.items():
        if c["pub"]:
            by_subject.setdefault(c["subject"], []).append((h, c))
    used, vectors = {}, []
    for h, c in sorted(certs.items(), key=lambda kv: kv[1]["path"]):
        if c["alg"] != g.OID_ECDSA_SHA256:
            continue
        r, s = g.sig_rs(c["sig"])
        seen = set()
        for ih, ca in by_subject.get(c["issuer"], []):
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
                pinned, expect = None, g.openssl_says(ca["pub"][0], ca["pub"][1], hashlib.sha256(c["tbs"]).hexdigest(), r, s)
            used[h], used[ih] = c, ca
            vectors.append(dict(cert=h, issuer=ih, source=c["path"], issuer_source=ca["path"], pinned_by=pinned, expect_valid=bool(expect),
                                low_s=bool(s <= g.HALF_N)))
    out = dict(generator="tests/golden/gen_ref_cert_chain_fixtures.py",
               certs={h: base64.b64encode(c["der"]).decode() for h, c in sorted(used.items())}, vectors=vectors)
    path = os.path.join(HERE, "ref_cert_chains.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("%d certificates, %d (child, issuer) vectors (%d pinned valid, %d high-S) -> %s (%d bytes)" % (
        len(used), len(vectors), sum(1 for v in vectors if v["pinned_by"]), sum(1 for v in vectors if not v["low_s"]), path, os.path.getsize(path)))


if __name__ == "__main__":
    main()
