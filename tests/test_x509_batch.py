"""x509 certificate signatures in batch: crypto/x509 CheckSignatureFrom - the ECDSA part - on the device.

Goldens: every ecdsa-with-SHA256 certificate of the reference tree with its candidate issuers (tests/golden/ref_cert_chains.json, made by
gen_ref_cert_chain_fixtures.py): 63 pairs pinned valid by AuthorityKeyIdentifier == SubjectKeyIdentifier or self-signature, the rest
decided by OpenSSL (valid chains and genuine wrong-key negatives).  69 of the signatures are high-S: valid here, an error in bccsp/sw."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import bccsp_sw_oracle as po
import fabgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_cert_chains.json")))
CERTS = {h: base64.b64decode(b) for h, b in FX["certs"].items()}
VEC = FX["vectors"]


def issuer_key(h):
    k = fabgpu.x509_p256_pubkey(CERTS[h], pem=False)
    assert k is not None
    return k


def test_fixture_inventory_and_the_host_side_certificate_walk():
    assert len(VEC) == 218 and sum(1 for v in VEC if v["pinned_by"]) == 63 and sum(1 for v in VEC if not v["low_s"]) >= 60
    assert sum(1 for v in VEC if v["expect_valid"]) == 96
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_ref_cert_kats as g
    for h, der in CERTS.items():
        assert hashlib.sha256(der).hexdigest() == h
        parts = fabgpu.x509_signature_parts(der)
        c = g.parse_cert(der)                                   # the KAT generator's independent DER reader
        assert parts is not None and parts[0] == c["tbs"] and parts[1] == c["sig"] and parts[2] == (c["alg"] == g.OID_ECDSA_SHA256)
    assert fabgpu.x509_signature_parts(b"\x30\x03\x02\x01\x01") is None and fabgpu.x509_signature_parts(b"") is None


def test_oracle_reproduces_every_expected_verdict_with_x509_semantics():
    """x509: r, s in [1, n-1] and the plain equation - NO low-S rule (crypto/x509 checkSignature calls ecdsa.Verify directly)."""
    for v in VEC:
        tbs, sig, alg = fabgpu.x509_signature_parts(CERTS[v["cert"]])
        qx, qy = issuer_key(v["issuer"])
        r, s = po.asn1_unmarshal_ecdsa_sig(sig)
        ok = po.ecdsa_verify_raw(int.from_bytes(qx, "big"), int.from_bytes(qy, "big"), hashlib.sha256(tbs).digest(), r, s)
        assert ok == v["expect_valid"], v["source"]
        assert (s <= po.N >> 1) == v["low_s"]


@pytest.mark.gpu
def test_x509_batch_on_the_reference_certificates():
    csp = fabgpu.GPUCSP()
    try:
        certs = [CERTS[v["cert"]] for v in VEC]
        keys = [issuer_key(v["issuer"]) for v in VEC]
        st = fabgpu.x509_check_signature_batch(csp, certs, keys)
        assert [int(x) for x in st] == [0 if v["expect_valid"] else 1 for v in VEC]
        assert all(st[i] == 0 for i, v in enumerate(VEC) if v["pinned_by"])
        # the same high-S certificates through identity.Verify semantics are an ERROR (bccsp/sw low-S rule): the two paths must differ
        hi = [i for i, v in enumerate(VEC) if not v["low_s"] and v["expect_valid"]]
        assert len(hi) > 10
        for i in hi[:5]:
            tbs, sig, _ = fabgpu.x509_signature_parts(certs[i])
            errs = csp.identity_verify_batch([fabgpu.ECDSAPublicKey(int.from_bytes(keys[i][0], "big"), int.from_bytes(keys[i][1], "big"))], [tbs], [sig])
            assert errs[0] is not None and "Invalid S" in errs[0]
        # mutations: a TBS bit, a signature bit, trailing bytes after the signature, another algorithm, an off-curve issuer key
        i0 = next(i for i, v in enumerate(VEC) if v["pinned_by"])
        good, key = certs[i0], keys[i0]
        tbs, sig, _ = fabgpu.x509_signature_parts(good)
        t_off, s_off = good.index(tbs), good.rindex(sig)
        m_tbs = bytearray(good); m_tbs[t_off + len(tbs) // 2] ^= 4
        m_sig = bytearray(good); m_sig[s_off + len(sig) - 1] ^= 1
        oid = bytes.fromhex("2a8648ce3d040302")
        a_off = good.index(oid, t_off + len(tbs))                              # the OUTER signatureAlgorithm (after the TBS)
        m_alg = bytearray(good); m_alg[a_off + 7] = 0x03                        # ecdsa-with-SHA384
        # a byte after the signature SEQUENCE, inside the BIT STRING ("x509: trailing data after ECDSA signature"): lengths rebuilt by hand
        assert good[s_off - 3] == 0x03 and good[s_off - 2] == len(sig) + 1 and good[1] == 0x82      # BIT STRING short form, outer SEQUENCE 2-byte length
        m_trail = bytearray(good[:s_off - 2] + bytes([len(sig) + 2]) + good[s_off - 1:] + b"\x00")
        total = int.from_bytes(good[2:4], "big") + 1
        m_trail[2:4] = total.to_bytes(2, "big")
        assert fabgpu.x509_signature_parts(bytes(m_trail))[1] == sig + b"\x00"
        st2 = fabgpu.x509_check_signature_batch(csp, [good, bytes(m_tbs), bytes(m_sig), bytes(m_alg), good, b"\x30\x00", b"", bytes(m_trail)],
                                                [key, key, key, key, (b"\x00" * 31 + b"\x01", b"\x00" * 31 + b"\x01"), key, key, key])
        assert [int(x) for x in st2] == [0, 1, 1, 6, 6, 6, 6, 5]
        assert fabgpu.x509_check_signature_batch(csp, [], []).size == 0
    finally:
        csp.close()
