"""Pins the CPU oracle (oracle/) before anything is allowed to trust it:
  * the reference's own certificate fixtures (positive KATs),
  * the literal DER vectors of bccsp/sw/impl_test.go:931-964 and the sign/zero cases of
    bccsp/utils/ecdsa_test.go:20-54,
  * the low-S boundary of bccsp/utils/ecdsa_test.go:64-88,
  * agreement pure-Python == plain-C == OpenSSL on seeded random and adversarial tuples."""
import hashlib
import json
import os

import numpy as np
import pytest

import bccsp_sw_oracle as po
import coracle
import ossl_check

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return json.load(open(os.path.join(G, name)))["vectors"]


def _h32(x):
    return bytes.fromhex(x.rjust(64, "0"))


def test_reference_cert_fixture_kats_pinned_positive():
    vs = [v for v in _load("ref_cert_kats.json") if v["pinned_by"]]
    assert len(vs) >= 60
    for v in vs:
        qx, qy, e, r, s = (_h32(v[k]) for k in ("qx", "qy", "e", "r", "s"))
        # the certificate signature is arithmetically valid under the issuer key ...
        assert po.ecdsa_verify_raw(int(v["qx"], 16), int(v["qy"], 16), e, int(v["r"], 16), int(v["s"], 16)), v["source"]
        # ... and bccsp/sw accepts it iff it is low-S (msp/cert.go:76-116 normalises the rest)
        want = po.ST_VALID if v["low_s"] else po.ST_HIGH_S
        assert coracle.verify_one(qx, qy, e, r, s) == want, v["source"]
        sig = bytes.fromhex(v["sig_der"])
        assert coracle.bccsp_verify(qx, qy, sig, e) == want
        if v["low_s"]:
            assert po.csp_verify((int(v["qx"], 16), int(v["qy"], 16)), sig, e) is True
        else:
            with pytest.raises(po.BCCSPError, match="Invalid S. Must be smaller than half the order"):
                po.csp_verify((int(v["qx"], 16), int(v["qy"], 16)), sig, e)


def test_reference_cert_fixture_cross_impl_vectors():
    for v in _load("ref_cert_kats.json"):
        if v["pinned_by"]:
            continue
        qx, qy, e, r, s = (_h32(v[k]) for k in ("qx", "qy", "e", "r", "s"))
        st = coracle.verify_one(qx, qy, e, r, s)
        if v["low_s"]:
            assert st == (po.ST_VALID if v["expect_valid"] else po.ST_BAD_MATH)
        else:
            assert st == po.ST_HIGH_S
        assert po.status_raw(int(v["qx"], 16), int(v["qy"], 16), e, int(v["r"], 16), int(v["s"], 16)) == st


def test_edge_vectors_python_c_agree():
    for v in _load("edge_kats.json"):
        r, s = int(v["r"], 16), int(v["s"], 16)
        e = bytes.fromhex(v["e"])
        assert po.status_raw(int(v["qx"], 16), int(v["qy"], 16), e, r, s) == v["status"], v["name"]
        if 0 <= r < 1 << 256 and 0 <= s < 1 << 256:
            got = coracle.verify_one(_h32(v["qx"]), _h32(v["qy"]), e, r.to_bytes(32, "big"), s.to_bytes(32, "big"))
            assert got == v["status"], v["name"]


def test_low_s_boundary_like_reference_tests():
    # bccsp/utils/ecdsa_test.go:64-88: s == n>>1 is low, (n>>1)+1 is not
    assert po.is_low_s(po.HALF_N) and not po.is_low_s(po.HALF_N + 1)
    assert po.HALF_N == 0x7fffffff800000007fffffffffffffffde737d56d38bcf4279dce5617e3192a8
    by = {v["name"]: v for v in _load("edge_kats.json")}
    assert by["s_eq_half_n"]["status"] == po.ST_VALID
    assert by["s_eq_half_n_plus_1"]["status"] == po.ST_HIGH_S


def test_der_vectors():
    for v in _load("der_kats.json"):
        raw = bytes.fromhex(v["der"])
        rc, r, s, fl = coracle.der_unmarshal(raw)
        if v["ok"]:
            assert rc == 0, v["name"]
            R, S = int(v["r"], 16), int(v["s"], 16)
            assert int.from_bytes(r, "big") == R % (1 << 256) and int.from_bytes(s, "big") == S % (1 << 256)
            assert fl == (1 if R >> 256 else 0) | (2 if S >> 256 else 0)
            assert po.unmarshal_ecdsa_signature(raw) == (R, S)
        else:
            assert rc != 0, v["name"]
            with pytest.raises(po.BCCSPError):
                po.unmarshal_ecdsa_signature(raw)
            if "larger than zero" in v["err"]:
                assert rc in (2, 3)
    # the five literal negatives of bccsp/sw/impl_test.go:931-964 are present
    assert sum(1 for v in _load("der_kats.json") if v["name"].startswith("ref_impl_test_negative")) == 5


def test_bccsp_verify_argument_errors():
    # bccsp/sw/impl.go:249-257 and sw_test.go:130-149
    with pytest.raises(po.BCCSPError, match="Invalid Key. It must not be nil."):
        po.csp_verify(None, b"\x01", b"\x01")
    with pytest.raises(po.BCCSPError, match="Invalid signature. Cannot be empty."):
        po.csp_verify((1, 1), b"", b"\x01")
    with pytest.raises(po.BCCSPError, match="Invalid digest. Cannot be empty."):
        po.csp_verify((1, 1), b"\x01", b"")
    with pytest.raises(po.BCCSPError, match="Failed unmashalling signature \\["):
        po.csp_verify((po.GX, po.GY), b"\x30\x00", b"\x01")
    assert coracle.bccsp_verify(b"\0" * 32, b"\0" * 32, b"", b"x") == 10
    assert coracle.bccsp_verify(b"\0" * 32, b"\0" * 32, b"x", b"") == 11
    assert coracle.bccsp_verify(b"\0" * 32, b"\0" * 32, b"\x30\x00", b"x") == 12


def test_sha256_matches_hashlib_like_reference_TestSHA():
    # bccsp/sw/impl_test.go:1293-1339: random messages of every length 0..99 (+ block boundaries)
    rng = np.random.default_rng(7)
    for ln in list(range(0, 100)) + [119, 120, 127, 128, 1855, 1856, 4608, 65537]:
        m = rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
        assert coracle.sha256(m) == hashlib.sha256(m).digest() == po.csp_hash(m)
    lens = rng.integers(0, 300, size=257)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    arena = rng.integers(0, 256, size=int(off[-1]) + 1, dtype=np.uint8)
    d = coracle.sha256_batch(arena, off)
    for i in range(257):
        assert d[i].tobytes() == hashlib.sha256(arena[off[i]:off[i + 1]].tobytes()).digest()


def test_random_batch_c_oracle_equals_openssl_and_python():
    b = coracle.make_batch(600, seed=20260921, invalid_frac=0.2)
    st = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    so = coracle.ossl_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (st == so).all()
    want = np.array([po.ST_VALID, po.ST_BAD_MATH, po.ST_BAD_MATH, po.ST_HIGH_S, po.ST_BAD_MATH], dtype=np.uint8)[b["kind"]]
    assert (st == want).all()
    for i in range(0, 600, 25):  # pure-Python restatement on a sample
        t = [int.from_bytes(b[k][i].tobytes(), "big") for k in ("qx", "qy", "r", "s")]
        assert po.status_raw(t[0], t[1], b["e"][i].tobytes(), t[2], t[3]) == st[i]
        if st[i] in (0, 1):
            assert ossl_check.verify_raw(t[0], t[1], b["e"][i].tobytes(), t[2], t[3]) == (st[i] == 0)


def test_identity_verify_flow():
    # msp/identities.go:169-196 + msp/msp_test.go:494-536: sign, verify, tamper message, tamper signature
    d, k = 0x1234567, 0x7654321
    Q = po.pt_mul(d, (po.GX, po.GY))
    msg = b"hello world, this is a proposal response payload" * 30
    r, s = po.sign_raw(d, hashlib.sha256(msg).digest(), k)
    sig = po.marshal_ecdsa_signature(r, s)
    assert po.identity_verify(Q, msg, sig) is None
    assert po.identity_verify(Q, msg + b"x", sig) == "The signature is invalid"
    bad = po.marshal_ecdsa_signature(r, po.N - s)
    assert "could not determine the validity of the signature" in po.identity_verify(Q, msg, bad)
    qx, qy = Q[0].to_bytes(32, "big"), Q[1].to_bytes(32, "big")
    assert coracle.bccsp_verify(qx, qy, sig, hashlib.sha256(msg).digest()) == 0
    assert coracle.bccsp_verify(qx, qy, sig, hashlib.sha256(msg + b"x").digest()) == 1
    assert coracle.bccsp_verify(qx, qy, bad, hashlib.sha256(msg).digest()) == 2


def test_rfc6979_public_vectors_pin_the_oracle():
    """RFC 6979 A.2.5 (P-256 / SHA-256): vectors published independently of the reference and of this repository."""
    f = json.load(open(os.path.join(G, "rfc6979_p256_sha256.json")))
    qx, qy, d = int(f["qx"], 16), int(f["qy"], 16), int(f["private_key"], 16)
    assert po.pt_mul(d, (po.GX, po.GY)) == (qx, qy)
    for v in f["vectors"]:
        dig = hashlib.sha256(v["message"].encode()).digest()
        r, s = int(v["r"], 16), int(v["s"], 16)
        assert po.ecdsa_verify_raw(qx, qy, dig, r, s)
        want = po.ST_VALID if po.is_low_s(s) else po.ST_HIGH_S
        assert coracle.verify_one(_h32(f["qx"]), _h32(f["qy"]), dig, _h32(v["r"]), _h32(v["s"])) == want
        assert coracle.verify_one(_h32(f["qx"]), _h32(f["qy"]), dig, _h32(v["r"]), (po.N - s).to_bytes(32, "big")) == \
            (po.ST_VALID if po.is_low_s(po.N - s) else po.ST_HIGH_S)
        assert coracle.verify_one(_h32(f["qx"]), _h32(f["qy"]), hashlib.sha256(b"x" + v["message"].encode()).digest(), _h32(v["r"]),
                                  min(s, po.N - s).to_bytes(32, "big")) == po.ST_BAD_MATH
