"""The idemix oracle (oracle/idemix_oracle.py) against the reference's own key-material fixtures, and the host-side FP256BN
code of the product (the headers the kernel is compiled from, run on the CPU through libfabgpu_hosttest.so) against the
oracle.  No GPU needed."""
import ctypes
import os
import random

import pytest

import fabgpu
import idemix_oracle as io
from idemix_common import ROOT, be32, fixtures, make_batch


@pytest.fixture(scope="module")
def fx():
    return fixtures()


@pytest.fixture(scope="module")
def hosttest():
    p = os.path.join(ROOT, "fabric-mod_amd", "lib", "libfabgpu_hosttest.so")
    if not os.path.exists(p):
        import __graft_entry__ as g
        g.build()
    L = ctypes.CDLL(p)
    L.hosttest_bn_issuer_new.restype = ctypes.c_void_p
    L.hosttest_bn_issuer_new.argtypes = [ctypes.c_char_p] * 4
    L.hosttest_bn_issuer_free.argtypes = [ctypes.c_void_p]
    L.hosttest_bn_tab_entry.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    L.hosttest_bn_nym_commitment.argtypes = [ctypes.c_void_p] + [ctypes.c_char_p] * 7
    L.hosttest_bn_nym_commitment_split.argtypes = [ctypes.c_void_p] + [ctypes.c_char_p] * 7
    return L


# ---- pins: the reference's fixtures ---------------------------------------------------------------------------------------
def test_every_fixture_point_is_on_the_curve(fx):
    for name, e in fx.items():
        ipk = e["ipk"]
        pts = [ipk.h_sk, ipk.h_rand, ipk.bar_g1, ipk.bar_g2] + ipk.h_attrs
        assert all(io.g1_on_curve(p) for p in pts), name
        if "signer" in e:
            assert io.g1_on_curve(e["signer"].cred.a) and io.g1_on_curve(e["signer"].cred.b)


def test_twist_and_g2_generator_are_consistent_across_fixtures(fx):
    name, _ = io.pin_twist([e["ipk"].w for e in fx.values()])
    assert name == "M-type b*xi"
    gens = {io.derive_gen_g2(e["ipk"].w, e["isk"]) for e in fx.values()}
    assert len({e["isk"] for e in fx.values()}) == 2   # the five directories hold two distinct issuer keys ...
    assert len(gens) == 1                              # ... and W = GenG2^isk gives ONE generator for both


def test_issuer_public_key_proofs_verify(fx):
    """idemix/issuerkey.go:114-172 on every fixture: pins G1/G2 arithmetic, ToBytes layouts and HashModOrder"""
    g2 = io.derive_gen_g2(fx["MSP1OU1"]["ipk"].w, fx["MSP1OU1"]["isk"])
    for name, e in fx.items():
        assert io.ipk_check(e["ipk"], g2), name
    # and the check is not vacuous
    bad = fx["MSP1OU1"]["ipk"]
    saved = bad.proof_s
    bad.proof_s = be32((int.from_bytes(saved, "big") + 1) % io.R)
    assert not io.ipk_check(bad, g2)
    bad.proof_s = saved


def test_credential_b_values_recompute(fx):
    """idemix/credential.go:110-146: B = g1 * HSk^sk * HRand^s * prod HAttrs^attrs on the reference's credentials"""
    n = 0
    for name, e in fx.items():
        if "signer" in e:
            assert io.credential_b_check(e["signer"].cred, e["signer"].sk, e["ipk"]), name
            n += 1
    assert n == 4


def test_sign_then_verify_and_tamper(fx):
    rng = random.Random(7)
    ipk, sk = fx["MSP1OU1"]["ipk"], fx["MSP1OU1"]["signer"].sk
    nym, r_nym = io.make_nym(sk, ipk, rng)
    sig = io.nym_sign(sk, nym, r_nym, ipk, b"some message", rng)
    assert io.nym_verify(sig, nym, ipk, b"some message") == io.NYM_VALID
    assert io.nym_verify(sig, nym, ipk, b"some messagf") == io.NYM_BAD_PROOF
    raw = io.nym_signature_marshal(sig)
    assert io.nym_signature_unmarshal(raw) == sig
    short = dict(sig, nonce=sig["nonce"][:31])
    assert io.nym_verify(short, nym, ipk, b"some message") == io.NYM_NEEDS_SW


# ---- the product's host code against the oracle ---------------------------------------------------------------------------
def _op(L, k, a, b):
    out = ctypes.create_string_buffer(32)
    L.hosttest_bn29_op(k, be32(a), be32(b), out)
    return int.from_bytes(out.raw, "big")


def test_bn29_field_ops(hosttest):
    P = io.P
    rng = random.Random(11)
    edge = [0, 1, 2, 3, P - 1, P - 2, (1 << 255) % P, (1 << 29) - 1, 1 << 29, (1 << 232) - 1]
    cases = [(a, b) for a in edge for b in edge] + [(rng.randrange(P), rng.randrange(P)) for _ in range(1500)]
    for a, b in cases:
        assert _op(hosttest, 0, a, b) == a * b % P
        assert _op(hosttest, 1, a, b) == a * a % P
        assert _op(hosttest, 2, a, b) == (a + b) * (a - b) % P        # lazy operands, L = 2 x 2
        assert _op(hosttest, 3, a, b) == int(a == b)
        assert _op(hosttest, 5, a, b) == 9 * a * a % P                # (3a)^2, L = 3 x 3
        assert _op(hosttest, 6, a, b) == 4 * a * b % P                # (4a) b, L = 4 x 1
    for a in (P, P + 5, (1 << 256) - 1):                              # any 256-bit input reduces
        assert _op(hosttest, 4, a, 0) == a % P
    out = ctypes.create_string_buffer(32)
    for _ in range(100):
        a = rng.randrange(1, P)
        hosttest.hosttest_bn_modinv(be32(a), out)
        assert int.from_bytes(out.raw, "big") == pow(a, -1, P)


def test_bn_comb_tables_and_commitment(hosttest, fx):
    ipk, sk = fx["MSP1OU1"]["ipk"], fx["MSP1OU1"]["signer"].sk
    h = ctypes.c_void_p(hosttest.hosttest_bn_issuer_new(be32(ipk.h_sk[0]), be32(ipk.h_sk[1]), be32(ipk.h_rand[0]), be32(ipk.h_rand[1])))
    ox, oy = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    try:
        for which, base in ((0, ipk.h_sk), (1, ipk.h_rand)):
            for w, d in ((0, 1), (0, 2), (0, 255), (1, 1), (5, 77), (17, 128), (31, 1), (31, 255)):
                hosttest.hosttest_bn_tab_entry(h, which, w, d, ox, oy)
                assert (int.from_bytes(ox.raw, "big"), int.from_bytes(oy.raw, "big")) == io.g1_mul(base, d << (8 * w)), (which, w, d)

        def commit(nym, c, s1, s2):
            """one lane per signature and two lanes per signature must tell the same story"""
            st = hosttest.hosttest_bn_nym_commitment(h, be32(nym[0]), be32(nym[1]), be32(c), be32(s1), be32(s2), ox, oy)
            t = (int.from_bytes(ox.raw, "big"), int.from_bytes(oy.raw, "big"))
            st2 = hosttest.hosttest_bn_nym_commitment_split(h, be32(nym[0]), be32(nym[1]), be32(c), be32(s1), be32(s2), ox, oy)
            assert st2 == st, (st, st2)
            if st == 0:
                assert t == (int.from_bytes(ox.raw, "big"), int.from_bytes(oy.raw, "big"))
            return st, t

        rng = random.Random(13)
        R = io.R
        lam = (36 * io.U**4 - 1) % R
        special = [(lam, None, None), (lam + 1, None, None), (R - lam, None, None), ((1 << 128) + 1, None, None),
                   (0, None, None), (None, 0, None), (None, None, 0), (None, 0, 0), (R - 1, None, None), (None, R - 1, R - 1), (1, 1, 1),
                   (1 << 255, 1 << 255, 1 << 255), ((1 << 255) - 1, 0xFF << 248, 1 << 248)]
        for it in range(44):
            nym, r_nym = io.make_nym(sk, ipk, rng)
            c, s1, s2 = rng.randrange(R), rng.randrange(R), rng.randrange(R)
            if it < len(special):
                c, s1, s2 = [v if f is None else f for v, f in zip((c, s1, s2), special[it])]
            st, t = commit(nym, c, s1, s2)
            want = io.g1_add(io.g1_mul2(ipk.h_sk, s1, ipk.h_rand, s2), io.g1_neg(io.g1_mul(nym, c)))
            assert st == 0 and t == want, it
        nym, r_nym = io.make_nym(sk, ipk, rng)
        c = rng.randrange(R)
        # exceptional cases of the final additions
        assert commit(nym, c, c * sk % R, c * r_nym % R)[0] == io.NYM_NEEDS_SW                  # t = infinity
        st, t = commit(nym, c, (-c * sk) % R, (-c * r_nym) % R)                                 # U == -c Nym: the last addition doubles
        assert st == 0 and t == io.g1_mul(nym, (-2 * c) % R)
        # domain gates
        assert commit(nym, R, 1, 1)[0] == io.NYM_BAD_PROOF
        assert commit(nym, 5, R, 1)[0] == io.NYM_NEEDS_SW and commit(nym, 5, 1, R + 7)[0] == io.NYM_NEEDS_SW
        assert commit((nym[0], (nym[1] + 1) % io.P), 5, 1, 1)[0] == io.NYM_NEEDS_SW
        assert commit((io.P, nym[1]), 5, 1, 1)[0] == io.NYM_NEEDS_SW
    finally:
        hosttest.hosttest_bn_issuer_free(h)


def test_the_16_bit_comb_table_of_the_device(hosttest, fx):
    """The device's issuer tables are 16-bit combs (16 windows x 65 535 entries, 80 MiB per base), built by the same template as
    the 8-bit ones above: entries d * 2^(16 w) * B at the corners and at random places."""
    ipk = fx["MSP2OU1"]["ipk"]
    hosttest.hosttest_bn_tab16_entry.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    ox, oy = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    rng = random.Random(19)
    places = [(0, 1), (0, 2), (0, 65535), (1, 1), (15, 65535), (15, 1), (7, 32768), (8, 257)] + [(rng.randrange(16), rng.randrange(1, 65536)) for _ in range(12)]
    for w, d in places:
        hosttest.hosttest_bn_tab16_entry(be32(ipk.h_rand[0]), be32(ipk.h_rand[1]), w, d, ox, oy)
        assert (int.from_bytes(ox.raw, "big"), int.from_bytes(oy.raw, "big")) == io.g1_mul(ipk.h_rand, d << (16 * w)), (w, d)


def test_glv_decomposition_and_double_scalar_loop(hosttest, fx):
    """c * Nym on the device goes through k = k1 + k2 lambda; the decomposition must be exact mod r and short, and the
    interleaved loop must report (not hide) a collision between accumulator and addend."""
    R, P = io.R, io.P
    lam = (36 * io.U**4 - 1) % R
    assert (lam * lam + lam + 1) % R == 0
    rng = random.Random(17)
    m1b, m2b = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    worst = 0
    ks = [0, 1, 2, R - 1, R - 2, lam, lam + 1, lam - 1, R // 2, 1 << 255, (1 << 256) - 1, R, R + 1, (1 << 128) - 1, 1 << 128]
    ks += [rng.randrange(1 << 256) for _ in range(20000)]
    for k in ks:
        fl = hosttest.hosttest_bn_glv_decompose(be32(k), m1b, m2b)
        m1, m2 = int.from_bytes(m1b.raw, "big"), int.from_bytes(m2b.raw, "big")
        k1 = -m1 if fl & 1 else m1
        k2 = -m2 if fl & 2 else m2
        assert (k1 + k2 * lam - k) % R == 0, hex(k)
        worst = max(worst, m1, m2)
    assert worst < 1 << 130                       # 27 windows of 5 bits cover 134
    # the endomorphism itself: phi(x, y) = (beta x, y) = lambda (x, y) on fixture points
    ipk = fx["MSP1OU1"]["ipk"]
    hosttest.hosttest_bn_glv_mult.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    ox, oy = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)

    def glv(Q, m1, n1, m2, n2, ident=0):
        fl = hosttest.hosttest_bn_glv_mult(be32(Q[0]), be32(Q[1]), be32(m1), n1, be32(m2), n2, ident, ox, oy)
        return fl, (int.from_bytes(ox.raw, "big"), int.from_bytes(oy.raw, "big"))

    Q = ipk.h_rand
    assert glv(Q, 0, 0, 1, 0) == (0, io.g1_mul(Q, lam))
    assert glv(Q, 0, 0, 1, 1) == (0, io.g1_neg(io.g1_mul(Q, lam)))
    for _ in range(12):
        m1, m2 = rng.randrange(1 << 129), rng.randrange(1 << 129)
        n1, n2 = rng.randrange(2), rng.randrange(2)
        want = io.g1_mul(Q, ((-m1 if n1 else m1) + (-m2 if n2 else m2) * lam) % R)
        assert glv(Q, m1, n1, m2, n2) == (0, want)
    assert glv(Q, 0, 0, 0, 0)[0] & 1                                  # infinity
    # collisions, made constructible by an identity "endomorphism": Q + Q and Q - Q inside one window
    assert glv(Q, 1, 0, 1, 0, ident=1)[0] & 2
    assert glv(Q, 1, 0, 1, 1, ident=1)[0] & 2
    assert glv(Q, 5, 0, 3, 0, ident=1) == (0, io.g1_mul(Q, 8))      # no collision: plain sum


def test_g1_gate_of_the_c_abi(fx):
    ipk = fx["MSP2OU1"]["ipk"]
    assert fabgpu.bn256_g1_on_curve(be32(ipk.h_sk[0]), be32(ipk.h_sk[1]))
    assert not fabgpu.bn256_g1_on_curve(be32(ipk.h_sk[0]), be32((ipk.h_sk[1] + 1) % io.P))
    assert not fabgpu.bn256_g1_on_curve(be32(io.P), be32(2))
    assert fabgpu.bn256_g1_on_curve(be32(1), be32(2))


def test_batch_maker_covers_every_kind(fx):
    issuers = [(fx[n]["ipk"], fx[n]["signer"].sk) for n in ("MSP1OU1", "MSP2OU1")]
    b = make_batch(issuers, 120, 3)
    kinds = set(b.what)
    assert {"valid", "msg bit", "other nym", "other issuer", "c >= r", "s >= r", "nym off curve", "t at infinity"} <= kinds
    ex = dict(zip(b.what, b.expect))
    assert ex["valid"] == 0 and ex["c >= r"] == 1 and ex["s >= r"] == 6 and ex["nym off curve"] == 6 and ex["t at infinity"] == 6
