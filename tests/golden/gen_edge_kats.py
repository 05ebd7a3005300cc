#!/usr/bin/env python3
"""Build boundary / exceptional-case vectors for the flattened (Qx,Qy,e,r,s) boundary and the
DER signature gate.  Expected results come from the pure-Python oracle
(oracle/bccsp_sw_oracle.py) and every arithmetic verdict is asserted equal to OpenSSL
libcrypto's before it is written (oracle/ossl_check.py).  Output (committed):
  tests/golden/edge_kats.json   -- tuple-level vectors with expected status codes
  tests/golden/der_kats.json    -- DER byte strings with expected parse result / verify class

Cases mirror what the reference tests pin as properties:
  low-S boundary s = n>>1 / (n>>1)+1      bccsp/utils/ecdsa_test.go:64-88, sw/ecdsa_test.go:66-73
  r,s in {-1,0}                           bccsp/utils/ecdsa_test.go:32-54
  DER negatives (literal bytes)           bccsp/sw/impl_test.go:931-964
  tampered message / signature            msp/msp_test.go:532-535
plus the group-law corner cases of SURVEY Appendix A step 8 that random data never reaches.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bccsp_sw_oracle as o  # noqa: E402
import ossl_check  # noqa: E402

rng = random.Random(20260921)
G = (o.GX, o.GY)


def neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % o.P)


def lift_x(x):
    """A curve point with the given x (p = 3 mod 4), or None."""
    rhs = (x * x * x + o.A * x + o.B) % o.P
    y = pow(rhs, (o.P + 1) // 4, o.P)
    return (x, y) if y * y % o.P == rhs else None


def from_R_ab(R, a, b):
    """Tuple that makes the verifier compute u1=a, u2=b and land on point R (Q has unknown dlog)."""
    Q = o.pt_mul(pow(b, -1, o.N), o.pt_add(R, neg(o.pt_mul(a, G))))
    r = R[0] % o.N
    s = r * pow(b, -1, o.N) % o.N
    e = a * s % o.N
    return Q, e, r, s


def from_d_ab(d, a, b):
    """Known key Q=dG, chosen u1=a, u2=b: R = (a + b d) G (may be infinity)."""
    Q = o.pt_mul(d, G)
    R = o.pt_add(o.pt_mul(a, G), o.pt_mul(b, Q))
    return Q, R


vectors = []


def emit(name, Q, digest, r, s, note=""):
    qx, qy = Q
    st = o.status_raw(qx, qy, digest, r, s)
    if st in (o.ST_VALID, o.ST_BAD_MATH) and 0 < r < 1 << 256 and 0 < s < 1 << 256:
        ov = ossl_check.verify_raw(qx, qy, digest, r, s)
        # OpenSSL reports an internal error (-1 -> None) when the sum is the point at infinity
        assert ov == (st == o.ST_VALID) or (ov is None and st == o.ST_BAD_MATH), (name, ov, st)
    vectors.append(dict(name=name, qx="%064x" % qx, qy="%064x" % qy, e=digest.hex(),
                        r="%x" % r if r >= 0 else "-%x" % -r, s="%x" % s if s >= 0 else "-%x" % -s,
                        status=st, note=note))
    return st


def b32(v):
    return v.to_bytes(32, "big")


def rand_scalar():
    return rng.randrange(1, o.N)


def signed_case(name, digest, d=None, k=None, low_s=True):
    d = d or rand_scalar()
    k = k or rand_scalar()
    Q = o.pt_mul(d, G)
    r, s = o.sign_raw(d, digest, k, low_s=low_s)
    return Q, r, s


# 1. plain random valid + tamper family ---------------------------------------------------
for i in range(8):
    dg = rng.randbytes(32)
    Q, r, s = signed_case("rand", dg)
    assert emit("random_valid_%d" % i, Q, dg, r, s) == o.ST_VALID
    if i < 3:
        bad = bytearray(dg)
        bad[rng.randrange(32)] ^= 1 << rng.randrange(8)
        assert emit("flipped_digest_bit_%d" % i, Q, bytes(bad), r, s) == o.ST_BAD_MATH
        Q2 = o.pt_mul(rand_scalar(), G)
        assert emit("wrong_key_%d" % i, Q2, dg, r, s) == o.ST_BAD_MATH
        assert emit("r_plus_1_%d" % i, Q, dg, r + 1, s) == o.ST_BAD_MATH
        assert emit("high_s_mirror_%d" % i, Q, dg, r, o.N - s) == o.ST_HIGH_S

# 2. low-S boundary -------------------------------------------------------------------------
for name, sval, want in (("s_eq_half_n", o.HALF_N, o.ST_VALID), ("s_eq_half_n_plus_1", o.HALF_N + 1, o.ST_HIGH_S),
                         ("s_eq_1", 1, o.ST_VALID), ("s_eq_2", 2, o.ST_VALID)):
    dg = rng.randbytes(32)
    k = rand_scalar()
    r = o.pt_mul(k, G)[0] % o.N
    d = (sval * k - o.hash_to_int(dg)) * pow(r, -1, o.N) % o.N
    assert emit(name, o.pt_mul(d, G), dg, r, sval, "key solved so that this s is arithmetically valid") == want

# 3. range gates -----------------------------------------------------------------------------
dg = rng.randbytes(32)
Q, r, s = signed_case("range", dg)
for name, rr, ss, want in (("r_zero", 0, s, o.ST_RANGE), ("s_zero", r, 0, o.ST_RANGE),
                           ("r_eq_n", o.N, s, o.ST_RANGE), ("r_eq_n_plus_r", o.N + r, s, o.ST_RANGE),
                           ("r_max", (1 << 256) - 1, s, o.ST_RANGE),
                           ("s_eq_n", r, o.N, o.ST_HIGH_S), ("s_eq_n_minus_1", r, o.N - 1, o.ST_HIGH_S),
                           ("s_max", r, (1 << 256) - 1, o.ST_HIGH_S),
                           ("r_and_s_zero", 0, 0, o.ST_RANGE),
                           ("r_ge_n_and_high_s", o.N, o.N - 1, o.ST_HIGH_S)):
    assert emit(name, Q, dg, rr, ss) == want, name

# 4. digest shapes (hashToInt) -------------------------------------------------------------
for name, dg in (("digest_all_zero", bytes(32)), ("digest_eq_n", b32(o.N)), ("digest_eq_n_plus_1", b32(o.N + 1)),
                 ("digest_all_ff", b"\xff" * 32), ("digest_1_byte", b"\x7f"), ("digest_1_zero_byte", b"\x00"),
                 ("digest_20_bytes", rng.randbytes(20)), ("digest_31_bytes", rng.randbytes(31)),
                 ("digest_33_bytes", rng.randbytes(33)), ("digest_48_bytes", rng.randbytes(48)),
                 ("digest_64_bytes", rng.randbytes(64))):
    Q, r, s = signed_case(name, dg)
    assert emit(name, Q, dg, r, s, "signed over hashToInt(digest)") == o.ST_VALID
    if len(dg) > 32:
        # bytes past the 32nd must not matter; the 32nd must
        t = bytearray(dg); t[-1] ^= 0xFF
        assert emit(name + "_tail_changed", Q, bytes(t), r, s) == o.ST_VALID
        t = bytearray(dg); t[31] ^= 0x01
        assert emit(name + "_byte31_changed", Q, bytes(t), r, s) == o.ST_BAD_MATH

# 5. chosen (u1,u2) patterns: table / window corner cases -----------------------------------
patterns = [("u1_1_u2_1", 1, 1), ("u1_1", 1, None), ("u2_1", None, 1), ("u1_nm1", o.N - 1, None),
            ("u2_nm1", None, o.N - 1), ("u1_nm1_u2_nm1", o.N - 1, o.N - 1),
            ("u2_pow2_255", None, 1 << 255), ("u2_pow2_252", None, 1 << 252), ("u2_15", None, 15), ("u2_16", None, 16),
            ("u2_17", None, 17), ("u1_16", 16, None), ("u1_pow2_128", 1 << 128, None),
            ("u2_low_window_only", None, 0xF), ("u2_top_window_only", None, 0xF << 252),
            ("u1_top_window_only", 0xF << 252, None), ("u2_alternating", None, int("f0" * 32, 16) % o.N),
            ("u1_alternating", int("0f" * 32, 16), None), ("u2_all_ones_windows", None, int("11" * 32, 16)),
            ("u1_eq_u2", 0x123456789ABCDEF, 0x123456789ABCDEF), ("u2_n_minus_16", None, o.N - 16),
            ("u2_eq_8_pattern", None, int("88" * 32, 16)), ("u2_eq_7_pattern", None, int("77" * 32, 16))]
for name, a, b in patterns:
    for attempt in range(64):
        aa = a if a is not None else rand_scalar()
        bb = b if b is not None else rand_scalar()
        d = rand_scalar()
        Q, R = from_d_ab(d, aa, bb)
        if R is None:
            continue
        r = R[0] % o.N
        s = r * pow(bb, -1, o.N) % o.N
        if r == 0 or s > o.HALF_N:
            continue
        e = aa * s % o.N
        assert emit(name, Q, b32(e), r, s, "u1=%x u2=%x" % (aa, bb)) == o.ST_VALID
        break
    else:
        raise SystemExit("no low-S instance for " + name)

# u1 == 0 (e == 0 mod n): R = u2 Q
for name, dg in (("u1_zero_digest0", bytes(32)), ("u1_zero_digest_n", b32(o.N))):
    Q, r, s = signed_case(name, dg)
    assert emit(name + "_valid", Q, dg, r, s) == o.ST_VALID
    assert emit(name + "_bad", Q, dg, r, (s + 1) if s + 1 <= o.HALF_N else s - 1) == o.ST_BAD_MATH

# 6. exceptional group-law cases in the final addition u1 G + u2 Q --------------------------
for i in range(3):
    while True:
        a, b = rand_scalar(), rand_scalar()
        d = a * pow(b, -1, o.N) % o.N            # u2 Q == u1 G  -> doubling
        Q, R = from_d_ab(d, a, b)
        r = R[0] % o.N
        s = r * pow(b, -1, o.N) % o.N
        if s <= o.HALF_N:
            break
    assert emit("final_add_is_doubling_%d" % i, Q, b32(a * s % o.N), r, s, "u1 G == u2 Q") == o.ST_VALID
    assert emit("final_add_is_doubling_wrong_r_%d" % i, Q, b32(a * s % o.N), r ^ 1 or 2, s) in (o.ST_BAD_MATH,)
for i in range(3):
    while True:
        a, b = rand_scalar(), rand_scalar()
        d = (-a) * pow(b, -1, o.N) % o.N         # u2 Q == -u1 G -> infinity
        Q, R = from_d_ab(d, a, b)
        assert R is None
        r = rand_scalar()
        s = r * pow(b, -1, o.N) % o.N
        if s <= o.HALF_N:
            break
    assert emit("final_add_is_infinity_%d" % i, Q, b32(a * s % o.N), r, s, "u1 G == -u2 Q") == o.ST_BAD_MATH
# Q = G, Q = -G, Q = 2G with ordinary scalars
for name, d in (("Q_eq_G", 1), ("Q_eq_negG", o.N - 1), ("Q_eq_2G", 2), ("Q_eq_nm2_G", o.N - 2)):
    dg = rng.randbytes(32)
    Q, r, s = signed_case(name, dg, d=d)
    assert emit(name, Q, dg, r, s) == o.ST_VALID

# 7. x(R) >= n : r = x - n must be accepted; and small / extreme r values -------------------
cnt = 0
x0 = 0
while cnt < 3:
    R = lift_x(o.N + x0)
    x0 += 1
    if R is None or (o.N + x0 - 1) >= o.P:
        continue
    while True:
        Q, e, r, s = from_R_ab(R, rand_scalar(), rand_scalar())
        if s <= o.HALF_N and r != 0:
            break
    assert r == R[0] - o.N
    assert emit("x_ge_n_%d" % cnt, Q, b32(e), r, s, "x(R) = n + %d" % r) == o.ST_VALID
    # same r but a point whose x really is r (if it exists) is a different signature: not added
    cnt += 1
R = lift_x(o.P - 1) or lift_x(o.P - 2) or lift_x(o.P - 3)
while True:
    Q, e, r, s = from_R_ab(R, rand_scalar(), rand_scalar())
    if s <= o.HALF_N:
        break
assert emit("x_near_p", Q, b32(e), r, s, "x(R) = p - small, r = x - n") == o.ST_VALID
for name, xs in (("r_tiny", range(1, 40)), ("r_near_n", range(o.N - 1, o.N - 40, -1))):
    for x in xs:
        R = lift_x(x)
        if R is None:
            continue
        while True:
            Q, e, r, s = from_R_ab(R, rand_scalar(), rand_scalar())
            if s <= o.HALF_N:
                break
        assert emit("%s_%x" % (name, x if x < 100 else o.N - x), Q, b32(e), r, s, "x(R) = r exactly") == o.ST_VALID
        # r + n < p does NOT hold for r near n; for tiny r the twin x = r + n is a different point:
        break
# r small where the candidate x = r + n exists but the real x(R) is r + n and the signature says r: covered by x_ge_n.
# Negative twin: take an x_ge_n vector and claim r' = r but with R' = point of x = r (if on curve) -> already "valid" case above.

# 8. off-curve public keys -------------------------------------------------------------------
dg = rng.randbytes(32)
Q, r, s = signed_case("offcurve", dg)
assert emit("off_curve_y_plus_1", (Q[0], (Q[1] + 1) % o.P), dg, r, s) == o.ST_OFF_CURVE
assert emit("off_curve_zero_point", (0, 0), dg, r, s) == o.ST_OFF_CURVE
assert emit("off_curve_x_ge_p", (o.P, Q[1]), dg, r, s) == o.ST_OFF_CURVE
assert emit("off_curve_y_ge_p", (Q[0], Q[1] + o.P) if Q[1] + o.P < 1 << 256 else (Q[0], o.P), dg, r, s) == o.ST_OFF_CURVE
assert emit("off_curve_xy_swapped", (Q[1], Q[0]), dg, r, s) == o.ST_OFF_CURVE
# precedence: sign/range and low-S gates fire before the curve check (reference order of checks)
assert emit("off_curve_but_high_s", (Q[0], (Q[1] + 1) % o.P), dg, r, o.N - s) == o.ST_HIGH_S
assert emit("off_curve_but_r_zero", (Q[0], (Q[1] + 1) % o.P), dg, 0, s) == o.ST_RANGE

json.dump(dict(generator="tests/golden/gen_edge_kats.py", seed=20260921, vectors=vectors),
          open(os.path.join(HERE, "edge_kats.json"), "w"), indent=0)
print("edge vectors:", len(vectors))

# ---------------------------------------------------------------------------------------------
# DER vectors
# ---------------------------------------------------------------------------------------------
der = []


def emit_der(name, raw, note=""):
    try:
        r, s = o.unmarshal_ecdsa_signature(raw)
        res = dict(ok=True, r="%x" % r, s="%x" % s)
    except o.BCCSPError as e:
        res = dict(ok=False, err=str(e))
    der.append(dict(name=name, der=raw.hex(), note=note, **res))
    return res


# literal vectors of bccsp/sw/impl_test.go:931-964 (all must fail to unmarshal)
for i, hx in enumerate(("3007020" "18f0202fff1", "3007020" "18f02020001", "3007020" "18f02810101",
                        "3007020" "18f0281018f", "300a02018f0205000000008f")):
    assert not emit_der("ref_impl_test_negative_%d" % i, bytes.fromhex(hx), "bccsp/sw/impl_test.go:931-964")["ok"]
# bccsp/utils/ecdsa_test.go:20-30: nil / empty / 1-byte signatures
assert not emit_der("empty", b"")["ok"]
assert not emit_der("one_zero_byte", b"\x00")["ok"]
# bccsp/utils/ecdsa_test.go:32-54: R,S in {-1,0}
for name, rr, ss in (("r_minus_1", -1, 1), ("r_zero", 0, 1), ("s_minus_1", 1, -1), ("s_zero", 1, 0)):
    res = emit_der(name, o.marshal_ecdsa_signature(rr, ss), "bccsp/utils/ecdsa_test.go:32-54")
    assert not res["ok"] and "must be larger than zero" in res["err"]
good = o.marshal_ecdsa_signature(*signed_case("der", dg)[1:])
assert emit_der("good", good)["ok"]
assert emit_der("good_trailing_garbage_after_sequence", good + b"\x00\x01\x02", "asn1.Unmarshal rest is discarded")["ok"]
r_, s_ = o.unmarshal_ecdsa_signature(good)
body = good[2:]
assert emit_der("good_extra_element_inside_sequence", b"\x30" + bytes([len(body) + 3]) + body + b"\x02\x01\x05",
                "Go allows trailing SEQUENCE content")["ok"]
assert not emit_der("truncated_by_one", good[:-1])["ok"]
assert not emit_der("sequence_length_too_long", b"\x30" + bytes([len(body) + 1]) + body)["ok"]
assert not emit_der("missing_s", b"\x30" + bytes([len(body) - (2 + body[body[1] + 3])]) + body[:2 + body[1]])["ok"]
assert not emit_der("wrong_outer_tag_31", b"\x31" + good[1:])["ok"]
assert not emit_der("outer_tag_primitive_10", b"\x10" + good[1:])["ok"]
assert not emit_der("int_tag_constructed_22", good[:2] + b"\x22" + good[3:])["ok"]
assert not emit_der("long_form_tag", b"\x3f\x10" + good[1:])["ok"]
assert not emit_der("indefinite_length", b"\x30\x80" + body + b"\x00\x00")["ok"]
assert not emit_der("long_form_short_length", b"\x30\x81" + good[1:2] + body)["ok"]
big = o.marshal_ecdsa_signature(1 << 1100, 1)
assert emit_der("long_form_length_valid_big_r", big, "r far above 2^256 parses; verify then rejects on range")["ok"]
assert emit_der("big_s", o.marshal_ecdsa_signature(1, 1 << 300))["ok"]
assert not emit_der("leading_zero_padding_r", b"\x30\x08\x02\x03\x00\x00\x01\x02\x01\x01")["ok"]
assert emit_der("needed_leading_zero", b"\x30\x07\x02\x02\x00\x80\x02\x01\x01")["ok"]
assert not emit_der("negative_r_0x80", b"\x30\x06\x02\x01\x80\x02\x01\x01")["ok"]
assert not emit_der("ff_padding", b"\x30\x07\x02\x02\xff\x80\x02\x01\x01")["ok"]
assert not emit_der("empty_integer", b"\x30\x05\x02\x00\x02\x01\x01")["ok"]
assert not emit_der("empty_sequence", b"\x30\x00")["ok"]
assert not emit_der("only_tag", b"\x30")["ok"]
assert not emit_der("length_leading_zero", b"\x30\x82\x00\x08" + b"\x02\x01\x01\x02\x01\x01\x00\x00")["ok"]
assert emit_der("r1_s1", b"\x30\x06\x02\x01\x01\x02\x01\x01")["ok"]
json.dump(dict(generator="tests/golden/gen_edge_kats.py", vectors=der),
          open(os.path.join(HERE, "der_kats.json"), "w"), indent=0)
print("der vectors:", len(der))
