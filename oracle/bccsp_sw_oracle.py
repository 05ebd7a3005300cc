"""CPU ORACLE (test infrastructure, NOT product code).

Pure-Python restatement of the accept/reject procedure of the reference's software
crypto provider for the block-validation signature path:

    msp/identities.go:169-196        identity.Verify  = Hash(msg) then Verify(pk, sig, digest)
    bccsp/sw/impl.go:177-194         CSP.Hash   argument checks
    bccsp/sw/hash.go:29-33           hasher.Hash = sha256
    bccsp/sw/impl.go:247-270         CSP.Verify argument checks + error wrapping
    bccsp/sw/ecdsa.go:41-57          verifyECDSA: DER -> (r,s), low-S gate, ecdsa.Verify
    bccsp/utils/ecdsa.go:43-67       UnmarshalECDSASignature
    bccsp/utils/ecdsa.go:84-92,27-33 IsLowS / curveHalfOrders

The arithmetic itself lives in a third-party dependency that is NOT under /root/reference:
the Go standard library 1.14.4 (reference Makefile:79, go.mod:3) -- crypto/ecdsa.Verify,
crypto/elliptic P-256, crypto/sha256, encoding/asn1.  Its published algorithm is restated
here: FIPS 186-4 ECDSA verification with Go's hashToInt truncation,
no on-curve check inside Verify, group-law handling of exceptional points, and Go's strict
DER rules for SEQUENCE{INTEGER,INTEGER} with trailing bytes tolerated.

Pinning: tests/test_oracle_golden.py checks this file against (a) the ECDSA-P256 signatures
of every X.509 fixture certificate in the reference tree (tests/golden/ref_cert_kats.json,
made by tests/golden/gen_ref_cert_kats.py), (b) the literal DER byte vectors of
bccsp/sw/impl_test.go:931-964, (c) the low-S boundary cases of bccsp/utils/ecdsa_test.go:64-88,
and (d) OpenSSL 3.0 libcrypto as an independent implementation.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import hashlib
from typing import Optional, Tuple

# --- P-256 domain parameters (FIPS 186-4 D.1.2.3) ------------------
P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF
A = P - 3
B = 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B
N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551
GX = 0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296
GY = 0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5
HALF_N = N >> 1  # bccsp/utils/ecdsa.go:27-33 curveHalfOrders[P256] = N >> 1

# --- verdict / status codes shared with include/fabgpu.h ---------------------------------
ST_VALID = 0        # (true,  nil)
ST_BAD_MATH = 1     # (false, nil)   arithmetic reject, incl. point at infinity
ST_HIGH_S = 2       # (false, err)   "Invalid S. Must be smaller than half the order"
ST_RANGE = 3        # r >= n (false,nil in Go) or r,s == 0 (false, err upstream)
ST_OFF_CURVE = 4    # public key not on the curve -> host must use the CPU provider


class BCCSPError(Exception):
    """Mirrors a non-nil Go `error` return."""


# ---------------------------------------------------------------------------------------
# encoding/asn1 restatement (Go 1.14 asn1.go parseTagAndLength/parseField/parseBigInt)
# ---------------------------------------------------------------------------------------
class ASN1Error(Exception):
    pass


def _parse_tag_and_length(buf: bytes, off: int) -> Tuple[int, int, bool, int, int]:
    """Returns (class, tag, is_compound, length, new_off). Go asn1.go parseTagAndLength."""
    if off >= len(buf):
        raise ASN1Error("asn1: internal error in parseTagAndLength")
    b = buf[off]
    off += 1
    cls, compound, tag = b >> 6, bool(b & 0x20), b & 0x1F
    if tag == 0x1F:
        # base-128 tag
        tag = 0
        shifted = 0
        while True:
            if off >= len(buf):
                raise ASN1Error("truncated base 128 integer")
            if shifted == 5:
                raise ASN1Error("base 128 integer too large")
            c = buf[off]
            off += 1
            if shifted == 0 and c == 0x80:
                raise ASN1Error("integer is not minimally encoded")
            tag = (tag << 7) | (c & 0x7F)
            shifted += 1
            if not c & 0x80:
                break
        if tag < 0x1F:
            raise ASN1Error("non-minimal tag")
    if off >= len(buf):
        raise ASN1Error("truncated tag or length")
    b = buf[off]
    off += 1
    if not b & 0x80:
        length = b & 0x7F
    else:
        nbytes = b & 0x7F
        if nbytes == 0:
            raise ASN1Error("indefinite length found (not DER)")
        length = 0
        for _ in range(nbytes):
            if off >= len(buf):
                raise ASN1Error("truncated tag or length")
            c = buf[off]
            off += 1
            if length >= 1 << 23:
                raise ASN1Error("length too large")
            length = (length << 8) | c
            if length == 0:
                raise ASN1Error("superfluous leading zeros in length")
        if length < 0x80:
            raise ASN1Error("non-minimal length")
    return cls, tag, compound, length, off


def _parse_bigint_field(buf: bytes, off: int) -> Tuple[int, int]:
    """One `*big.Int` struct field: universal, primitive, tag 2, minimally encoded."""
    if off == len(buf):
        raise ASN1Error("sequence truncated")
    cls, tag, compound, length, off = _parse_tag_and_length(buf, off)
    if cls != 0 or tag != 2 or compound:
        raise ASN1Error("tags don't match")
    if off + length > len(buf):
        raise ASN1Error("data truncated")
    body = buf[off:off + length]
    if len(body) == 0:
        raise ASN1Error("empty integer")
    if len(body) > 1 and ((body[0] == 0 and not body[1] & 0x80) or
                          (body[0] == 0xFF and body[1] & 0x80)):
        raise ASN1Error("integer not minimally-encoded")
    return int.from_bytes(body, "big", signed=True), off + length


def asn1_unmarshal_ecdsa_sig(raw: bytes) -> Tuple[int, int]:
    """asn1.Unmarshal(raw, &ECDSASignature{R,S *big.Int}); bccsp/utils/ecdsa.go:46.
    Bytes after the SEQUENCE and extra elements inside it are tolerated (Go discards `rest`
    and allows trailing SEQUENCE content)."""
    if len(raw) == 0:
        raise ASN1Error("sequence truncated")
    cls, tag, compound, length, off = _parse_tag_and_length(raw, 0)
    if cls != 0 or tag != 16 or not compound:
        raise ASN1Error("tags don't match")
    if off + length > len(raw):
        raise ASN1Error("data truncated")
    inner = raw[off:off + length]
    r, ioff = _parse_bigint_field(inner, 0)
    s, ioff = _parse_bigint_field(inner, ioff)
    return r, s


def unmarshal_ecdsa_signature(raw: bytes) -> Tuple[int, int]:
    """bccsp/utils/ecdsa.go:43-67."""
    try:
        r, s = asn1_unmarshal_ecdsa_sig(raw)
    except ASN1Error as e:
        raise BCCSPError("failed unmashalling signature [%s]" % e)
    if r <= 0:
        raise BCCSPError("invalid signature, R must be larger than zero")
    if s <= 0:
        raise BCCSPError("invalid signature, S must be larger than zero")
    return r, s


def marshal_ecdsa_signature(r: int, s: int) -> bytes:
    """asn1.Marshal(ECDSASignature{r,s}); bccsp/utils/ecdsa.go:39-41 (minimal DER)."""
    def enc_int(v: int) -> bytes:
        body = v.to_bytes(max(1, (v.bit_length() + 8) // 8), "big", signed=True) if v >= 0 else \
            v.to_bytes((v.bit_length() + 8) // 8 or 1, "big", signed=True)
        # strip redundant leading bytes
        while len(body) > 1 and ((body[0] == 0 and not body[1] & 0x80) or
                                 (body[0] == 0xFF and body[1] & 0x80)):
            body = body[1:]
        return b"\x02" + _enc_len(len(body)) + body

    def _enc_len(n: int) -> bytes:
        if n < 0x80:
            return bytes([n])
        b = n.to_bytes((n.bit_length() + 7) // 8, "big")
        return bytes([0x80 | len(b)]) + b

    body = enc_int(r) + enc_int(s)
    return b"\x30" + _enc_len(len(body)) + body


def is_low_s(s: int) -> bool:
    """bccsp/utils/ecdsa.go:84-92: s.Cmp(halfOrder) != 1."""
    return s <= HALF_N


# ---------------------------------------------------------------------------------------
# crypto/elliptic P-256 group law (affine, textbook; the obviously-correct version)
# ---------------------------------------------------------------------------------------
INF = None  # point at infinity


def on_curve(x: int, y: int) -> bool:
    if not (0 <= x < P and 0 <= y < P):
        return False
    return (y * y - (x * x * x + A * x + B)) % P == 0


def pt_add(p1, p2):
    if p1 is INF:
        return p2
    if p2 is INF:
        return p1
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = (3 * x1 * x1 + A) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def pt_mul(k: int, pt):
    acc = INF
    addend = pt
    while k:
        if k & 1:
            acc = pt_add(acc, addend)
        addend = pt_add(addend, addend)
        k >>= 1
    return acc


def hash_to_int(digest: bytes) -> int:
    """Go crypto/ecdsa hashToInt for P-256: leftmost 32 bytes, big-endian, no reduction."""
    if len(digest) > 32:
        digest = digest[:32]
    return int.from_bytes(digest, "big")


def ecdsa_verify_raw(qx: int, qy: int, digest: bytes, r: int, s: int) -> bool:
    """Go 1.14 crypto/ecdsa.Verify(pub, hash, r, s) on elliptic.P256().
    The public key is assumed on-curve (the boundary gates off-curve keys to the CPU provider)."""
    if r <= 0 or s <= 0:
        return False
    if r >= N or s >= N:
        return False
    e = hash_to_int(digest)
    w = pow(s, -1, N)
    u1 = e * w % N
    u2 = r * w % N
    pt = pt_add(pt_mul(u1, (GX, GY)), pt_mul(u2, (qx % P, qy % P)))
    if pt is INF:
        return False
    return pt[0] % N == r


# ---------------------------------------------------------------------------------------
# bccsp/sw restatement
# ---------------------------------------------------------------------------------------
def csp_hash(msg: Optional[bytes]) -> bytes:
    """CSP.Hash(msg, &bccsp.SHA256Opts{}) (bccsp/sw/impl.go:177-194 -> hash.go:29-33)."""
    return hashlib.sha256(msg or b"").digest()


def verify_ecdsa(qx: int, qy: int, signature: bytes, digest: bytes) -> bool:
    """bccsp/sw/ecdsa.go:41-57 verifyECDSA. Raises BCCSPError where Go returns (false, err)."""
    try:
        r, s = unmarshal_ecdsa_signature(signature)
    except BCCSPError as e:
        raise BCCSPError("Failed unmashalling signature [%s]" % e)
    if not is_low_s(s):
        raise BCCSPError("Invalid S. Must be smaller than half the order [%d][%d]." % (s, HALF_N))
    return ecdsa_verify_raw(qx, qy, digest, r, s)


def csp_verify(key: Optional[Tuple[int, int]], signature: bytes, digest: bytes) -> bool:
    """CSP.Verify (bccsp/sw/impl.go:247-270) for an ECDSA public key (X,Y)."""
    if key is None:
        raise BCCSPError("Invalid Key. It must not be nil.")
    if len(signature) == 0:
        raise BCCSPError("Invalid signature. Cannot be empty.")
    if len(digest) == 0:
        raise BCCSPError("Invalid digest. Cannot be empty.")
    try:
        return verify_ecdsa(key[0], key[1], signature, digest)
    except BCCSPError as e:
        raise BCCSPError("Failed verifing with opts [<nil>]: %s" % e)


def identity_verify(key: Tuple[int, int], msg: bytes, sig: bytes) -> Optional[str]:
    """msp/identities.go:169-196. Returns None (== nil error) or the error string."""
    digest = csp_hash(msg)
    try:
        ok = csp_verify(key, sig, digest)
    except BCCSPError as e:
        return "could not determine the validity of the signature: %s" % e
    return None if ok else "The signature is invalid"


def status_raw(qx: int, qy: int, digest: bytes, r: int, s: int) -> int:
    """Status code of the flattened-tuple boundary (include/fabgpu.h) for one (Q,e,r,s).
    Precedence mirrors the reference order of checks: r/s sign (utils/ecdsa.go:59-64),
    low-S (sw/ecdsa.go:47-54), range (ecdsa.Verify), then curve arithmetic; an off-curve key
    is reported before any arithmetic because the reference would have rejected it at import."""
    if r <= 0 or s <= 0:
        return ST_RANGE
    if not is_low_s(s):
        return ST_HIGH_S
    if r >= N:
        return ST_RANGE
    if not on_curve(qx, qy):
        return ST_OFF_CURVE
    return ST_VALID if ecdsa_verify_raw(qx, qy, digest, r, s) else ST_BAD_MATH


# ---------------------------------------------------------------------------------------
# signing helper (used only to build test vectors; bccsp/sw/ecdsa.go:27-39 signECDSA)
# ---------------------------------------------------------------------------------------
def sign_raw(d: int, digest: bytes, k: int, low_s: bool = True) -> Tuple[int, int]:
    e = hash_to_int(digest)
    R = pt_mul(k, (GX, GY))
    r = R[0] % N
    s = pow(k, -1, N) * (e + r * d) % N
    if low_s and s > HALF_N:
        s = N - s  # utils.ToLowS, bccsp/utils/ecdsa.go:94-109
    return r, s
