"""CPU oracle for the Idemix pseudonym-signature verification path (BASELINE config 5).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench tooling as the checker, never by the
product path.

What it restates (reference file:line):
  * idemix/nymsignature.go:74-109   NymSignature.Ver      -> nym_verify()
  * idemix/nymsignature.go:25-71    NewNymSignature       -> nym_sign()      (to make test inputs)
  * idemix/util.go:46-51            HashModOrder          -> hash_mod_order()
  * idemix/util.go:57-61,79-83      appendBytesG1 / appendBytesBig (ECP.ToBytes uncompressed: 0x04 || x || y, 32-byte big-endian)
  * idemix/util.go:128-131          EcpFromProto -> FP256BN.NewECPbigs
  * idemix/issuerkey.go:114-172     IssuerPublicKey.Check -> ipk_check()     (pins G1/G2 arithmetic, serialisation and
                                                                              HashModOrder on the reference's fixtures)
  * idemix/credential.go:110-146    the B-value recomputation of Credential.Ver -> credential_b_check()  (pins G1
                                                                              multi-scalar multiplication on the fixtures)
  * bccsp/idemix/bridge/nymsignaturescheme.go:66-89, bccsp/idemix/handlers/nymsigner.go:62-95  (the BCCSP wrapper:
    proto.Unmarshal of the signature, recover() of panics into errors)

The curve arithmetic itself lives in a third-party dependency that is NOT in /root/reference:
  github.com/hyperledger/fabric-amcl v0.0.0-20200128223036-d1aa2665426a (go.mod:44), package amcl/FP256BN.
It is restated here from the published definition of the curve: BN curve "FP256BN" (ISO/IEC 15946-5), u = -0x6882F5C030B0A801,
p = 36u^4+36u^3+24u^2+6u+1, r = 36u^4+36u^3+18u^2+6u+1, E: y^2 = x^3 + 3, G1 generator (1, 2); Fp2 = Fp[i]/(i^2+1); the
sextic twist E': y^2 = x^3 + 3/(1+i)... -- the twist constant and the G2 generator are NOT taken from memory: they are derived
from / checked against the reference's fixtures (see pin_* below and tests/test_idemix_oracle.py).

Pinning status: the reference holds no byte-level known-answer test for NymSignature.Ver.  What IS pinned against reference
fixtures (msp/testdata/idemix/*): curve constants (every fixture point is on the curve), G1 multi-scalar multiplication
(credential B-values), G1/G2 serialisation + HashModOrder + the Schnorr-proof structure t = g^s * h^-c (IssuerPublicKey.Check on the
two independent issuer keys of the five fixture directories).  Nym signatures themselves are produced by nym_sign() below -> "parity pinned at the
arithmetic / hashing layer, unpinned at the NymSignature byte level"; corner cases that depend on amcl internals (points at
infinity, off-curve or unreduced inputs, short fields) are therefore NOT answered by the device path: it reports NEEDS_SW.
"""
import hashlib

U = -0x6882F5C030B0A801
P = 36 * U**4 + 36 * U**3 + 24 * U**2 + 6 * U + 1
R = 36 * U**4 + 36 * U**3 + 18 * U**2 + 6 * U + 1
B = 3
G1 = (1, 2)
FIELD_BYTES = 32
SIGN_LABEL = b"sign"                      # idemix/signature.go:19

assert P == 0xFFFFFFFFFFFCF0CD46E5F25EEE71A49F0CDC65FB12980A82D3292DDBAED33013
assert R == 0xFFFFFFFFFFFCF0CD46E5F25EEE71A49E0CDC65FB1299921AF62D536CD10B500D

# status codes shared with include/fabgpu.h (FABGPU_NYM_*)
NYM_VALID = 0
NYM_BAD_PROOF = 1          # "pseudonym signature invalid: zero-knowledge proof is invalid"
NYM_NEEDS_SW = 6           # outside the pinned domain: the caller must ask bccsp/sw (same code as TUPLE_ST_NEEDS_SW)


# ---------------------------------------------------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fp.  None = point at infinity.
# ---------------------------------------------------------------------------------------------------------------------
def g1_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B) % P == 0


def g1_neg(pt):
    return None if pt is None else (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def g1_mul(pt, k):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, pt)
        pt = g1_add(pt, pt)
        k >>= 1
    return acc


def g1_mul2(a, ka, b, kb):
    """ECP.Mul2: ka*a + kb*b"""
    return g1_add(g1_mul(a, ka), g1_mul(b, kb))


def ecp_to_bytes(pt):
    """FP256BN.ECP.ToBytes(b, false) for a finite point (idemix/util.go:57-61)"""
    assert pt is not None
    return b"\x04" + pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def big_to_bytes(x):
    return x.to_bytes(32, "big")


def hash_mod_order(data):
    """idemix/util.go:46-51"""
    return int.from_bytes(hashlib.sha256(data).digest(), "big") % R


# ---------------------------------------------------------------------------------------------------------------------
# Fp2 = Fp[i]/(i^2 + 1) and G2 on the sextic twist y^2 = x^3 + B2 (B2 derived from the fixtures: pin_twist()).
# ---------------------------------------------------------------------------------------------------------------------
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


XI = (1, 1)                                     # 1 + i
TWIST_B_CANDIDATES = {
    "D-type b/xi": f2_scalar(f2_inv(XI), B),
    "M-type b*xi": f2_scalar(XI, B),
}


def g2_rhs_minus_lhs(pt):
    x, y = pt
    return f2_sub(f2_mul(y, y), f2_mul(f2_mul(x, x), x))


def pin_twist(points):
    """Which twist constant do the reference's W values satisfy?  Returns (name, B2)."""
    for name, b2 in TWIST_B_CANDIDATES.items():
        if all(g2_rhs_minus_lhs(pt) == b2 for pt in points):
            return name, b2
    raise ValueError("fixture G2 points satisfy neither twist equation")


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_scalar(f2_mul(x1, x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_neg(a):
    return None if a is None else (a[0], ((-a[1][0]) % P, (-a[1][1]) % P))


def g2_mul(pt, k):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, pt)
        pt = g2_add(pt, pt)
        k >>= 1
    return acc


def ecp2_to_bytes(pt):
    """FP256BN.ECP2.ToBytes: xa || xb || ya || yb (idemix/util.go:69-73, Ecp2ToProto :141-147)"""
    (xa, xb), (ya, yb) = pt
    return b"".join(v.to_bytes(32, "big") for v in (xa, xb, ya, yb))


# ---------------------------------------------------------------------------------------------------------------------
# protobuf (idemix/idemix.proto; msp.IdemixMSPSignerConfig of fabric-protos-go msp/msp_config.proto)
# ---------------------------------------------------------------------------------------------------------------------
def _varint(b, i):
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 0x7F) << s
        s += 7
        if c < 0x80:
            return v, i


def pb_fields(b):
    i, out = 0, []
    while i < len(b):
        k, i = _varint(b, i)
        f, w = k >> 3, k & 7
        if w == 2:
            ln, i = _varint(b, i)
            if i + ln > len(b):
                raise ValueError("truncated")
            out.append((f, bytes(b[i:i + ln])))
            i += ln
        elif w == 0:
            v, i = _varint(b, i)
            out.append((f, v))
        elif w == 5:
            out.append((f, bytes(b[i:i + 4])))
            i += 4
        elif w == 1:
            out.append((f, bytes(b[i:i + 8])))
            i += 8
        else:
            raise ValueError("wire type %d" % w)
    return out


def pb_bytes_field(f, v):
    out = bytearray()
    k = (f << 3) | 2
    out.append(k)
    n = len(v)
    while True:
        c = n & 0x7F
        n >>= 7
        out.append(c | (0x80 if n else 0))
        if not n:
            break
    return bytes(out) + bytes(v)


def _ecp(b):
    d = dict(pb_fields(b))
    return (int.from_bytes(d.get(1, b""), "big"), int.from_bytes(d.get(2, b""), "big"))


def _ecp2(b):
    d = dict(pb_fields(b))
    g = lambda k: int.from_bytes(d.get(k, b""), "big")
    return ((g(1), g(2)), (g(3), g(4)))


class IssuerPublicKey:
    """idemix.proto IssuerPublicKey"""

    def __init__(self, raw):
        self.attribute_names, self.h_attrs = [], []
        self.h_sk = self.h_rand = self.w = self.bar_g1 = self.bar_g2 = None
        self.proof_c = self.proof_s = self.hash = b""
        for f, v in pb_fields(raw):
            if f == 1:
                self.attribute_names.append(v)
            elif f == 2:
                self.h_sk = _ecp(v)
            elif f == 3:
                self.h_rand = _ecp(v)
            elif f == 4:
                self.h_attrs.append(_ecp(v))
            elif f == 5:
                self.w = _ecp2(v)
            elif f == 6:
                self.bar_g1 = _ecp(v)
            elif f == 7:
                self.bar_g2 = _ecp(v)
            elif f == 8:
                self.proof_c = v
            elif f == 9:
                self.proof_s = v
            elif f == 10:
                self.hash = v


class Credential:
    def __init__(self, raw):
        self.attrs = []
        for f, v in pb_fields(raw):
            if f == 1:
                self.a = _ecp(v)
            elif f == 2:
                self.b = _ecp(v)
            elif f == 3:
                self.e = int.from_bytes(v, "big")
            elif f == 4:
                self.s = int.from_bytes(v, "big")
            elif f == 5:
                self.attrs.append(int.from_bytes(v, "big"))


class SignerConfig:
    """msp.IdemixMSPSignerConfig: cred = 1, sk = 2, organizational_unit_identifier = 3, role = 4, enrollment_id = 5, ..."""

    def __init__(self, raw):
        d = {}
        for f, v in pb_fields(raw):
            d[f] = v
        self.cred = Credential(d[1])
        self.sk = int.from_bytes(d[2], "big")


# ---------------------------------------------------------------------------------------------------------------------
# fixture-backed checks (idemix/credential.go:110-146, idemix/issuerkey.go:114-172)
# ---------------------------------------------------------------------------------------------------------------------
def credential_b_check(cred, sk, ipk):
    bp = G1
    bp = g1_add(bp, g1_mul2(ipk.h_sk, sk, ipk.h_rand, cred.s))
    n = len(cred.attrs)
    for i in range(n // 2):
        bp = g1_add(bp, g1_mul2(ipk.h_attrs[2 * i], cred.attrs[2 * i], ipk.h_attrs[2 * i + 1], cred.attrs[2 * i + 1]))
    if n % 2:
        bp = g1_add(bp, g1_mul(ipk.h_attrs[n - 1], cred.attrs[n - 1]))
    return bp == cred.b


def derive_gen_g2(w, isk):
    """W = GenG2^isk (idemix/issuerkey.go:63) => GenG2 = W^(1/isk): the generator constant of amcl's ROM, recovered from a fixture."""
    return g2_mul(w, pow(isk, -1, R))


def ipk_check(ipk, gen_g2):
    """idemix/issuerkey.go:146-169"""
    c = int.from_bytes(ipk.proof_c, "big")
    s = int.from_bytes(ipk.proof_s, "big")
    t1 = g2_add(g2_mul(gen_g2, s), g2_mul(ipk.w, (-c) % R))
    t2 = g1_add(g1_mul(ipk.bar_g1, s), g1_mul(ipk.bar_g2, (-c) % R))
    data = ecp2_to_bytes(t1) + ecp_to_bytes(t2) + ecp2_to_bytes(gen_g2) + ecp_to_bytes(ipk.bar_g1) + ecp2_to_bytes(ipk.w) + ecp_to_bytes(ipk.bar_g2)
    assert len(data) == 18 * FIELD_BYTES + 3
    return c == hash_mod_order(data)


# ---------------------------------------------------------------------------------------------------------------------
# the path: NymSignature sign / verify
# ---------------------------------------------------------------------------------------------------------------------
def make_nym(sk, ipk, rng):
    """idemix/util.go:100-107"""
    r_nym = rng.randrange(R)
    return g1_mul2(ipk.h_sk, sk, ipk.h_rand, r_nym), r_nym


def nym_sign(sk, nym, r_nym, ipk, msg, rng):
    """idemix/nymsignature.go:25-71 -> dict of the four 32-byte fields"""
    nonce = rng.randrange(R)
    r_sk, r_rnym = rng.randrange(R), rng.randrange(R)
    t = g1_mul2(ipk.h_sk, r_sk, ipk.h_rand, r_rnym)
    c = hash_mod_order(SIGN_LABEL + ecp_to_bytes(t) + ecp_to_bytes(nym) + ipk.hash + msg)
    proof_c = hash_mod_order(big_to_bytes(c) + big_to_bytes(nonce))
    return {
        "proof_c": big_to_bytes(proof_c),
        "proof_s_sk": big_to_bytes((r_sk + proof_c * sk) % R),
        "proof_s_r_nym": big_to_bytes((r_rnym + proof_c * r_nym) % R),
        "nonce": big_to_bytes(nonce),
    }


def nym_signature_marshal(sig):
    """idemix.proto NymSignature: proof_c = 1, proof_s_sk = 2, proof_s_r_nym = 3, nonce = 4"""
    return (pb_bytes_field(1, sig["proof_c"]) + pb_bytes_field(2, sig["proof_s_sk"]) + pb_bytes_field(3, sig["proof_s_r_nym"])
            + pb_bytes_field(4, sig["nonce"]))


def nym_signature_unmarshal(raw):
    sig = {"proof_c": b"", "proof_s_sk": b"", "proof_s_r_nym": b"", "nonce": b""}
    names = {1: "proof_c", 2: "proof_s_sk", 3: "proof_s_r_nym", 4: "nonce"}
    for f, v in pb_fields(raw):
        if f in names and isinstance(v, bytes):
            sig[names[f]] = v
    return sig


def in_pinned_domain(sig, nym_xy, ipk_hash):
    """The inputs for which this restatement stands on pinned ground (see module docstring): all four fields exactly 32 bytes
    (FP256BN.FromBytes reads MODBYTES bytes: shorter panics, longer is truncated), nym coordinates < p and on the curve
    (NewECPbigs turns anything else into the point at infinity), s-values below the group order, ipk.Hash 32 bytes.
    (ProofC >= r needs no gate: the comparison of BIGs at :104 can never succeed for it - nym_verify answers BAD_PROOF.)"""
    if any(len(sig[k]) != 32 for k in ("proof_c", "proof_s_sk", "proof_s_r_nym", "nonce")):
        return False
    if len(ipk_hash) != 32:
        return False
    x, y = nym_xy
    if x >= P or y >= P or not g1_on_curve((x, y)):
        return False
    # s-values at or above the group order: what ECP.Mul2 does with an unreduced scalar is amcl-internal (honest signers
    # always emit reduced values, idemix/nymsignature.go:61-62)
    if int.from_bytes(sig["proof_s_sk"], "big") >= R or int.from_bytes(sig["proof_s_r_nym"], "big") >= R:
        return False
    return True


def nym_verify(sig, nym_xy, ipk, msg):
    """idemix/nymsignature.go:74-109.  Returns NYM_VALID / NYM_BAD_PROOF, or NYM_NEEDS_SW outside the pinned domain."""
    if not in_pinned_domain(sig, nym_xy, ipk.hash):
        return NYM_NEEDS_SW
    proof_c = int.from_bytes(sig["proof_c"], "big")
    s_sk = int.from_bytes(sig["proof_s_sk"], "big")
    s_rnym = int.from_bytes(sig["proof_s_r_nym"], "big")
    nonce = sig["nonce"]
    t = g1_mul2(ipk.h_sk, s_sk, ipk.h_rand, s_rnym)
    t = g1_add(t, g1_neg(g1_mul(nym_xy, proof_c)))
    if t is None:
        return NYM_NEEDS_SW               # ToBytes of the point at infinity: amcl-internal
    c = hash_mod_order(SIGN_LABEL + ecp_to_bytes(t) + ecp_to_bytes(nym_xy) + ipk.hash + msg)
    # *ProofC != *HashModOrder(...) compares BIGs: an unreduced ProofC (>= r) can never match
    return NYM_VALID if proof_c == hash_mod_order(big_to_bytes(c) + nonce) else NYM_BAD_PROOF


def nym_verify_t(sig, nym_xy, ipk):
    """the intermediate commitment t (affine) - exposed so that tests can compare the device's point arithmetic on its own"""
    t = g1_mul2(ipk.h_sk, int.from_bytes(sig["proof_s_sk"], "big"), ipk.h_rand, int.from_bytes(sig["proof_s_r_nym"], "big"))
    return g1_add(t, g1_neg(g1_mul(nym_xy, int.from_bytes(sig["proof_c"], "big"))))
