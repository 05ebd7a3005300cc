#!/usr/bin/env python3
"""Pseudonym-signature known-answer vectors from a SECOND, independent implementation (test infrastructure).

    python3 tests/golden/gen_idemix_nym_kats.py   ->   tests/golden/idemix_nym_kats.json

The reference tree stores NO pseudonym signature: idemix/idemix_test.go:155-161 and bccsp/idemix/bridge/bridge_test.go only sign and
verify with fresh randomness; msp/testdata/idemix/* holds key material (IssuerPublicKey, SignerConfig), not signatures.  So the
NymSignature bytes cannot be pinned to reference output.  What CAN be done is to stop the oracle (oracle/idemix_oracle.py) from being its
own witness: this script re-derives everything the verification equation needs WITHOUT importing the oracle -

    * its own protobuf field reader for the two fixture messages it needs (IssuerPublicKey.h_sk / h_rand / hash, SignerConfig.sk);
    * its own G1 arithmetic: affine chord-and-tangent on y^2 = x^3 + 3 over the FP256BN prime, binary double-and-add, inverses by
      Fermat (pow(x, p - 2, p)) - no shared code, no shared formulas (the oracle uses its own g1_add / g1_mul);
    * its own HashModOrder and ECP.ToBytes from idemix/util.go:46-61,

and then plays BOTH roles of idemix/nymsignature.go with fixed randomness: the prover's commitment t = HSk^r_sk HRand^r_rnym (:40-46)
and the verifier's t' = HSk^s_sk HRand^s_rnym Nym^(-c) (:84-92).  A vector is emitted only if t == t' (Schnorr completeness computed by
this implementation alone).  The stored t is then what oracle.nym_verify_t, the kernel headers on the host and the device must each
reproduce, and the stored signature is one the oracle and the device must accept - agreement of three implementations written against
the same published algorithm, two of them sharing nothing.
"""
import hashlib
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))

# FP256BN (fabric-amcl, go.mod:44): u = -0x6882F5C030B0A801
_u = -0x6882F5C030B0A801
PRIME = 36 * _u ** 4 + 36 * _u ** 3 + 24 * _u ** 2 + 6 * _u + 1
ORDER = 36 * _u ** 4 + 36 * _u ** 3 + 18 * _u ** 2 + 6 * _u + 1


def rd_varint(b, i):
    v = s = 0
    while True:
        c = b[i]
        i += 1
        v |= (c & 127) << s
        s += 7
        if c < 128:
            return v, i


def rd_fields(b):
    i, out = 0, {}
    while i < len(b):
        k, i = rd_varint(b, i)
        if k & 7 == 2:
            n, i = rd_varint(b, i)
            out.setdefault(k >> 3, []).append(b[i:i + n])
            i += n
        elif k & 7 == 0:
            v, i = rd_varint(b, i)
            out.setdefault(k >> 3, []).append(v)
        else:
            raise ValueError("unexpected wire type")
    return out


def rd_point(b):
    f = rd_fields(b)
    return int.from_bytes(f[1][0], "big"), int.from_bytes(f[2][0], "big")


# ---- affine arithmetic, written from the textbook group law --------------------------------------------------------------
def pt_double(a):
    if a is None or a[1] == 0:
        return None
    lam = 3 * a[0] * a[0] * pow(2 * a[1], PRIME - 2, PRIME) % PRIME
    x = (lam * lam - 2 * a[0]) % PRIME
    return x, (lam * (a[0] - x) - a[1]) % PRIME


def pt_plus(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        return pt_double(a) if a[1] == b[1] else None
    lam = (b[1] - a[1]) * pow(b[0] - a[0], PRIME - 2, PRIME) % PRIME
    x = (lam * lam - a[0] - b[0]) % PRIME
    return x, (lam * (a[0] - x) - a[1]) % PRIME


def pt_times(a, k):
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = pt_double(acc)
        if bit == "1":
            acc = pt_plus(acc, a)
    return acc


def pt_minus(a):
    return None if a is None else (a[0], (-a[1]) % PRIME)


def on_curve(a):
    return (a[1] * a[1] - a[0] ** 3 - 3) % PRIME == 0


def point_bytes(a):
    return b"\x04" + a[0].to_bytes(32, "big") + a[1].to_bytes(32, "big")      # ECP.ToBytes(b, false), idemix/util.go:57-61


def hash_to_zr(data):
    return int.from_bytes(hashlib.sha256(data).digest(), "big") % ORDER       # HashModOrder, idemix/util.go:46-51


def main():
    fx = json.load(open(os.path.join(HERE, "idemix_fixtures.json")))["msps"]
    rng = random.Random(20260923)
    out = []
    for name in sorted(fx):
        ent = fx[name]
        if "signer_config" not in ent:
            continue
        ipk = rd_fields(bytes.fromhex(ent["ipk"]))
        h_sk, h_rand, ipk_hash = rd_point(ipk[2][0]), rd_point(ipk[3][0]), ipk[10][0]
        sk = int.from_bytes(rd_fields(bytes.fromhex(ent["signer_config"]))[2][0], "big")     # IdemixMSPSignerConfig.sk = 2
        assert on_curve(h_sk) and on_curve(h_rand) and len(ipk_hash) == 32 and 0 < sk < ORDER
        for j in range(6):
            r_nym, r_sk, r_rnym, nonce = (rng.randrange(1, ORDER) for _ in range(4))
            msg = bytes(rng.randrange(256) for _ in range(rng.choice((0, 1, 55, 56, 64, 119, 300, 4608))))
            nym = pt_plus(pt_times(h_sk, sk), pt_times(h_rand, r_nym))                       # MakeNym, idemix/util.go:100-107
            t = pt_plus(pt_times(h_sk, r_sk), pt_times(h_rand, r_rnym))                      # prover, idemix/nymsignature.go:40-46
            c = hash_to_zr(b"sign" + point_bytes(t) + point_bytes(nym) + ipk_hash + msg)
            proof_c = hash_to_zr(c.to_bytes(32, "big") + nonce.to_bytes(32, "big"))
            s_sk, s_rnym = (r_sk + proof_c * sk) % ORDER, (r_rnym + proof_c * r_nym) % ORDER
            # verifier, idemix/nymsignature.go:84-92 - by THIS implementation
            t2 = pt_plus(pt_plus(pt_times(h_sk, s_sk), pt_times(h_rand, s_rnym)), pt_minus(pt_times(nym, proof_c)))
            assert t2 == t and on_curve(t) and on_curve(nym)
            c2 = hash_to_zr(b"sign" + point_bytes(t2) + point_bytes(nym) + ipk_hash + msg)
            assert hash_to_zr(c2.to_bytes(32, "big") + nonce.to_bytes(32, "big")) == proof_c
            out.append(dict(msp=name, index=j, msg=msg.hex(), nym_x="%064x" % nym[0], nym_y="%064x" % nym[1], proof_c="%064x" % proof_c,
                            proof_s_sk="%064x" % s_sk, proof_s_r_nym="%064x" % s_rnym, nonce="%064x" % nonce, t_x="%064x" % t[0], t_y="%064x" % t[1]))
    path = os.path.join(HERE, "idemix_nym_kats.json")
    json.dump(dict(generator="tests/golden/gen_idemix_nym_kats.py", note="second implementation: affine big-int arithmetic, no oracle import",
                   vectors=out), open(path, "w"), indent=0)
    print(len(out), "vectors ->", path)


if __name__ == "__main__":
    main()
