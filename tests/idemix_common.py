"""Shared helpers of the idemix tests: the reference's key-material fixtures (tests/golden/idemix_fixtures.json) and seeded
pseudonym-signature batches made with the oracle."""
import json
import os
import random

import numpy as np

import idemix_oracle as io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fixtures():
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]
    out = {}
    for name, ent in d.items():
        e = {"ipk": io.IssuerPublicKey(bytes.fromhex(ent["ipk"])), "isk": int(ent["isk"], 16)}
        if "signer_config" in ent:
            e["signer"] = io.SignerConfig(bytes.fromhex(ent["signer_config"]))
        out[name] = e
    return out


def be32(x):
    return int(x).to_bytes(32, "big")


class NymBatch:
    """n signatures as the C ABI wants them + the oracle's status for each"""

    def __init__(self):
        self.msgs, self.issuer, self.rows, self.expect, self.what = [], [], [], [], []

    def add(self, issuer_idx, ipk, nym, sig, msg, what="valid", expect=None):
        self.msgs.append(bytes(msg))
        self.issuer.append(issuer_idx)
        self.rows.append((be32(nym[0]), be32(nym[1]), sig["proof_c"], sig["proof_s_sk"], sig["proof_s_r_nym"], sig["nonce"]))
        self.expect.append(io.nym_verify(sig, nym, ipk, bytes(msg)) if expect is None else expect)
        self.what.append(what)

    def arrays(self):
        n = len(self.msgs)
        off = np.zeros(n + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(m) for m in self.msgs])
        arena = np.frombuffer(b"".join(self.msgs) or b"\0", dtype=np.uint8).copy()
        cols = [np.frombuffer(b"".join(r[k] for r in self.rows), dtype=np.uint8).reshape(n, 32).copy() for k in range(6)]
        return arena, off, np.array(self.issuer, dtype=np.uint32), cols, np.array(self.expect, dtype=np.uint8)


MSG_LENGTHS = [0, 1, 2, 17, 25, 26, 27, 63, 64, 89, 90, 91, 154, 155, 500, 1856, 4608]   # 26: header + msg = 3 whole blocks; 90: 4


_BATCH_CACHE = {}


def make_batch(issuers, n, seed, tamper=True):
    """issuers: list of (ipk, sk).  A mix of valid signatures and every kind of invalid / out-of-domain input.
    Deterministic in its arguments and read-only to its users, so the (pure-Python, seconds per hundred signatures) signing is done once
    per process: the GPU suite runs every case under five kernel configurations (the suite's time)."""
    key = (tuple(bytes(ipk.hash) for ipk, _ in issuers), n, seed, tamper)
    if key not in _BATCH_CACHE:
        _BATCH_CACHE[key] = _make_batch(issuers, n, seed, tamper)
    return _BATCH_CACHE[key]


def _make_batch(issuers, n, seed, tamper=True):
    rng = random.Random(seed)
    b = NymBatch()
    for i in range(n):
        k = rng.randrange(len(issuers))
        ipk, sk = issuers[k]
        nym, r_nym = io.make_nym(sk, ipk, rng)
        ln = MSG_LENGTHS[i % len(MSG_LENGTHS)] if i < 3 * len(MSG_LENGTHS) else rng.randrange(0, 300)
        msg = bytes(rng.getrandbits(8) for _ in range(ln))
        sig = io.nym_sign(sk, nym, r_nym, ipk, msg, rng)
        kind = rng.randrange(16) if tamper and i >= len(MSG_LENGTHS) else 0
        if kind <= 7:
            b.add(k, ipk, nym, sig, msg)
        elif kind == 8:      # one bit of the message
            m2 = bytearray(msg or b"\0")
            m2[rng.randrange(len(m2))] ^= 1 << rng.randrange(8)
            b.add(k, ipk, nym, sig, bytes(m2), "msg bit")
        elif kind == 9:      # one bit of one signature field
            f = rng.choice(["proof_c", "proof_s_sk", "proof_s_r_nym", "nonce"])
            v = bytearray(sig[f])
            v[rng.randrange(1, 32)] ^= 1 << rng.randrange(8)
            s2 = dict(sig)
            s2[f] = bytes(v)
            b.add(k, ipk, nym, s2, msg, "field bit " + f)
        elif kind == 10:     # somebody else's pseudonym
            nym2, _ = io.make_nym(sk, ipk, rng)
            b.add(k, ipk, nym2, sig, msg, "other nym")
        elif kind == 11:     # signed under another issuer
            k2 = (k + 1) % len(issuers)
            b.add(k2, issuers[k2][0], nym, sig, msg, "other issuer")
        elif kind == 12:     # unreduced ProofC (c + r does not fit 256 bits unless c is tiny -> use c >= r directly)
            s2 = dict(sig)
            s2["proof_c"] = be32(io.R + rng.randrange(1 << 200))
            b.add(k, ipk, nym, s2, msg, "c >= r")
        elif kind == 13:     # s-value >= r: out of the pinned domain
            s2 = dict(sig)
            f = rng.choice(["proof_s_sk", "proof_s_r_nym"])
            s2[f] = be32(io.R + rng.randrange(1 << 100))
            b.add(k, ipk, nym, s2, msg, "s >= r")
        elif kind == 14:     # nym not on the curve / coordinate >= p
            bad = (nym[0], (nym[1] + 1) % io.P) if rng.randrange(2) else (io.P + 1, nym[1])
            b.add(k, ipk, bad, sig, msg, "nym off curve")
        else:                # commitment at infinity: s_sk = c sk, s_rnym = c r_nym
            c = int.from_bytes(sig["proof_c"], "big")
            s2 = dict(sig)
            s2["proof_s_sk"] = be32(c * sk % io.R)
            s2["proof_s_r_nym"] = be32(c * r_nym % io.R)
            b.add(k, ipk, nym, s2, msg, "t at infinity")
    return b
