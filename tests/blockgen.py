"""Synthetic endorser-transaction blocks with VALID signatures, for the GPU tests, tools/make_bench_blocks.py and the block-pass legs of
bench.py (test infrastructure: it signs with the CPU oracle's sign side, oracle/p256_oracle.c, and encodes with tests/blockbuilder.py -
nothing of the product path imports it).

What the blocks look like is SURVEY.md Appendix C / section 8(d): N transactions x (1 creator signature over Envelope.payload + 3
endorsement signatures over prp || endorser), TxID and proposal hash as the reference's validators recompute them
(protoutil/proputils.go:357-364, protoutil/txutils.go:431-447).  Besides the six fixture signers (tests/golden/block_identities.json) it can
mint any number of identities NOBODY HAS MET: the fixture creator's certificate with another P-256 key in its SubjectPublicKeyInfo.
The pass reads exactly that key out of the certificate (msp/mspimpl.go:408-421); whether the certificate chains to a CA is the MSP's
business (msp/mspimplvalidate.go), not the signature path's - so a re-keyed certificate is, for this path, a new client."""
import base64
import ctypes
import hashlib
import json
import os

import numpy as np

import bccsp_sw_oracle as po
import blockbuilder as bb
import coracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_IDS = [i for i in json.load(open(os.path.join(ROOT, "tests", "golden", "block_identities.json")))["identities"] if i["curve"] == "prime256v1"]


def _pubkey(d32: bytes) -> bytes:
    qx, qy = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    coracle.lib().oracle_p256_pubkey(d32, qx, qy)
    return qx.raw + qy.raw


def make_signer(seed: int):
    """sign(d32, msg) -> DER signature (low-S, as bccsp/sw/ecdsa.go:27-39 produces them); nonces from a seeded generator"""
    rng = np.random.default_rng(seed)
    L = coracle.lib()

    def sign(d32: bytes, msg: bytes) -> bytes:
        nonce = b"\x00" + bytes(rng.integers(1, 255, size=31, dtype=np.uint8))
        r, s = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        assert L.oracle_p256_sign(d32, hashlib.sha256(msg).digest(), nonce, 1, r, s) == 0
        return po.marshal_ecdsa_signature(int.from_bytes(r.raw, "big"), int.from_bytes(s.raw, "big"))
    return sign


def fixture_signers():
    """[(SerializedIdentity bytes, d32)] of the six P-256 fixture identities: 0..3 endorsers, 4..5 creators"""
    return [(bb.serialized_identity("Org1MSP", i["pem"]), int(i["d"], 16).to_bytes(32, "big")) for i in _IDS]


def _pem_der(pem: str) -> bytes:
    body = "".join(line for line in pem.strip().splitlines() if not line.startswith("-----"))
    return base64.b64decode(body)


def _pem_wrap(der: bytes) -> str:
    b = base64.b64encode(der).decode()
    return "-----BEGIN CERTIFICATE-----\n" + "\n".join(b[i:i + 64] for i in range(0, len(b), 64)) + "\n-----END CERTIFICATE-----\n"


def fresh_identities(n: int, seed: int, mspid: str = "Org1MSP"):
    """n identities nobody has met -> [(SerializedIdentity bytes, d32)]: the fixture creator's certificate, re-keyed"""
    tmpl = _IDS[4]
    der = _pem_der(tmpl["pem"])
    at = der.index(_pubkey(int(tmpl["d"], 16).to_bytes(32, "big")))
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        d32 = b"\x00" + bytes(rng.integers(1, 255, size=31, dtype=np.uint8))
        # (another key, and - as a certificate a CA really issued would have - another signature value: its last 20 bytes)
        out.append((bb.serialized_identity(mspid, _pem_wrap(der[:at] + _pubkey(d32) + der[at + 64:-20] + bytes(rng.integers(1, 255, size=20, dtype=np.uint8)))), d32))
    return out


def endorser_tx(t, rng, creator, endorsers, sign, craft=None, ext_bytes=990):
    """one envelope: creator = (identity, d32); endorsers = three of them; craft(t, j, der) may replace endorsement j's signature bytes"""
    cid, cd = creator

    def ends(prp):
        out = []
        for j, (eid, ed) in enumerate(endorsers):
            sig = sign(ed, prp + eid)
            if craft is not None:
                sig = craft(t, j, sig)
            out.append((eid, sig))
        return out
    payload, _ = bb.consistent_endorser_tx("mychannel", cid, bytes(rng.integers(0, 256, size=24, dtype=np.uint8)), bytes(rng.integers(0, 256, size=300, dtype=np.uint8)),
                                           bytes(rng.integers(0, 256, size=ext_bytes, dtype=np.uint8)), ends)
    return bb.envelope(payload, sign(cd, payload))


def endorser_block(n_tx, seed, creators=None, endorsers=None, craft=None, number=1):
    """-> (block bytes, [envelope bytes]).  creators: transaction t is signed by creators[t % len]; endorsers: a pool, three per transaction"""
    fx = fixture_signers()
    creators = creators if creators is not None else fx[4:6]
    endorsers = endorsers if endorsers is not None else fx[:4]
    rng = np.random.default_rng(seed)
    sign = make_signer(seed + 1)
    envs = []
    for t in range(n_tx):
        picks = [int(j) for j in rng.choice(len(endorsers), size=3, replace=False)]
        envs.append(endorser_tx(t, rng, creators[t % len(creators)], [endorsers[j] for j in picks], sign, craft))
    return bb.block(number, envs), envs


def split_envelopes(block: bytes):
    """the envelope byte strings of a marshalled common.Block (BlockData{1 repeated bytes data}) and the block number"""
    def varint(b, i):
        v = s = 0
        while True:
            c = b[i]
            i += 1
            v |= (c & 0x7F) << s
            s += 7
            if not c & 0x80:
                return v, i

    def fields(b):
        i = 0
        while i < len(b):
            key, i = varint(b, i)
            if key & 7 == 0:
                v, i = varint(b, i)
                yield key >> 3, v
            else:
                assert key & 7 == 2
                n, i = varint(b, i)
                yield key >> 3, b[i:i + n]
                i += n
    number, envs = 0, []
    for num, v in fields(block):
        if num == 1:
            for n2, v2 in fields(v):
                if n2 == 1:
                    number = v2
        if num == 2:
            envs = [v2 for n2, v2 in fields(v) if n2 == 1]
    return number, envs


def pack_envelopes(envs) -> bytes:
    """a list of envelopes as one byte string (u32 length prefixes): what tools/make_bench_blocks.py stores for the blocks it patches together"""
    return b"".join(len(e).to_bytes(4, "little") + e for e in envs)


def unpack_envelopes(b: bytes):
    out, i = [], 0
    while i < len(b):
        n = int.from_bytes(b[i:i + 4], "little")
        out.append(b[i + 4:i + 4 + n])
        i += 4 + n
    return out


# ---- signature encodings only Go's asn1 package and the general parser agree on (bccsp/utils/ecdsa.go:43-67) -----------------------
def _der_len(n: int) -> bytes:
    if n < 128:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([0x80 | len(b)]) + b


def der_tlv(tag: int, content: bytes) -> bytes:
    return bytes([tag]) + _der_len(len(content)) + content


def crafted(sig_der: bytes, how: str) -> bytes:
    """a good DER signature re-encoded: 'trailing' (bytes behind the SEQUENCE: accepted by the reference, still VALID), 'third' (a third
    element inside the SEQUENCE: accepted, VALID), 'long_r' (r of 200 bytes, long-form lengths: parses, then r >= n: (false, nil)),
    'long_s' (s of 200 bytes: high-S error), 'neg_r' (negative r: error), 'nonminimal' (long-form length below 128: does not unmarshal)"""
    r, s = po.unmarshal_ecdsa_signature(sig_der)
    ri, si = der_tlv(2, sig_der[4:4 + sig_der[3]]), der_tlv(2, sig_der[6 + sig_der[3]:6 + sig_der[3] + sig_der[5 + sig_der[3]]])
    if how == "trailing":
        return sig_der + b"\x05\x00\xde\xad"
    if how == "third":
        return der_tlv(0x30, ri + si + b"\x02\x01\x07")
    if how == "long_r":
        return der_tlv(0x30, der_tlv(2, b"\x01" * 200) + si)
    if how == "long_s":
        return der_tlv(0x30, ri + der_tlv(2, b"\x01" * 200))
    if how == "neg_r":
        return der_tlv(0x30, der_tlv(2, b"\x80" + r.to_bytes(32, "big")[1:]) + si)
    if how == "nonminimal":
        return b"\x30\x81" + sig_der[1:]
    raise ValueError(how)
