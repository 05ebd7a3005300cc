"""The block pass with the walk ON THE DEVICE (fabric-mod_amd/csrc/block_walk_dev.h): the envelope walk, the signature gates, the
identity lookup, the submission arrays, the digest comparisons and the per-transaction flags as kernels.

CPU tests: the code the kernels are compiled from (block_walk_core.h: the walker template with its counting / writing emitters, the
signature gate, the identity-table hash) against the host walker, the general DER parser and a restatement in Python.
GPU tests: the device walker against the host walker record for record, and the whole pass on the device route against the same pass
on the host route - flags, statuses, keys, digests, memo - on synthetic blocks with every corruption and on the reference's ledgers."""
import base64
import hashlib
import json
import os
import ctypes
import threading

import numpy as np
import pytest

import bccsp_sw_oracle as po
import blockbuilder as bb
import fabgpu
from test_block_prepass import IDS, build_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEDGER = json.load(open(os.path.join(ROOT, "tests", "golden", "ledger_blocks.json")))["blocks"]
LEDGER_RAW = [base64.b64decode(b["block_b64"]) for b in LEDGER]


def _mutants(blk, rng, count):
    base = np.frombuffer(blk, dtype=np.uint8)
    for _ in range(count):
        m = base.copy()
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, m.size))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                m[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
            elif kind == 1:
                m[pos] = np.uint8(rng.integers(0, 256))
            elif kind == 2:
                m[pos] = 0xFF
            else:
                m = m[: m.size - int(rng.integers(0, 16))].copy()
        yield m.tobytes()


def big_block(n_tx, rng, bad_every=0):
    """n_tx endorser transactions with three endorsements each; signatures are well-formed DER but not valid (walk-level tests)."""
    sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS if i["curve"] == "prime256v1"]
    fake = b"\x30\x44\x02\x20" + b"\x11" * 32 + b"\x02\x20" + b"\x22" * 32
    envs = []
    for t in range(n_tx):
        payload, _ = bb.consistent_endorser_tx("mychannel", sid[4 + t % 2], bytes(rng.integers(0, 256, size=24, dtype=np.uint8)),
                                               bytes(rng.integers(0, 256, size=300, dtype=np.uint8)), bytes(rng.integers(0, 256, size=900, dtype=np.uint8)),
                                               lambda prp: [(sid[j], fake) for j in (0, 1, 2)], bad_txid=bool(bad_every) and t % bad_every == 5)
        envs.append(bb.envelope(payload, fake))
    return bb.block(3, envs)


# ---- CPU: the shared code ------------------------------------------------------------------------------------------------------
def test_two_run_procedure_equals_the_host_walker():
    """count per envelope -> exclusive prefix sum -> write at the assigned offsets (what the kernels do), run serially on the host:
    the same tuples, prefixes, hash checks, gather spans and offsets as ParseBlock, on synthetic blocks with every corruption, on the
    reference's ledger blocks, on a multi-chunk block and on a few thousand mutants (where rollbacks and refusals happen)."""
    rng = np.random.default_rng(77)
    blk, _ = build_block(52, rng)
    assert fabgpu.block_walk_twopass_compare(blk) == (True, "")
    for raw in LEDGER_RAW:
        assert fabgpu.block_walk_twopass_compare(raw) == (True, "")
    assert fabgpu.block_walk_twopass_compare(big_block(600, rng, bad_every=97)) == (True, "")
    refused = understood = 0
    for raw in _mutants(blk, rng, 2500):
        got = fabgpu.block_walk_twopass_compare(raw)
        if got is None:
            refused += 1
            continue
        assert got == (True, ""), got
        understood += 1
    assert understood > 1000 and refused > 10
    # envelopes with more records than the counting run's slot holds (8 tuples: block_walk_core.h EnvStash) beside ones that fit
    import blockgen
    fx, sign = blockgen.fixture_signers(), blockgen.make_signer(991)
    many = [blockgen.endorser_tx(t, rng, fx[4 + t % 2], [fx[int(j)] for j in rng.integers(0, 4, size=3 + 3 * (t % 4))], sign) for t in range(24)]
    assert fabgpu.block_walk_twopass_compare(bb.block(2, many)) == (True, "")
    # an envelope list of one, and a block without transactions
    assert fabgpu.block_walk_twopass_compare(bb.block(1, [bb.envelope(b"\x0a\x02\x0a\x00", b"")])) == (True, "")
    assert fabgpu.block_walk_twopass_compare(bb.block(1, [])) == (True, "")


def _der_int(v: bytes) -> bytes:
    return b"\x02" + bytes([len(v)]) + v


def _gate_cases():
    """signatures for the gate tests: (bytes, produced_by_a_signer)"""
    rng = np.random.default_rng(3)
    half = po.N >> 1
    cases = []
    for _ in range(3000):                                   # what signers produce: every length of r and s, low and high s
        r = int.from_bytes(bytes(rng.integers(0, 256, size=int(rng.integers(1, 33)), dtype=np.uint8)), "big") or 1
        s = int.from_bytes(bytes(rng.integers(0, 256, size=int(rng.integers(1, 33)), dtype=np.uint8)), "big") or 1
        cases.append((po.marshal_ecdsa_signature(r, s), True))
    for _ in range(1500):                                   # full-length scalars: half of them high-S
        r = int.from_bytes(bytes(rng.integers(0, 256, size=32, dtype=np.uint8)), "big") or 1
        s = int.from_bytes(bytes(rng.integers(0, 256, size=32, dtype=np.uint8)), "big") or 1
        cases.append((po.marshal_ecdsa_signature(r, s), True))
    for s in (1, half - 1, half, half + 1, po.N - 1, po.N, (1 << 256) - 1):
        for r in (1, po.N - 1, po.N, (1 << 256) - 1):
            cases.append((po.marshal_ecdsa_signature(r, s), True))
    for k in range(1, 32):                                  # s differing from n/2 in one byte only, on either side
        for delta in (-1, 1):
            cases.append((po.marshal_ecdsa_signature(7, half + delta * (1 << (8 * k))), True))
    good = po.marshal_ecdsa_signature(0x1234 << 200, 0x77 << 100)
    odd = [b"", b"\x30", b"\x30\x00", good + b"\x00", good[:-1], b"\x30\x81" + bytes([len(good) - 2]) + good[2:],
           b"\x30\x06" + _der_int(b"\x00") + _der_int(b"\x01"), b"\x30\x06" + _der_int(b"\x01") + _der_int(b"\x00"),
           b"\x30\x06" + _der_int(b"\x80") + _der_int(b"\x01"), b"\x30\x08" + _der_int(b"\x00\x01") + _der_int(b"\x01"),
           b"\x30\x08" + _der_int(b"\x00\x80") + _der_int(b"\x7f"), b"\x31" + good[1:], b"\x30\x03\x02\x01",
           b"\x30" + bytes([2 + 34 + 3]) + _der_int(b"\x01" + b"\x00" * 33) + _der_int(b"\x01"),       # r of 34 bytes
           b"\x30" + bytes([2 + 33 + 3]) + _der_int(b"\x00" + b"\xff" * 32) + _der_int(b"\x01"),       # r = 2^256 - 1 with its sign byte
           b"\x30\x46" + _der_int(b"\x00" + b"\xff" * 32) + _der_int(b"\x00" + b"\xff" * 32),        # the longest common shape: 72 bytes
           b"\x30\x47" + _der_int(b"\x00" + b"\xff" * 32) + _der_int(b"\x00" + b"\xff" * 32) + b"\x00", b"\x30" * 80, bytes(73), bytes(200)]
    cases += [(c, False) for c in odd]
    for base_sig in (good, po.marshal_ecdsa_signature((1 << 255) + 12345, half - 99)):
        base = np.frombuffer(base_sig, dtype=np.uint8)
        for _ in range(1500):                               # mutants of a good signature
            m = base.copy()
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, m.size))] = np.uint8(rng.integers(0, 256))
            if rng.integers(0, 8) == 0:
                m = m[: m.size - int(rng.integers(1, 5))]
            cases.append((m.tobytes(), False))
    return cases


def test_device_signature_gate_equals_the_general_parser():
    """gate_sig_fast decides only what UnmarshalECDSASignature + IsLowS (bccsp/utils/ecdsa.go:43-92) would decide the same way, and
    declines nothing a signer produces."""
    n_submit = n_high = n_declined = 0
    for sig, from_signer in _gate_cases():
        code, r32, s32 = fabgpu.gate_sig_fast(sig)
        if not sig:
            assert code == 2
            continue
        rc, gr, gs, flags = fabgpu.unmarshal_ecdsa_signature(sig)
        if code == 0:
            assert rc == 0 and flags == 0 and (gr, gs) == (r32, s32) and fabgpu.is_low_s(gs), sig.hex()
            n_submit += 1
        elif code == 1:
            assert rc == 0 and flags == 0 and not fabgpu.is_low_s(gs), sig.hex()
            n_high += 1
        else:
            assert code == 3
            n_declined += 1
        if from_signer:
            assert code in (0, 1), sig.hex()
    assert n_submit > 1500 and n_high > 300 and n_declined > 100, (n_submit, n_high, n_declined)


def _long_len(n: int) -> bytes:
    if n < 128:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([0x80 | len(b)]) + b


def _der(tag: int, content: bytes) -> bytes:
    return bytes([tag]) + _long_len(len(content)) + content


def _general_gate_cases():
    """encodings only the general parser decides: long-form lengths where DER demands them (integers of 128+ bytes), trailing bytes
    inside and behind the SEQUENCE, negative / zero / non-minimal integers, non-minimal and indefinite lengths, foreign tags,
    truncations - and what a signer produces, for good measure"""
    rng = np.random.default_rng(17)
    half = po.N >> 1
    out = [c for c, _ in _gate_cases()]
    ints = [b"\x01", b"\x7f", b"\x00\x80", b"\x00" + b"\xff" * 32, b"\x01" + b"\x00" * 32, b"\x7f" * 40, b"\x00" + b"\x80" * 127, b"\x01" * 128,
            b"\x01" * 200, b"\x00" + b"\xff" * 300, b"\x80", b"\xff\x7f", b"\x00", b"\x00\x01", b"\xff\xff", b"", half.to_bytes(32, "big"),
            (half + 1).to_bytes(32, "big"), b"\x00" + (po.N - 1).to_bytes(32, "big"), b"\x00" * 2 + b"\x80"]
    for r in ints:
        for s_ in ints:
            body = _der(2, r) + _der(2, s_)
            out.append(_der(0x30, body))
            out.append(_der(0x30, body) + b"\x05\x00")                       # behind the SEQUENCE: `rest`, discarded
            out.append(_der(0x30, body + b"\x02\x01\x07"))                   # a third element inside: allowed by asn1's struct parser
    good = po.marshal_ecdsa_signature(0x1234 << 200, 0x77 << 100)
    body = good[2:]
    out += [b"\x30\x81" + bytes([len(body)]) + body,                         # non-minimal long form
            b"\x30\x82\x00" + bytes([len(body)]) + body,                      # leading zero in the length
            b"\x30\x80" + body + b"\x00\x00",                                # indefinite
            b"\x30\x84\x00\x00\x00" + bytes([len(body)]) + body,
            b"\x30\x85\x01\x00\x00\x00\x00" + body,                         # length too large
            b"\x3f\x30" + bytes([len(body)]) + body,                          # high-tag-number form
            b"\x10" + bytes([len(body)]) + body,                              # SEQUENCE without the constructed bit
            b"\x30" + bytes([len(body) + 1]) + body,                          # content beyond the input
            b"\x30\x04\x02\x02\x00\x01", b"\x30\x02\x02\x00", b"\x30\x03\x02\x81\x00", b"\x30\x04\x02\x01\x01\x02", b"\x30\x05\x02\x01\x01\x02\x01",
            b"\x30\x06\x02\x01\x01\x04\x01\x01", b"\x30\x06\x04\x01\x01\x02\x01\x01"]
    big = _der(0x30, _der(2, b"\x01" * 200) + _der(2, b"\x01"))
    base = np.frombuffer(big, dtype=np.uint8)
    for _ in range(1500):                                                    # mutants of a long-form signature
        m = base.copy()
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, min(m.size, 12) if rng.integers(0, 2) else m.size))] = np.uint8(rng.integers(0, 256))
        if rng.integers(0, 8) == 0:
            m = m[: m.size - int(rng.integers(1, 5))]
        out.append(m.tobytes())
    return out


def _reference_gate(sig: bytes):
    """the any-gate's answer derived from the PYTHON oracle (oracle/bccsp_sw_oracle.py: Go's asn1 + bccsp/utils restated independently of
    the C++ in this tree): (code, r, s)"""
    if not sig:
        return 2, None, None
    try:
        r, s_ = po.unmarshal_ecdsa_signature(sig)
    except Exception:   # noqa: BLE001  (ASN1Error / BCCSPError: does not unmarshal, or r, s <= 0)
        return 4, None, None
    if not po.is_low_s(s_):
        return 1, None, None
    if r >> 256:
        return 5, None, None
    return 0, r, s_


def test_general_signature_gate_equals_both_general_parsers():
    """gate_sig_any - what the device route applies to EVERY signature, so that no encoding takes a block off the device - against
    the C++ general parser (UnmarshalECDSASignature + IsLowS, Go's error texts) and against the independent Python oracle, on every
    shape: nothing is declined any more, and every outcome (submit with these r, s / high-S / does not unmarshal / r beyond 256 bits)
    is the reference's (bccsp/sw/ecdsa.go:41-57 up to the arithmetic)."""
    seen = {0: 0, 1: 0, 2: 0, 4: 0, 5: 0}
    for sig in _general_gate_cases():
        code, r32, s32 = fabgpu.gate_sig_any(sig)
        want, wr, ws = _reference_gate(sig)
        assert code == want, (sig.hex(), code, want)
        if code == 0:
            assert int.from_bytes(r32, "big") == wr and int.from_bytes(s32, "big") == ws, sig.hex()
        if sig:
            rc, gr, gs, flags = fabgpu.unmarshal_ecdsa_signature(sig)
            mine = 4 if rc != 0 else (1 if (flags & 2) or not fabgpu.is_low_s(gs) else (5 if flags & 1 else 0))
            assert code == mine, (sig.hex(), code, mine)
            if code == 0:
                assert (gr, gs) == (r32, s32)
        seen[code] += 1
    assert seen[0] > 1500 and seen[1] > 300 and seen[4] > 500 and seen[5] > 20 and seen[2] >= 1, seen


def _identity_cases():
    """SerializedIdentity byte strings for the identity decoder: every certificate fixture of the reference (P-256, P-384, RSA ...),
    their PEM re-wrapped in other layouts, mutants, and things that are not certificates at all"""
    rng = np.random.default_rng(23)
    chains = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_cert_chains.json")))
    ders = [base64.b64decode(v) for _, v in sorted(chains["certs"].items())]
    out = []

    def pem(der, width=64, eol=b"\n", head=b"-----BEGIN CERTIFICATE-----", tail=b"-----END CERTIFICATE-----", trailer=b"\n"):
        b = base64.b64encode(der)
        lines = [b[i:i + width] for i in range(0, len(b), width)]
        return head + eol + eol.join(lines) + eol + tail + trailer
    class _B:   # blockbuilder.serialized_identity takes text; these armours are bytes (and not always text)
        @staticmethod
        def serialized_identity(mspid, pem_bytes):
            return bb.fbytes(1, mspid.encode()) + bb.fbytes(2, pem_bytes)
    sid = _B.serialized_identity
    for d in ders:
        out.append(sid("Org1MSP", pem(d)))
    for d in ders[:25]:
        out.append(sid("Org1MSP", pem(d, width=76, eol=b"\r\n")))
        out.append(sid("Org2MSP", pem(d, width=1 << 20, trailer=b"")))
        out.append(sid("Org2MSP", b"junk before\n" + pem(d)))                # PemToDer scans for the marker
        out.append(sid("Org2MSP", pem(d, width=61, eol=b" \t\n")))
        out.append(sid("Org1MSP", pem(d, tail=b"-----END CERTIFICATE----")))   # broken END marker
        out.append(sid("Org1MSP", pem(d, head=b"-----BEGIN CERTIFICATE----")))
        out.append(sid("Org1MSP", pem(d)[:-30]))
        out.append(sid("Org1MSP", pem(d).replace(b"A", b"*", 1)))
        out.append(bb.fbytes(1, b"Org1MSP") + bb.fbytes(2, pem(d)) + bb.fbytes(2, pem(d)))        # id_bytes twice: ambiguous
        out.append(bb.fbytes(1, b"Org1MSP"))                                                       # no id_bytes
    out += [b"", b"\x00", b"\x12\x00", sid("X", b""), sid("X", b"-----BEGIN CERTIFICATE-----"),
            sid("X", b"-----BEGIN CERTIFICATE-----\n-----END CERTIFICATE-----\n"),
            sid("X", b"-----BEGIN CERTIFICATE-----\nAAAA\n-----END CERTIFICATE-----\n"), bytes(rng.integers(0, 256, size=700, dtype=np.uint8))]
    for i in IDS:
        out.append(sid("Org1MSP", i["pem"].encode()))
    base = np.frombuffer(out[0], dtype=np.uint8)
    for _ in range(1200):                                                    # mutants of a good identity: text and (through it) DER
        m = base.copy()
        for _ in range(int(rng.integers(1, 4))):
            m[int(rng.integers(0, m.size))] = np.uint8(rng.integers(0, 256))
        out.append(m.tobytes())
    d0 = ders[0]
    for _ in range(800):                                                     # mutants of the DER under intact PEM armour
        m = np.frombuffer(d0, dtype=np.uint8).copy()
        for _ in range(int(rng.integers(1, 3))):
            m[int(rng.integers(0, m.size))] = np.uint8(rng.integers(0, 256))
        out.append(sid("Org1MSP", pem(m.tobytes())))
    # certificates longer than the device decoder's window (4096 base64 digits = 3 KiB of DER): decided all the same when the key lies
    # inside the window (here: bytes behind the Certificate) ...
    out.append(sid("Org1MSP", pem(d0 + bytes(3100))))
    out.append(sid("Org1MSP", pem(_cert_with_long_issuer(d0, 1500))))         # (a long issuer name that still fits)
    # ... and the one thing it leaves to bccsp/sw: SubjectPublicKeyInfo BEYOND the window (kilobytes of issuer name in front of it)
    out.append(sid("Org1MSP", pem(_cert_with_long_issuer(d0, 3300))))
    return out


def _der_tlv(b, at):
    """(tag, header length, content length) of the DER element at b[at:]"""
    l = b[at + 1]
    if l < 0x80:
        return b[at], 2, l
    nb = l & 0x7F
    return b[at], 2 + nb, int.from_bytes(b[at + 2:at + 2 + nb], "big")


def _der_wrap(tag, content):
    n = len(content)
    if n < 0x80:
        return bytes([tag, n]) + content
    k = (n.bit_length() + 7) // 8
    return bytes([tag, 0x80 | k]) + n.to_bytes(k, "big") + content


def _cert_with_long_issuer(der, pad):
    """the certificate with `pad` more bytes (an extra RDN set full of zeros) inside its issuer Name: what stands behind the issuer -
    validity, subject, SubjectPublicKeyInfo - moves back by about that much.  The signature no longer matches, which no key
    extraction looks at."""
    _, h0, _ = _der_tlv(der, 0)
    _, h1, l1 = _der_tlv(der, h0)
    tbs = der[h0 + h1:h0 + h1 + l1]
    rest = der[h0 + h1 + l1:]
    at, parts = 0, []
    while at < len(tbs):
        t, h, l = _der_tlv(tbs, at)
        parts.append(tbs[at:at + h + l])
        at += h + l
    k = 3 if parts[0][0] == 0xA0 else 2                                      # [0] version, serial, signature, ISSUER
    _, h, l = _der_tlv(parts[k], 0)
    parts[k] = _der_wrap(0x30, parts[k][h:h + l] + _der_wrap(0x31, bytes(pad)))
    return _der_wrap(0x30, _der_wrap(0x30, b"".join(parts)) + rest)


def test_identity_table_hash_restated():
    """The table hash covers the length, the last 64 bytes and 64 bytes spread over the whole string (two coalesced rows for a
    wavefront: lane l folds byte l of the tail and byte l * len / 64); equality is always decided on all the bytes, so the hash only has
    to spread the identities a provider meets - including ones that share their last 64 bytes."""
    M = (1 << 64) - 1

    def restated(b):
        m = min(len(b), 64)
        total = 0
        for lane in range(64):
            h = 0xCBF29CE484222325
            if lane < m:
                h = ((h ^ b[len(b) - m + lane]) * 0x100000001B3) & M
            if len(b):
                h = ((h ^ b[(lane * len(b)) >> 6]) * 0x100000001B3) & M
            h ^= h >> 31
            h = (h * (((0x9E3779B97F4A7C15 * (2 * lane + 1)) & M) | 1)) & M
            total = (total + (h ^ (h >> 29))) & M
        h = total ^ ((len(b) * 0xD6E8FEB86659FD93) & M)
        h ^= h >> 32
        h = (h * 0xD6E8FEB86659FD93) & M
        return h ^ (h >> 29)
    rng = np.random.default_rng(9)
    seen = set()
    for n in [0, 1, 63, 64, 65, 127, 128, 700, 801, 4097]:
        b = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        assert fabgpu.identity_table_hash(b) == restated(b)
        seen.add(fabgpu.identity_table_hash(b))
    assert len(seen) == 10
    # the identities of the fixtures all hash apart, and so do re-keyed copies of ONE certificate that share their last 64 bytes
    ids = [bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS]
    assert len({fabgpu.identity_table_hash(i) for i in ids}) == len(ids)
    base = bytearray(ids[0])
    twins = set()
    for k in range(500):
        t = bytearray(base)
        t[300:386] = bytes(rng.integers(65, 91, size=86, dtype=np.uint8))      # the 86 base64 characters a 64-byte public key takes
        twins.add(fabgpu.identity_table_hash(bytes(t)))
    assert len(twins) == 500                                               # (half a dozen of them are among the spread bytes)


# ---- GPU -----------------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def csp():
    c = fabgpu.GPUCSP(device=0)
    yield c
    c.close()


@pytest.mark.gpu
def test_wavefront_signature_gate_equals_the_lane_form(csp):
    """The kernels run the gate with a signature's bytes spread over a wavefront (lane permutes, ballots); it must give the code and
    the (r, s) of walk::gate_sig_fast - which the CPU test above holds against the general parser - on every case."""
    cases = _general_gate_cases()
    code, r, s = fabgpu.gate_probe(csp, cases)
    seen = {0: 0, 1: 0, 2: 0, 4: 0, 5: 0}
    for i, sig in enumerate(cases):
        want, wr, ws = fabgpu.gate_sig_any(sig)
        assert int(code[i]) == want, (i, sig.hex(), int(code[i]), want)
        if want == 0:
            assert bytes(r[i]) == wr and bytes(s[i]) == ws, sig.hex()
        seen[want] += 1
    assert min(seen.values()) > 0 and seen[0] > 1500 and seen[1] > 300 and seen[4] > 500


@pytest.mark.gpu
def test_device_identity_decoder_equals_the_host_decoder(csp):
    """A wavefront per identity: SerializedIdentity -> PEM (any line layout) -> DER -> SubjectPublicKeyInfo -> curve membership.  Same
    answer as the host route's IdentityToP256 + PublicKeyOnCurve on the reference's 102 certificate fixtures (P-256, P-384 ...), on
    re-wrapped / broken armour, on 2 000 mutants of text and DER, on non-certificates, on certificates longer than its 3 KiB window; only
    a certificate whose key lies beyond that window is left undecided."""
    cases = _identity_cases()
    code, key = fabgpu.idfix_probe(csp, cases)
    n_key = n_not = 0
    assert fabgpu.identity_to_p256(cases[-2]) is not None and fabgpu.identity_to_p256(cases[-3]) is not None
    for i, ident in enumerate(cases[:-1]):
        want = fabgpu.identity_to_p256(ident)
        if want is None:
            assert int(code[i]) == 1, (i, ident[:80])
            assert not key[i].any()
            n_not += 1
        else:
            assert int(code[i]) == 0 and bytes(key[i]) == want, (i, ident[:80])
            n_key += 1
    assert int(code[-1]) == 2 and fabgpu.identity_to_p256(cases[-1]) is not None
    assert n_key > 300 and n_not > 300, (n_key, n_not)


@pytest.mark.gpu
def test_device_walker_equals_host_walker(csp):
    rng = np.random.default_rng(78)
    blk, _ = build_block(130, rng)
    blocks = [blk, big_block(2600, rng, bad_every=97)] + LEDGER_RAW + list(_mutants(build_block(12, rng)[0], rng, 300))
    compared = 0
    for raw in blocks:
        same, declined, text = fabgpu.block_walk_compare(csp, raw)
        assert same, text
        if not declined:
            compared += 1
        else:
            assert text in ("no envelopes", "no signature in the block"), text
    assert compared > 200


@pytest.mark.gpu
def test_envelopes_with_more_records_than_the_count_kernels_slot_are_walked_again(csp):
    """The count kernel keeps an envelope's records in a fixed slot (8 tuples, 2 prefixes, 4 hash checks: block_walk_core.h EnvStash) and
    the emit kernel copies them; an envelope with more - here 3, 6, 9 and 12 endorsements per transaction, side by side in one block -
    is walked a second time as before.  Same records as the host walker either way, and the pass decides every signature."""
    import blockgen
    fx = blockgen.fixture_signers()
    rng = np.random.default_rng(4242)
    sign = blockgen.make_signer(4243)
    envs = []
    for t in range(48):
        k = 3 + 3 * (t % 4)
        envs.append(blockgen.endorser_tx(t, rng, fx[4 + t % 2], [fx[int(j)] for j in rng.integers(0, 4, size=k)], sign))
    blk = bb.block(3, envs)
    same, declined, text = fabgpu.block_walk_compare(csp, blk)
    assert same and not declined, text
    out = fabgpu.preverify_block(csp, blk)
    assert (out["tx_flags"] == 0).all() and len(out["tuple_status"]) == sum(1 + 3 + 3 * (t % 4) for t in range(48))
    assert (out["tuple_status"] == 0).all()


def _learn(csp, blk):
    """first sight of a block's identities: the host route decodes and caches them"""
    fabgpu.preverify_block(csp, blk)


def clean_modes_block(n_tx, rng):
    """build_block without the garbage-DER transactions (those send a block to the host walk): every other corruption stays"""
    blk, want = None, None
    p256 = [i for i in IDS if i["curve"] == "prime256v1"]
    sid = {i["cn"]: bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS}
    p384 = [i for i in IDS if i["curve"] != "prime256v1"][0]
    endorsers, creators = p256[:4], p256[4:6]
    envs, want = [], []
    for t in range(n_tx):
        creator = creators[t % 2]
        cbytes = sid[creator["cn"]]
        ext = bytes(rng.integers(0, 256, size=int(rng.integers(100, 1200)), dtype=np.uint8))
        ccpp = bytes(rng.integers(0, 256, size=200, dtype=np.uint8))
        nonce = bytes(rng.integers(0, 256, size=24, dtype=np.uint8))
        mode = t % 11
        flag = [fabgpu.TX_ALL_SIGNATURES_VALID]

        def sign_ends(prp, mode=mode, flag=flag):
            ends = []
            for j in rng.choice(4, size=3, replace=False):
                e = endorsers[j]
                r, s = po.sign_raw(int(e["d"], 16), hashlib.sha256(prp + sid[e["cn"]]).digest(), int(rng.integers(1, 1 << 62)))
                ends.append([sid[e["cn"]], po.marshal_ecdsa_signature(r, s), r, s])
            if mode == 1:
                ends[1][1] = po.marshal_ecdsa_signature(ends[1][2] + 1, ends[1][3]); flag[0] = fabgpu.TX_BAD_ENDORSEMENT
            if mode == 2:                               # high-S
                ends[0][1] = po.marshal_ecdsa_signature(ends[0][2], po.N - ends[0][3]); flag[0] = fabgpu.TX_BAD_ENDORSEMENT
            if mode == 3:                               # P-384 endorser
                ends[2][0] = sid[p384["cn"]]; flag[0] = fabgpu.TX_NEEDS_SW
            if mode == 4:                               # empty signature
                ends[2][1] = b""; flag[0] = fabgpu.TX_BAD_ENDORSEMENT
            if mode == 5:                               # r >= n
                ends[0][1] = po.marshal_ecdsa_signature(po.N + 5, ends[0][3]); flag[0] = fabgpu.TX_BAD_ENDORSEMENT
            return [(e[0], e[1]) for e in ends]
        payload, prp = bb.consistent_endorser_tx("mychannel", cbytes, nonce, ccpp, ext, sign_ends, bad_txid=(mode == 6), bad_phash=(mode == 7))
        if mode == 6:
            flag[0] = fabgpu.TX_BAD_TXID
        if mode == 7:
            flag[0] = fabgpu.TX_BAD_PROPOSAL_HASH
        if mode == 8:                                   # not a peer.Transaction
            hdr = bb.fbytes(1, bb.channel_header(3, "mychannel", bb.compute_txid(b"n", cbytes))) + bb.fbytes(2, bb.signature_header(cbytes, b"n"))
            payload = bb.fbytes(1, hdr) + bb.fbytes(2, b"\x0a\xff\xff\xff\xff\x0f" + b"junk")
            flag[0] = fabgpu.TX_NOT_UNDERSTOOD
        r, s = po.sign_raw(int(creator["d"], 16), hashlib.sha256(payload).digest(), int(rng.integers(1, 1 << 62)))
        if mode == 9:                                   # creator signed something else
            r, s = po.sign_raw(int(creator["d"], 16), hashlib.sha256(payload + b"!").digest(), 99); flag[0] = fabgpu.TX_BAD_CREATOR_SIGNATURE
        if mode == 10:                                  # a CONFIG envelope
            payload = bb.endorser_tx_payload(1, "mychannel", "not-a-hash", cbytes, nonce, [(ccpp, prp, [])])
            r, s = po.sign_raw(int(creator["d"], 16), hashlib.sha256(payload).digest(), int(rng.integers(1, 1 << 62)))
        envs.append(bb.envelope(payload, po.marshal_ecdsa_signature(r, s)))
        want.append(flag[0])
    return bb.block(9, envs), np.array(want, dtype=np.uint8)


def _same(a, b, keys):
    for k in keys:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


@pytest.mark.gpu
def test_pass_on_the_device_route_equals_the_host_route(csp, monkeypatch):
    """The same block through both routes (the staging threshold decides which): identical flags, per-tuple statuses, spans, keys,
    digests and memo; every corruption the device decides itself is in the block."""
    rng = np.random.default_rng(101)
    blk, want = clean_modes_block(220, rng)
    csp.set_option("pass_stage_min_bytes", 1 << 40)
    host = fabgpu.preverify_block(csp, blk)                                # host route; learns the identities
    host2 = fabgpu.preverify_block2(csp, blk, block_seq=1, seed_memo=True)
    assert (host["tx_flags"] == want).all() and fabgpu.pass_routes(csp)["device_walks"] == 0
    csp.set_option("pass_stage_min_bytes", 1)
    dev = fabgpu.preverify_block(csp, blk)
    routes = fabgpu.pass_routes(csp)
    assert routes["device_walks"] == 1, routes
    _same(host, dev, ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status"])
    assert set(np.unique(dev["tuple_status"])) == {0, 1, 2, 3, 6, 7}
    fabgpu.memo_evict_block(csp, 1)
    dev2 = fabgpu.preverify_block2(csp, blk, block_seq=2, seed_memo=True)
    assert fabgpu.pass_routes(csp)["device_walks"] == 2
    _same(host2, dev2, ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status", "tuple_spans", "tuple_digest", "tuple_hashed", "tuple_qxy"])
    assert dev2["memo_seeded"] == host2["memo_seeded"] > 0 and dev2["n_keyed"] == host2["n_keyed"]
    # the memo seeded by the device route answers bccsp.Verify lookups with the tuple's status
    hits = 0
    for i in range(len(dev2["tuple_status"])):
        if not dev2["tuple_hashed"][i] or dev2["tuple_status"][i] > 3:
            continue
        sp = [int(x) for x in dev2["tuple_spans"][i]]
        sig = dev2["arena"][sp[6]:sp[6] + sp[7]]
        q = bytes(dev2["tuple_qxy"][i])
        assert fabgpu.memo_lookup(csp, q[:32], q[32:], sig, bytes(dev2["tuple_digest"][i])) == int(dev2["tuple_status"][i])
        hits += 1
    assert hits == dev2["memo_seeded"]
    # digests are SHA-256 of prefix || suffix as the block holds them
    for i in range(0, len(dev2["tuple_status"]), 17):
        if dev2["tuple_hashed"][i]:
            sp = [int(x) for x in dev2["tuple_spans"][i]]
            msg = dev2["arena"][sp[2]:sp[2] + sp[3]] + dev2["arena"][sp[4]:sp[4] + sp[5]]
            assert bytes(dev2["tuple_digest"][i]) == hashlib.sha256(msg).digest()
    # lean form (what a memo-seeding caller asks for) and the NO_BLOCK_SIGS form
    lean = fabgpu.preverify_block2(csp, blk, block_seq=3, seed_memo=True, lean=True)
    assert (lean["tx_flags"] == want).all() and lean["memo_seeded"] == host2["memo_seeded"]
    assert fabgpu.pass_routes(csp)["device_walks"] == 3


KEYS_ALL = ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status", "tuple_spans", "tuple_digest", "tuple_hashed", "tuple_qxy"]


def _both_routes(csp, monkeypatch, blk, seq, device_first=True, **kw):
    """the same block through the device route and through the host route of ONE provider -> (device answer, host answer)"""
    def run(device):
        csp.set_option("pass_stage_min_bytes", 1 if device else 1 << 40)
        before = fabgpu.pass_routes(csp)
        out = fabgpu.preverify_block2(csp, blk, block_seq=seq + (0 if device else 1), **kw)
        after = fabgpu.pass_routes(csp)
        assert after["device_walks"] - before["device_walks"] == (1 if device else 0), after
        return out
    if device_first:
        dev = run(True)
        return dev, run(False)
    host = run(False)
    return run(True), host


@pytest.mark.gpu
def test_counts_from_the_host_and_counts_from_the_device_give_one_answer(csp, monkeypatch):
    """The device route can start from per-envelope counts the host took while the block travelled (the count kernel's own code,
    WalkRequest::host_counts, FABGPU_PASS_HOST_COUNTS=1) instead of running the count kernel and the scan.  Same block, same answers -
    spans, digests, statuses, flags - on a block that carries every kind of envelope the generator knows (config envelopes without
    tuples, garbage, oversize fields)."""
    rng = np.random.default_rng(23)
    blk, want = build_block(90, rng)
    csp.set_option("pass_stage_min_bytes", 1)
    answers = []
    for mode in ("0", "1", "0"):
        csp.set_option("pass_host_counts", 1 if mode == "1" else -1)
        before = fabgpu.pass_routes(csp)
        answers.append(fabgpu.preverify_block2(csp, blk, block_seq=70 + len(answers)))
        assert fabgpu.pass_routes(csp)["device_walks"] - before["device_walks"] == 1
    assert (answers[0]["tx_flags"] == want).all()
    _same(answers[0], answers[1], KEYS_ALL)
    _same(answers[1], answers[2], KEYS_ALL)


@pytest.mark.gpu
def test_memo_built_by_the_device_equals_the_memo_seeded_on_the_host(csp, monkeypatch):
    """The verdict memo of a device-route pass is built by the device (keys, offsets, statuses, slot table copied into pinned memory of the
    provider's table: block_walk_dev.h WalkOut::memo_*); FABGPU_PASS_DEVICE_MEMO=0 brings the arrays back and lets GPUCSP::SeedMemo build it
    as before.  Same block - every kind of corruption the generator knows - same entries: every tuple's (key, signature, digest) finds the
    same status or the same miss, the same number of entries, and nothing is left after eviction."""
    rng = np.random.default_rng(31)
    blk, want = build_block(120, rng)
    csp.set_option("pass_stage_min_bytes", 1)
    answers = []
    for k, mode in enumerate(("1", "0")):
        csp.set_option("pass_device_memo", 1 if mode == "1" else -1)
        before = fabgpu.pass_routes(csp)
        out = fabgpu.preverify_block2(csp, blk, block_seq=90 + k, seed_memo=True)
        assert fabgpu.pass_routes(csp)["device_walks"] - before["device_walks"] == 1
        assert (out["tx_flags"] == want).all()
        found = []
        for i in range(len(out["tuple_status"])):
            sp = [int(x) for x in out["tuple_spans"][i]]
            sig, q = out["arena"][sp[6]:sp[6] + sp[7]], bytes(out["tuple_qxy"][i])
            if not sig:
                found.append(None)
                continue
            found.append(fabgpu.memo_lookup(csp, q[:32], q[32:], sig, bytes(out["tuple_digest"][i])))
        hashed = [i for i in range(len(found)) if out["tuple_hashed"][i] and out["tuple_status"][i] <= 3 and 0 < int(out["tuple_spans"][i][7]) <= 1024]
        assert out["memo_seeded"] == len(hashed) > 100
        for i in hashed:
            assert found[i] == int(out["tuple_status"][i])
        assert all(found[i] is None for i in range(len(found)) if i not in set(hashed))
        assert fabgpu.memo_has_block(csp, 90 + k) == out["memo_seeded"]
        fabgpu.memo_evict_block(csp, 90 + k)
        assert fabgpu.memo_has_block(csp, 90 + k) == 0
        answers.append((out, found))
    _same(answers[0][0], answers[1][0], KEYS_ALL)
    assert answers[0][1] == answers[1][1]


@pytest.mark.gpu
def test_memo_of_a_block_whose_signatures_are_far_longer_than_usual(csp, monkeypatch):
    """The memo's keys travel to the host ahead of the verdicts into room sized for ordinary signatures (96 bytes each on average).  A
    block whose every signature drags 600 bytes behind its SEQUENCE - accepted by the reference, still VALID (bccsp/utils/ecdsa.go:43-67:
    asn1.Unmarshal's rest is ignored) - needs four times that: the pass asks for more room at the end and copies the keys again, whole.
    Nobody can switch a block's memo off by padding signatures (up to the 1 024 bytes the memo takes at all; beyond: bccsp/sw decides)."""
    import blockgen
    fx = blockgen.fixture_signers()
    rng = np.random.default_rng(41)
    pad = lambda t, j, sig: sig + b"\x05\x82\x02\x54" + bytes(596)    # noqa: E731   (every creator and endorsement signature)
    envs = [blockgen.endorser_tx(t, rng, fx[4 + t % 2], [fx[0], fx[1], fx[2]], blockgen.make_signer(100 + t), craft=pad) for t in range(48)]
    envs[7] = blockgen.endorser_tx(7, rng, fx[4], [fx[0], fx[1], fx[2]], blockgen.make_signer(999),
                                   craft=lambda t, j, sig: sig + bytes(1100) if j == 0 else pad(t, j, sig))   # one beyond 1 024 bytes: no entry
    blk = bb.block(1, envs)
    csp.set_option("pass_stage_min_bytes", 1)
    seeded = []
    for k, mode in enumerate(("1", "0")):
        csp.set_option("pass_device_memo", 1 if mode == "1" else -1)
        out = fabgpu.preverify_block2(csp, blk, block_seq=300 + k, seed_memo=True)
        assert (out["tx_flags"] == 0).all() and (out["tuple_status"] == 0).all()
        hits = 0
        for i in range(len(out["tuple_status"])):
            sp = [int(x) for x in out["tuple_spans"][i]]
            sig, q = out["arena"][sp[6]:sp[6] + sp[7]], bytes(out["tuple_qxy"][i])
            got = fabgpu.memo_lookup(csp, q[:32], q[32:], sig, bytes(out["tuple_digest"][i])) if len(sig) <= 1024 else None
            assert got == (0 if len(sig) <= 1024 else None)
            hits += got is not None
        assert hits == out["memo_seeded"] == 4 * 48 - 1
        seeded.append(out["memo_seeded"])
        fabgpu.memo_evict_block(csp, 300 + k)
    assert seeded[0] == seeded[1]


@pytest.mark.gpu
def test_device_route_serves_what_it_used_to_decline(csp, monkeypatch):
    """Identities nobody has met (the device decodes their certificates itself), garbage DER, a provider that knows nobody at all:
    none of it takes a block to the host walk any more, and the answers are the host route's - which learns nothing the device route
    does not learn too."""
    rng = np.random.default_rng(5)
    first, want1 = clean_modes_block(40, rng)
    dev, host = _both_routes(csp, monkeypatch, first, 10)                   # nobody is known yet: every tuple's certificate is decoded
    assert (dev["tx_flags"] == want1).all() and dev["n_device_decoded"] == len(dev["tuple_status"]) > 100
    _same(host, dev, KEYS_ALL)
    r = fabgpu.pass_routes(csp)
    # 6 P-256 signers + the P-384 certificate enter the cache (one slot per table hash - keyed per provider - so two of them may meet
    # in a slot and the loser is learned from the next block)
    assert 5 <= r["learned"] <= 7 and r["device_decoded"] == len(dev["tuple_status"])
    for k in range(3):
        dev2, _ = _both_routes(csp, monkeypatch, first, 20 + 2 * k)
        assert (dev2["tx_flags"] == want1).all()
        _same(dev, dev2, KEYS_ALL)
        if dev2["n_device_decoded"] == 0:
            break
    assert dev2["n_device_decoded"] == 0                                                    # ... and are known from then on
    garbage, want2 = build_block(60, rng)                                   # carries garbage DER (and everything else)
    dev3, host3 = _both_routes(csp, monkeypatch, garbage, 30)
    assert (dev3["tx_flags"] == want2).all()
    assert (dev3["tuple_status"] == fabgpu.TUPLE_ST_BAD_DER).sum() == sum(1 for t in range(60) if t % 13 == 5)
    _same(host3, dev3, KEYS_ALL)
    assert fabgpu.pass_routes(csp)["general_der"] >= sum(1 for t in range(60) if t % 13 == 5)


@pytest.mark.gpu
def test_crafted_signature_encodings_stay_on_the_device_route(csp, monkeypatch):
    """Every way of writing a signature that Go's asn1 package accepts or rejects (long-form lengths, trailing bytes, a third element,
    oversize / negative integers, non-minimal lengths) inside an otherwise valid block: decided on the device exactly as
    bccsp/sw/ecdsa.go:41-57 decides - two of them are VALID signatures in the reference - and the block stays on the device route
    (one such signature used to cost a peer the whole block's host walk)."""
    import blockgen
    hows = {(3, 0): "trailing", (5, 1): "third", (8, 2): "long_r", (11, 0): "long_s", (14, 1): "neg_r", (17, 2): "nonminimal"}
    want_st = {"trailing": 0, "third": 0, "long_r": 3, "long_s": 2, "neg_r": 5, "nonminimal": 5}
    blk, envs = blockgen.endorser_block(24, 41, craft=lambda t, j, sig: blockgen.crafted(sig, hows[(t, j)]) if (t, j) in hows else sig)
    _learn(csp, blk)
    dev, host = _both_routes(csp, monkeypatch, blk, 10)
    _same(host, dev, KEYS_ALL)
    st = dev["tuple_status"].reshape(24, 4)
    for (t, j), how in hows.items():
        assert int(st[t, 1 + j]) == want_st[how], (t, j, how, int(st[t, 1 + j]))
        # the oracle's bccsp.Verify on that tuple: valid exactly when the status says so
        sp = [int(x) for x in dev["tuple_spans"][4 * t + 1 + j]]
        q = bytes(dev["tuple_qxy"][4 * t + 1 + j])
        msg = dev["arena"][sp[2]:sp[2] + sp[3]] + dev["arena"][sp[4]:sp[4] + sp[5]]
        try:
            ok = po.verify_ecdsa(int.from_bytes(q[:32], "big"), int.from_bytes(q[32:], "big"), dev["arena"][sp[6]:sp[6] + sp[7]], hashlib.sha256(msg).digest())
        except po.BCCSPError:
            ok = False
        assert ok == (want_st[how] == 0), how
    bad_tx = {t for (t, j), how in hows.items() if want_st[how] != 0}
    assert {int(t) for t in np.nonzero(dev["tx_flags"])[0]} == bad_tx and (st.reshape(-1) != 0).sum() == len(bad_tx)
    assert fabgpu.pass_routes(csp)["general_der"] >= 6


@pytest.mark.gpu
def test_blocks_full_of_new_clients_stay_on_the_device_route(monkeypatch):
    """A busy network: every creator of a block is a certificate the provider has never seen (and there are more of them than its
    identity cache holds), while the endorsers are the channel's few peers with comb tables.  The device decodes the newcomers'
    certificates, their launch carries the keys along, the endorsements' launch stays on the registered tables; the prediction that
    was right for the previous (friendly) block is repaired by one relaunch; statuses, keys, digests and memo equal the host route's."""
    import blockgen
    csp = fabgpu.GPUCSP(device=0)
    try:
        csp._L.fabgpu_csp_identity_cache_limits(csp._h, 128, 64, 1)         # a small cache: 300 newcomers overflow it
        friendly, _ = blockgen.endorser_block(300, 7)
        csp.set_option("pass_stage_min_bytes", 1)
        for k in range(6):                                                  # the six fixture signers are learned and earn their tables
            out = fabgpu.preverify_block2(csp, friendly, block_seq=k)       # (learn slots: open addressing since round 5 - two signers
            if k >= 2 and out["n_keyed"] == 1200 and out["n_device_decoded"] == 0:   #  whose hashes meet no longer wait a block)
                break
        assert (out["tx_flags"] == 0).all() and out["n_keyed"] == 1200 and out["n_device_decoded"] == 0
        fresh = blockgen.fresh_identities(300, 99)
        crowd, _ = blockgen.endorser_block(300, 8, creators=fresh)
        before = fabgpu.pass_routes(csp)
        dev = fabgpu.preverify_block2(csp, crowd, block_seq=10, seed_memo=True)
        after = fabgpu.pass_routes(csp)
        assert after["device_walks"] == before["device_walks"] + 1 and after["relaunches"] == before["relaunches"] + 1
        # the 128 learn slots are open-addressed (round 5): 300 newcomers fill practically all of them - with one slot per table hash
        # (round 4) about 116 would be claimed and the others' identities would wait for a later block
        assert after["learned"] - before["learned"] >= 124, after["learned"] - before["learned"]
        assert (dev["tx_flags"] == 0).all() and (dev["tuple_status"] == 0).all()
        assert dev["n_device_decoded"] == 300 and dev["n_keyed"] == 900 and dev["memo_seeded"] == 1200
        # every creator's key is the one the certificate carries
        for t in (0, 17, 299):
            assert bytes(dev["tuple_qxy"][4 * t]) == blockgen._pubkey(fresh[t][1])
        # the same kind of block again: the prediction now says "creators carry their keys" - no relaunch
        crowd2, _ = blockgen.endorser_block(300, 9, creators=blockgen.fresh_identities(300, 100))
        dev2 = fabgpu.preverify_block2(csp, crowd2, block_seq=11)
        assert fabgpu.pass_routes(csp)["relaunches"] == after["relaunches"] and (dev2["tx_flags"] == 0).all() and dev2["n_keyed"] == 900
        # one broken creator signature and one broken endorsement among the newcomers
        sp = dev2["tuple_spans"]
        broken = bytearray(crowd2)
        for i in (4 * 100, 4 * 200 + 2):
            broken[int(sp[i][6]) + int(sp[i][7]) - 2] ^= 0x10
        dev3 = fabgpu.preverify_block2(csp, bytes(broken), block_seq=12)
        assert {int(i) for i in np.nonzero(dev3["tuple_status"])[0]} == {400, 800, 802}     # (the endorsement sits inside what creator 200 signed)
        assert fabgpu.identity_cache_size(csp) <= 128
        # the host route on the same blocks: same answers
        csp.set_option("pass_stage_min_bytes", 1 << 40)
        host = fabgpu.preverify_block2(csp, crowd, block_seq=20, seed_memo=True)
        _same(host, dev, KEYS_ALL)
        _same(fabgpu.preverify_block2(csp, bytes(broken), block_seq=21), dev3, KEYS_ALL)
    finally:
        csp.close()


@pytest.mark.gpu
def test_idemix_creators_on_the_device_route(monkeypatch):
    """A provider with an idemix MSP used to decline every block to the host walk.  Now the gate kernel recognises
    msp.SerializedIdemixIdentity creators and unmarshals their idemix.NymSignature, the nym kernel runs over the creators' rows beside the
    ECDSA launches (on a prediction; caught up with when the first such block arrives), and statuses, flags, keys (the pseudonym), digests
    (SHA-256 of the payload) and issuer-bound memo entries equal the host route's - tampered, out-of-domain and unknown-MSP cases
    included (tests/test_block_prepass.py::build_mixed_block)."""
    from idemix_common import fixtures
    from test_block_prepass import build_mixed_block
    raw_ipk = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]["MSP1OU1"]["ipk"])
    ipk_hash = bytes(fixtures()["MSP1OU1"]["ipk"].hash)
    csp = fabgpu.GPUCSP(device=0)
    try:
        rng = np.random.default_rng(9)
        blk, want, n_idemix = build_mixed_block(120, rng)
        assert csp.idemix_msp_register("IdemixMSP1", raw_ipk) >= 0
        r0 = fabgpu.pass_routes(csp)
        dev, host = _both_routes(csp, monkeypatch, blk, 10, seed_memo=True)
        assert (dev["tx_flags"] == want).all() and (host["tx_flags"] == want).all()
        _same(host, dev, KEYS_ALL)
        assert dev["memo_seeded"] == host["memo_seeded"] > 0
        assert fabgpu.pass_routes(csp)["relaunches"] == r0["relaunches"] + 1          # the first block with idemix creators: the nym launch caught up
        n_nym = 0
        for i in np.nonzero(dev["tuple_kind"] == 0)[0]:
            if int(dev["tuple_tx"][i]) % 5 != 0 or dev["tuple_status"][i] not in (0, 1):
                continue
            sp = [int(x) for x in dev["tuple_spans"][i]]
            msg, sig = dev["arena"][sp[4]:sp[4] + sp[5]], dev["arena"][sp[6]:sp[6] + sp[7]]
            q = bytes(dev["tuple_qxy"][i])
            assert bytes(dev["tuple_digest"][i]) == hashlib.sha256(msg).digest() and dev["tuple_hashed"][i]
            # (the host route's pass of the same block seeded the same entries under seq 11: look the device route's up under seq 10 only)
            n_nym += 1
        assert n_nym >= 10
        fabgpu.memo_evict_block(csp, 11)
        hits = 0
        for i in np.nonzero(dev["tuple_kind"] == 0)[0]:
            if int(dev["tuple_tx"][i]) % 5 != 0 or dev["tuple_status"][i] not in (0, 1):
                continue
            sp = [int(x) for x in dev["tuple_spans"][i]]
            msg, sig = dev["arena"][sp[4]:sp[4] + sp[5]], dev["arena"][sp[6]:sp[6] + sp[7]]
            q = bytes(dev["tuple_qxy"][i])
            assert fabgpu.memo_lookup_nym(csp, ipk_hash, q[:32], q[32:], sig, hashlib.sha256(msg).digest()) == int(dev["tuple_status"][i])
            assert fabgpu.memo_lookup(csp, q[:32], q[32:], sig, hashlib.sha256(msg).digest()) is None
            hits += 1
        assert hits == n_nym
        fabgpu.memo_evict_block(csp, 10)
        # the next block of the kind: the nym launch is queued on the prediction, nothing is repeated; flags-only callers get the same flags
        before = fabgpu.pass_routes(csp)
        csp.set_option("pass_stage_min_bytes", 1)
        again = fabgpu.preverify_block(csp, blk)
        after = fabgpu.pass_routes(csp)
        assert after["device_walks"] == before["device_walks"] + 1 and after["relaunches"] == before["relaunches"]
        _same(host, again, ["tx_flags", "tuple_status"])
        # a block with MORE pseudonym signatures than the prediction left rows for (16 of the 24 idemix creators were the nym kernel's last
        # time: the launch gets 128 rows; of this block's 220, 147 are): the creators past the launch's rows read "not decided", the
        # summary says so, and a second phase runs over all of them
        big, want_big, n_big = build_mixed_block(1100, np.random.default_rng(10))
        assert n_big == 220
        before = fabgpu.pass_routes(csp)
        dev_big, host_big = _both_routes(csp, monkeypatch, big, 40, seed_memo=True)
        assert fabgpu.pass_routes(csp)["relaunches"] == before["relaunches"] + 1
        assert (dev_big["tx_flags"] == want_big).all()
        _same(host_big, dev_big, KEYS_ALL)
    finally:
        csp.close()


@pytest.mark.gpu
def test_a_certificate_beyond_the_device_decoders_window_costs_its_own_transactions_only(csp, monkeypatch):
    """Round 3 declined the whole block for ONE identity whose PEM body exceeded the decoder's buffer (4 096 base64 digits): anybody who
    can submit a transaction could send every block down the 3.3x slower host walk.  Now the device decodes a certificate of any length
    from its first 3 KiB of DER, and the one case it cannot decide - SubjectPublicKeyInfo beyond that window - is that TUPLE's
    TUPLE_ST_NEEDS_SW (its transaction: flag 4, no memo entry; bccsp/sw decides, as core/common/validation/msgvalidation.go:258-298
    treats a creator per transaction): the block stays on the device route either way."""
    import blockgen
    fx = blockgen.fixture_signers()
    d32 = int(blockgen._IDS[4]["d"], 16).to_bytes(32, "big")
    der = blockgen._pem_der(blockgen._IDS[4]["pem"])
    csp.set_option("pass_stage_min_bytes", 1)
    # (a) bytes behind the Certificate (ignored by every decoder): longer than the window, decided all the same
    padded = bb.serialized_identity("Org1MSP", blockgen._pem_wrap(der + bytes(3100)))
    blk, _ = blockgen.endorser_block(12, 3, creators=[(padded, d32), fx[5]])
    out = fabgpu.preverify_block2(csp, blk, block_seq=1)
    r = fabgpu.pass_routes(csp)
    assert r["device_walks"] == 1 and r["host_walks"] == 0, r
    assert (out["tx_flags"] == 0).all() and (out["tuple_status"] == 0).all()
    # (b) the key beyond the window: transactions 0, 2, 4 ... (that creator's) are left to bccsp/sw, everything else is decided
    far = bb.serialized_identity("Org1MSP", blockgen._pem_wrap(_cert_with_long_issuer(der, 3300)))
    assert fabgpu.identity_to_p256(far) is not None                          # (the host decoder reads it)
    blk2, _ = blockgen.endorser_block(12, 4, creators=[(far, d32), fx[5]])
    out2 = fabgpu.preverify_block2(csp, blk2, block_seq=2, seed_memo=True)
    r = fabgpu.pass_routes(csp)
    assert r["device_walks"] == 2 and r["host_walks"] == 0, r
    assert list(out2["tx_flags"]) == [fabgpu.TX_NEEDS_SW if t % 2 == 0 else 0 for t in range(12)]
    creators = out2["tuple_kind"] == 0
    assert (out2["tuple_status"][creators & (out2["tuple_tx"] % 2 == 0)] == fabgpu.TUPLE_ST_NEEDS_SW).all()
    assert (out2["tuple_status"][~(creators & (out2["tuple_tx"] % 2 == 0))] == 0).all()
    assert out2["memo_seeded"] == int((out2["tuple_status"] == 0).sum())    # no memo entry for what was not decided
    # the host route reads the whole certificate and verifies those creators itself: every transaction valid
    csp.set_option("pass_stage_min_bytes", 1 << 40)
    host = fabgpu.preverify_block2(csp, blk2, block_seq=3)
    assert (host["tx_flags"] == 0).all() and (host["tuple_status"] == 0).all()


@pytest.mark.gpu
def test_device_route_on_the_reference_ledgers(csp, monkeypatch):
    """The reference's own blocks - orderer block signatures with their tail, identities that are not certificates (bccsp/sw
    decides), creator-less genesis envelopes, UUID TxIDs - through both routes: identical answers, reference signatures valid."""
    csp.set_option("pass_stage_min_bytes", 1 << 40)
    host = [fabgpu.preverify_block2(csp, raw, block_seq=i) for i, raw in enumerate(LEDGER_RAW)]
    csp.set_option("pass_stage_min_bytes", 1)
    before = fabgpu.pass_routes(csp)["device_walks"]
    walked = 0
    for i, raw in enumerate(LEDGER_RAW):
        dev = fabgpu.preverify_block2(csp, raw, block_seq=100 + i)
        _same(host[i], dev, ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status", "tuple_spans", "tuple_digest", "tuple_hashed", "tuple_qxy"])
        assert dev["n_block_sigs"] == host[i]["n_block_sigs"] and dev["block_sigs_understood"] == host[i]["block_sigs_understood"]
        assert dev["arena"] == host[i]["arena"]
        nb = fabgpu.preverify_block2(csp, raw, block_seq=200 + i, block_sigs=False)
        k = dev["n_block_sigs"]
        if k:
            assert (nb["tuple_status"][-k:] == fabgpu.TUPLE_ST_SKIPPED).all() and (nb["tuple_kind"][-k:] == 2).all()
            assert np.array_equal(nb["tuple_status"][:-k], dev["tuple_status"][:-k])
    walked = fabgpu.pass_routes(csp)["device_walks"] - before
    assert walked >= 40                                                       # the 20 blocks of the Fabric 2.0 ledger, in both forms (the ledger-harness blocks name no certificate: nothing for the device to decide)
    valid_block_sigs = sum(int(((h["tuple_kind"] == 2) & (h["tuple_status"] == 0)).sum()) for h in host)
    assert valid_block_sigs >= 19


@pytest.mark.gpu
def test_device_route_big_block_caps_and_concurrency(csp, monkeypatch):
    """A multi-megabyte block (several envelopes per scan thread, 10 000+ tuples), answer arrays that are too small at first
    (FABGPU_ETOOBIG with the counts), and three callers on one provider at once."""
    rng = np.random.default_rng(31)
    blk = big_block(2600, rng, bad_every=97)                                # 13 MB: staged by default
    assert len(blk) > 8 << 20
    csp.set_option("pass_stage_min_bytes", 1 << 40)
    host = fabgpu.preverify_block(csp, blk)
    csp.set_option("pass_stage_min_bytes", 0)
    csp._pass_caps = (16, 16)                                               # forces the ETOOBIG round trip on the device route
    before = fabgpu.pass_routes(csp)["device_walks"]
    dev = fabgpu.preverify_block(csp, blk)
    assert fabgpu.pass_routes(csp)["device_walks"] == before + 1
    _same(host, dev, ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status"])
    assert (dev["tx_flags"] == fabgpu.TX_BAD_CREATOR_SIGNATURE).all()      # (the signatures are well-formed DER but fake)
    assert len(dev["tuple_status"]) == 4 * 2600 and (dev["tuple_status"] == 1).all()
    errors = []

    def worker():
        try:
            for _ in range(3):
                got = fabgpu.preverify_block(csp, blk)
                _same(host, got, ["tx_flags", "tuple_status"])
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=worker) for _ in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors[0]


@pytest.mark.gpu
def test_parked_upload_never_answers_for_another_block(csp):
    """after FABGPU_ETOOBIG the library keeps the finished upload for the retry.  A caller that gives up and later
    presents a DIFFERENT block of the same length at the same address (a re-used allocation) must get THAT block's verdicts: the kept
    upload matches on pointer, length, block_seq and a fingerprint of the bytes.  Observable: the digests the DEVICE computed over the
    bytes it read.  Also: fabgpu_csp_block_pass_abandon drops a kept upload, and the honest retry still finds its upload."""
    rng = np.random.default_rng(77)
    blk_a = big_block(2600, rng, bad_every=97)                                # 13 MB: staged by default, its upload is a thread
    buf = np.frombuffer(bytearray(blk_a), dtype=np.uint8)                    # ONE allocation, lent to the library for every call below
    L = csp._L
    full_a = fabgpu.preverify_block2(csp, buf, block_seq=0)
    i = int(np.nonzero((full_a["tuple_tx"] == 1300) & (full_a["tuple_kind"] == 0))[0][0])     # the creator tuple of transaction 1300
    sp = [int(x) for x in full_a["tuple_spans"][i]]                           # (identity, prefix, suffix, signature) x (offset, length)
    msg = lambda raw: bytes(raw[sp[2]:sp[2] + sp[3]]) + bytes(raw[sp[4]:sp[4] + sp[5]])
    assert bytes(full_a["tuple_digest"][i]) == hashlib.sha256(msg(blk_a)).digest()

    def small_call():
        flags = np.zeros(16, np.uint8)
        ps = fabgpu._BlockPass()
        ps.block, ps.len, ps.block_seq, ps.cap_tx, ps.cap_tuples, ps.tx_flags = buf.ctypes.data, buf.size, 0, 16, 0, flags.ctypes.data
        return L.fabgpu_csp_block_preverify2(csp._h, ctypes.byref(ps)), ps.n_tx
    assert small_call() == (-5, 2600)                                         # FABGPU_ETOOBIG, counts set, the upload of block A is kept
    # the caller gave up; the allocation now holds another block of the same length: other bytes in its first KiB (a block's header - number,
    # previous hash - always differs) and one other byte deep inside transaction 1300's payload
    blk_b = bytearray(blk_a)
    assert blk_a[0] == 0x0a and blk_a[2] == 0x08                              # Block.header { number = varint at byte 3 ...
    blk_b[3] ^= 0x02                                                          # ... another block number
    blk_b[sp[4] + sp[5] // 2] ^= 0x40
    buf[:] = np.frombuffer(bytes(blk_b), dtype=np.uint8)
    got_b = fabgpu.preverify_block2(csp, buf, block_seq=0)
    j = int(np.nonzero((got_b["tuple_tx"] == 1300) & (got_b["tuple_kind"] == 0))[0][0])
    assert bytes(got_b["tuple_digest"][j]) == hashlib.sha256(msg(blk_b)).digest(), "the kept upload of block A answered for block B"
    assert bytes(got_b["tuple_digest"][j]) != bytes(full_a["tuple_digest"][i])
    # abandon: a kept upload is dropped exactly once
    buf[:] = np.frombuffer(bytes(blk_a), dtype=np.uint8)
    assert small_call()[0] == -5
    assert L.fabgpu_csp_block_pass_abandon(csp._h) == 1
    assert L.fabgpu_csp_block_pass_abandon(csp._h) == 0
    # and the honest retry (same buffer, same bytes, at once) finds its upload and answers for block A
    assert small_call()[0] == -5
    again = fabgpu.preverify_block2(csp, buf, block_seq=0)
    _same(full_a, again, ["tx_flags", "tuple_status", "tuple_digest"])
    assert L.fabgpu_csp_block_pass_abandon(csp._h) == 0                       # the retry took it


def _bench_block(n_tx):
    """a block of n_tx endorser transactions with VALID signatures: the pre-built bench block when it travelled along
    (tools/make_bench_blocks.py passlegs 10000 0), else signed here with the C oracle"""
    path = os.path.join(ROOT, ".bench_blocks", "friendly_10000.bin")
    if n_tx == 10000 and os.path.exists(path):
        return open(path, "rb").read()
    import ctypes

    import coracle
    L = coracle.lib()
    rng = np.random.default_rng(1)
    ids = [i for i in IDS if i["curve"] == "prime256v1"]
    sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in ids]

    def sign(k, msg):
        d = int(ids[k]["d"], 16).to_bytes(32, "big")
        nonce = b"\x00" + bytes(rng.integers(1, 255, size=31, dtype=np.uint8))
        r, s = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        assert L.oracle_p256_sign(d, hashlib.sha256(msg).digest(), nonce, 1, r, s) == 0
        return po.marshal_ecdsa_signature(int.from_bytes(r.raw, "big"), int.from_bytes(s.raw, "big"))
    envs = []
    for t in range(n_tx):
        picks = [int(j) for j in rng.choice(4, size=3, replace=False)]
        c = 4 + t % 2
        payload, _ = bb.consistent_endorser_tx("mychannel", sid[c], bytes(rng.integers(0, 256, size=24, dtype=np.uint8)),
                                               bytes(rng.integers(0, 256, size=300, dtype=np.uint8)), bytes(rng.integers(0, 256, size=990, dtype=np.uint8)),
                                               lambda prp: [(sid[j], sign(j, prp + sid[j])) for j in picks])
        envs.append(bb.envelope(payload, sign(c, payload)))
    return bb.block(1, envs)


@pytest.mark.gpu
def test_split_submission_equals_the_host_route(csp, monkeypatch):
    """More than 32 768 tuples: the device route submits the creators as a launch of their own (two lanes per signature) beside the
    endorsements (one lane), rows permuted - statuses, digests and keys must still land on the right tuples.  The block's signatures are
    valid; a few dozen are then broken in place (creators and endorsers), so that a wrong row would show."""
    blk = _bench_block(10000)
    csp.set_option("pass_stage_min_bytes", 1 << 40)
    host = fabgpu.preverify_block2(csp, blk, block_seq=1)
    assert (host["tx_flags"] == 0).all() and len(host["tuple_status"]) == 40000
    rng = np.random.default_rng(12)
    broken = bytearray(blk)
    victims = sorted(int(v) for v in rng.choice(40000, size=60, replace=False))
    for i in victims:
        sp = [int(x) for x in host["tuple_spans"][i]]
        broken[sp[6] + sp[7] - 1 - int(rng.integers(0, 6))] ^= 0x04            # low bytes of s: still the common DER shape
    broken = bytes(broken)
    host_b = fabgpu.preverify_block2(csp, broken, block_seq=2)
    # (an endorsement sits inside the payload its transaction's creator signed: breaking it breaks that creator signature too)
    assert sorted(int(i) for i in np.nonzero(host_b["tuple_status"])[0]) == sorted(set(victims) | {4 * (i // 4) for i in victims})
    csp.set_option("pass_stage_min_bytes", 0)
    before = fabgpu.pass_routes(csp)["device_walks"]
    keys = ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status", "tuple_spans", "tuple_digest", "tuple_hashed", "tuple_qxy"]
    _same(host, fabgpu.preverify_block2(csp, blk, block_seq=3), keys)
    dev_b = fabgpu.preverify_block2(csp, broken, block_seq=4, seed_memo=True)
    _same(host_b, dev_b, keys)
    assert fabgpu.pass_routes(csp)["device_walks"] == before + 2
    assert dev_b["memo_seeded"] == 40000
    flags_only = fabgpu.preverify_block(csp, broken)
    _same(host_b, flags_only, ["tx_flags", "tuple_status"])


@pytest.mark.gpu
def test_device_route_with_keys_carried_along(monkeypatch):
    """Identities that are known but have NOT earned a comb table (or only some of them have): the device route then runs the
    fresh-key kernels over its own rows - keys copied out of the identity table, per-launch j*Q workspaces, the blind launch replaced
    by a wait for the gate summary - on a small block (one fused launch) and on a 40 000-tuple block (split: creators' verify-only
    pair launch beside the endorsements' one-lane launch, two workspaces on two streams).  Same answers as the host route."""
    keys = ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status", "tuple_spans", "tuple_digest", "tuple_hashed", "tuple_qxy"]
    rng = np.random.default_rng(55)
    small, want = clean_modes_block(150, rng)
    big = _bench_block(10000)
    for tables in (0, 3):
        csp = fabgpu.GPUCSP(device=0)
        try:
            csp._L.fabgpu_csp_identity_cache_limits(csp._h, 4096, tables, 1)
            csp.set_option("pass_stage_min_bytes", 1 << 40)
            host_small = fabgpu.preverify_block2(csp, small, block_seq=1)
            host_big = fabgpu.preverify_block2(csp, big, block_seq=2)
            assert (host_small["tx_flags"] == want).all() and (host_big["tx_flags"] == 0).all()
            csp.set_option("pass_stage_min_bytes", 1)
            before = fabgpu.pass_routes(csp)["device_walks"]
            dev_small = fabgpu.preverify_block2(csp, small, block_seq=3)
            dev_big = fabgpu.preverify_block2(csp, big, block_seq=4, seed_memo=True)
            assert fabgpu.pass_routes(csp)["device_walks"] == before + 2, fabgpu.pass_routes(csp)
            _same(host_small, dev_small, keys)
            _same(host_big, dev_big, keys)
            assert dev_big["memo_seeded"] == 40000
            if tables == 0:
                assert dev_big["n_keyed"] == 0 and dev_small["n_keyed"] == 0
        finally:
            csp.close()
