"""The digest memo: bccsp.Hash answered from what a block pass already hashed on the device (include/fabgpu_bccsp.h
fabgpu_csp_hash_lookup; reference call site msp/identities.go:173-181 -> bccsp/sw/impl.go:177-194 -> bccsp/sw/hash.go:29-33).

The contract under test: a stored digest is handed out ONLY for a message whose every byte equals the bytes the device hashed - never on
the fingerprint that chooses where to look.  Ground truth is hashlib (= Go's crypto/sha256, which the reference's own test pins against
the standard library for 100 lengths: bccsp/sw/impl_test.go:1293-1339).

CPU tests: the fingerprint's restatement (the lines the device and the host share) and its independence of how a message is split.
GPU tests: hits equal hashlib on synthetic blocks with every corruption and on the reference's 74 ledger blocks; one flipped byte anywhere,
another length, an evicted block, a message the device did not hash, a switched-off memo: miss; equal fingerprints with different middles:
each message gets ITS digest; the host-built index equals the device-built one; the pool of kept block copies is bounded."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import blockbuilder as bb
import fabgpu
from test_block_prepass import IDS
from test_device_walk import LEDGER_RAW, clean_modes_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1


def fingerprint(msg: bytes) -> int:
    """walk::msg_fingerprint (fabric-mod_amd/csrc/block_walk_core.h) restated: the length and eight 8-byte little-endian samples spread
    evenly from the first to the last byte."""
    n = len(msg)
    h = ((n + 1) * 0x9E3779B97F4A7C15) & M64
    if n < 8:
        for b in msg:
            h = ((h ^ b) * 0x100000001B3) & M64
        return h ^ (h >> 32)
    for k in range(8):
        p = (n - 8) * k // 7
        w = int.from_bytes(msg[p:p + 8], "little")
        h = ((h ^ w) * 0xD6E8FEB86659FD93) & M64
        h ^= h >> 29
    return h ^ (h >> 32)


def sampled_positions(n: int):
    return {(n - 8) * k // 7 + j for k in range(8) for j in range(8)} if n >= 8 else set(range(n))


def _hosttest():
    path = os.path.join(ROOT, "fabric-mod_amd", "lib", "libfabgpu_hosttest.so")
    L = ctypes.CDLL(path)
    L.fabgpu_hosttest_msg_fingerprint.restype = ctypes.c_uint64
    L.fabgpu_hosttest_msg_fingerprint.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32]
    return L


def test_fingerprint_restated_and_split_invariant():
    """The device computes the fingerprint over a tuple's two spans (prp, endorser), the host over the caller's contiguous bytes: the value
    must not depend on where the message is cut.  Checked against the restatement above for every length 0..200 and a few long ones."""
    L = _hosttest()
    rng = np.random.default_rng(5)
    for n in list(range(0, 201)) + [1855, 1856, 1857, 4608, 65537]:
        msg = bytes(rng.integers(0, 256, size=n, dtype=np.uint8))
        want = fingerprint(msg)
        for cut in {0, n, n // 2, min(n, 7), min(n, 8), max(0, n - 8), max(0, n - 1)}:
            assert L.fabgpu_hosttest_msg_fingerprint(msg[:cut], cut, msg[cut:], n - cut) == want, (n, cut)


def test_fingerprint_sees_only_its_samples():
    """What the fingerprint does NOT see is the point of the byte comparison behind it: a byte outside the eight windows leaves it unchanged."""
    rng = np.random.default_rng(6)
    msg = bytearray(rng.integers(0, 256, size=1856, dtype=np.uint8).tobytes())
    inside = sampled_positions(len(msg))
    outside = [p for p in range(len(msg)) if p not in inside]
    assert len(inside) == 64 and len(outside) == 1856 - 64
    base = fingerprint(bytes(msg))
    m2 = bytearray(msg); m2[outside[700]] ^= 1
    assert fingerprint(bytes(m2)) == base
    m3 = bytearray(msg); m3[sorted(inside)[20]] ^= 1
    assert fingerprint(bytes(m3)) != base


# ---- GPU -------------------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def csp():
    c = fabgpu.GPUCSP(device=0)
    yield c
    c.close()


def _messages(out):
    """(tuple index, signed message bytes) of every tuple of a pass answer (virtual arena: block || padding || tail)"""
    for i in range(len(out["tuple_status"])):
        sp = [int(x) for x in out["tuple_spans"][i]]
        yield i, out["arena"][sp[2]:sp[2] + sp[3]] + out["arena"][sp[4]:sp[4] + sp[5]]


def _check_block(csp, out, min_hits=1):
    """every message the device hashed and decided is answered with hashlib's digest; every other message misses"""
    hits = 0
    for i, msg in _messages(out):
        got = fabgpu.hash_lookup(csp, msg)
        live = bool(out["tuple_hashed"][i]) and int(out["tuple_status"][i]) <= 3 and 0 < int(out["tuple_spans"][i][7]) <= 1024 and len(msg) >= 64
        if live:
            assert got == hashlib.sha256(msg).digest(), i
            assert got == bytes(out["tuple_digest"][i])
            hits += 1
        elif got is not None:
            # a message that ALSO belongs to a live tuple (the same payload twice in a block) may be answered - correctly
            assert got == hashlib.sha256(msg).digest(), i
    assert hits >= min_hits
    return hits


@pytest.mark.gpu
def test_hash_is_answered_only_on_full_byte_equality(csp):
    rng = np.random.default_rng(201)
    blk, want = clean_modes_block(220, rng)
    fabgpu.preverify_block(csp, blk)                                         # first sight: identities learned
    out = fabgpu.preverify_block2(csp, blk, block_seq=11, seed_memo=True)
    assert fabgpu.pass_routes(csp)["device_walks"] >= 1 and (out["tx_flags"] == want).all()
    hits = _check_block(csp, out, min_hits=600)
    assert hits == out["memo_seeded"]
    st0 = fabgpu.hash_memo_stats(csp)
    assert st0["hits"] >= hits and st0["blocks_held"] == 1 and st0["bytes_held"] >= len(blk)
    # one flipped byte ANYWHERE => miss: every position of one creator message and one endorsement message ...
    live = [(i, m) for i, m in _messages(out) if out["tuple_hashed"][i] and out["tuple_status"][i] <= 3]
    creator = next(m for i, m in live if out["tuple_kind"][i] == 0)
    endorse = next(m for i, m in live if out["tuple_kind"][i] == 1)
    for msg in (creator, endorse):
        assert fabgpu.hash_lookup(csp, msg) == hashlib.sha256(msg).digest()
        for pos in range(len(msg)):
            m = bytearray(msg)
            m[pos] ^= 1 << (pos % 8)
            assert fabgpu.hash_lookup(csp, bytes(m)) is None, pos
        # ... another length (one byte less / more at either end), the two halves swapped, the empty and the short message
        for m in (msg[:-1], msg[1:], msg + b"\0", b"\0" + msg, msg[len(msg) // 2:] + msg[:len(msg) // 2], b"", msg[:63]):
            assert fabgpu.hash_lookup(csp, m) is None
    # ... and a random sample of single-bit flips over every live message of the block
    for i, msg in live[::7]:
        m = bytearray(msg)
        m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        assert fabgpu.hash_lookup(csp, bytes(m)) is None
    # the verdict memo is keyed on the digest the lookup hands out: Hash then Verify, as identity.Verify does (msp/identities.go:173-188)
    for i, msg in live[::5]:
        d = fabgpu.hash_lookup(csp, msg)
        sp = [int(x) for x in out["tuple_spans"][i]]
        q = bytes(out["tuple_qxy"][i])
        assert fabgpu.memo_lookup(csp, q[:32], q[32:], out["arena"][sp[6]:sp[6] + sp[7]], d) == int(out["tuple_status"][i])
    # evicted block => miss, and its host copy is back in the pool
    assert fabgpu.memo_evict_block(csp, 11) == out["memo_seeded"]
    assert fabgpu.hash_lookup(csp, creator) is None and fabgpu.hash_lookup(csp, endorse) is None
    assert fabgpu.hash_memo_stats(csp)["blocks_held"] == 0


def _twin_block(rng):
    """Two transactions whose signed messages have EQUAL length and EQUAL fingerprints and differ in one byte the fingerprint does not sample
    (creator payloads, and every endorsement message through the shared prp).  Signatures are well-formed but fake: the device hashes and
    decides (status 1), which is all the digest memo needs."""
    sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS if i["curve"] == "prime256v1"]
    fake = b"\x30\x44\x02\x20" + b"\x11" * 32 + b"\x02\x20" + b"\x22" * 32
    nonce, ccpp = bytes(range(24)), bytes(rng.integers(0, 256, size=300, dtype=np.uint8))
    ext = bytearray(rng.integers(0, 256, size=900, dtype=np.uint8).tobytes())
    for at in range(100, 800):
        ext2 = bytearray(ext)
        ext2[at] ^= 0x5A
        pay, prps = [], []
        for e in (ext, ext2):
            p, prp = bb.consistent_endorser_tx("mychannel", sid[4], nonce, ccpp, bytes(e), lambda prp: [(sid[j], fake) for j in (0, 1, 2)])
            pay.append(p)
            prps.append(prp)
        assert len(pay[0]) == len(pay[1]) and len(prps[0]) == len(prps[1])
        dp = [k for k in range(len(pay[0])) if pay[0][k] != pay[1][k]]
        de = [k for k in range(len(prps[0])) if prps[0][k] != prps[1][k]]
        if len(dp) == 1 and len(de) == 1 and dp[0] not in sampled_positions(len(pay[0])) and de[0] not in sampled_positions(len(prps[0]) + len(sid[0])):
            return bb.block(5, [bb.envelope(pay[0], fake), bb.envelope(pay[1], fake)]), pay, [[prps[t] + sid[j] for j in (0, 1, 2)] for t in (0, 1)]
    raise AssertionError("no position outside the sampled windows found")


@pytest.mark.gpu
def test_equal_fingerprints_different_middles(csp):
    rng = np.random.default_rng(202)
    blk, pay, ends = _twin_block(rng)
    assert fingerprint(pay[0]) == fingerprint(pay[1]) and pay[0] != pay[1]
    for j in range(3):
        assert fingerprint(ends[0][j]) == fingerprint(ends[1][j]) and ends[0][j] != ends[1][j]
    fabgpu.preverify_block(csp, blk)
    out = fabgpu.preverify_block2(csp, blk, block_seq=21, seed_memo=True)
    assert out["memo_seeded"] == 8
    for msg in pay + ends[0] + ends[1]:                                       # each message gets ITS OWN digest, whichever of the twins is met first
        assert fabgpu.hash_lookup(csp, msg) == hashlib.sha256(msg).digest()
    # a third message with the same fingerprint that is in NO block: miss
    third = bytearray(pay[0])
    d = next(k for k in range(len(pay[0])) if pay[0][k] != pay[1][k])
    third[d] ^= 0xFF
    assert fingerprint(bytes(third)) == fingerprint(pay[0]) and fabgpu.hash_lookup(csp, bytes(third)) is None


@pytest.mark.gpu
def test_digest_memo_on_the_reference_ledgers(csp):
    """The reference's own blocks (tests/golden/ledger_blocks.json: 74 blocks of its sample ledgers): every message the device hashed -
    creator payloads, prp || endorser, and the orderers' block-signature messages, which are NOT in the block (the walker builds them:
    Metadata.value || signature_header || BlockHeaderBytes) - is answered with hashlib's digest."""
    total = 0
    for i, raw in enumerate(LEDGER_RAW):
        out = fabgpu.preverify_block2(csp, raw, block_seq=1000 + i, seed_memo=True)
        if out["memo_seeded"]:
            total += _check_block(csp, out)
            kinds = {int(out["tuple_kind"][k]) for k in range(len(out["tuple_status"])) if out["tuple_hashed"][k]}
            if 2 in kinds:                                                   # an orderer's signature message lives in the tail
                k = next(k for k in range(len(out["tuple_status"])) if out["tuple_kind"][k] == 2 and out["tuple_hashed"][k])
                sp = [int(x) for x in out["tuple_spans"][k]]
                assert sp[4] >= len(raw)
        fabgpu.memo_evict_block(csp, 1000 + i)
    assert total >= 60                                                       # the Fabric 2.0 ledger's reference-produced signatures


@pytest.mark.gpu
def test_index_built_on_the_host_equals_the_index_built_on_the_device(csp):
    """Three ways a block gets its digest memo: the device route with the device-built table, the device route with the table seeded on the
    host (pass_device_memo off), the host walk of a staged block (pass_device_walk off).  Same hits, same misses.  A block that was not
    staged (no upload ahead, so no host copy) has a verdict memo and no digest memo."""
    rng = np.random.default_rng(203)
    blk, want = clean_modes_block(120, rng)
    fabgpu.preverify_block(csp, blk)
    answers = []
    for seq, (opt, val) in enumerate([("pass_device_memo", 1), ("pass_device_memo", -1), ("pass_device_walk", -1)]):
        prev = csp.set_option(opt, val)
        out = fabgpu.preverify_block2(csp, blk, block_seq=30 + seq, seed_memo=True)
        csp.set_option(opt, prev)
        assert (out["tx_flags"] == want).all() and out["memo_seeded"] > 300
        got = [fabgpu.hash_lookup(csp, msg) for _, msg in _messages(out)]
        assert _check_block(csp, out) == out["memo_seeded"]
        fabgpu.memo_evict_block(csp, 30 + seq)
        answers.append(got)
    assert answers[0] == answers[1] == answers[2]
    csp.set_option("pass_stage_min_bytes", 1 << 40)                           # not staged: the block rides with the submission
    out = fabgpu.preverify_block2(csp, blk, block_seq=40, seed_memo=True)
    assert out["memo_seeded"] > 300
    assert all(fabgpu.hash_lookup(csp, msg) is None for _, msg in _messages(out))
    sp = [int(x) for x in out["tuple_spans"][0]]
    q = bytes(out["tuple_qxy"][0])
    assert fabgpu.memo_lookup(csp, q[:32], q[32:], out["arena"][sp[6]:sp[6] + sp[7]], bytes(out["tuple_digest"][0])) == int(out["tuple_status"][0])


@pytest.mark.gpu
def test_switched_off_and_bounded():
    """pass_hash_memo < 0: no copy is kept, every lookup misses, the verdict memo is untouched.  hash_memo_blocks bounds the pinned host
    copies: a block beyond the bound has no digest memo (counted as refused) until an earlier block is evicted."""
    rng = np.random.default_rng(204)
    blk, _ = clean_modes_block(60, rng)
    c = fabgpu.GPUCSP(device=0, devices=[0], pass_hash_memo=-1)
    try:
        fabgpu.preverify_block(c, blk)
        out = fabgpu.preverify_block2(c, blk, block_seq=1, seed_memo=True)
        assert out["memo_seeded"] > 100 and all(fabgpu.hash_lookup(c, m) is None for _, m in _messages(out))
        assert fabgpu.hash_memo_stats(c)["blocks_held"] == 0
        c.set_option("pass_hash_memo", 1)                                     # ... and back on, on the living provider
        out = fabgpu.preverify_block2(c, blk, block_seq=2, seed_memo=True)
        assert _check_block(c, out) == out["memo_seeded"]
    finally:
        c.close()
    c = fabgpu.GPUCSP(device=0, devices=[0], hash_memo_blocks=2)
    try:
        fabgpu.preverify_block(c, blk)
        outs = [fabgpu.preverify_block2(c, blk, block_seq=10 + k, seed_memo=True) for k in range(3)]
        st = fabgpu.hash_memo_stats(c)
        assert st["blocks_held"] == 2 and st["refused"] == 1
        assert all(o["memo_seeded"] == outs[0]["memo_seeded"] for o in outs)  # the verdict memo does not care
        assert fabgpu.memo_has_block(c, 12) == outs[2]["memo_seeded"]
        fabgpu.memo_evict_block(c, 10)
        fabgpu.memo_evict_block(c, 11)
        assert fabgpu.hash_memo_stats(c)["blocks_held"] == 0
        # block 12 was passed while the pool was exhausted: its messages miss; a pass after the evictions has its copy again
        _, some = next(iter((i, m) for i, m in _messages(outs[2]) if outs[2]["tuple_hashed"][i] and outs[2]["tuple_status"][i] <= 3))
        assert fabgpu.hash_lookup(c, some) is None
        out = fabgpu.preverify_block2(c, blk, block_seq=13, seed_memo=True)
        assert fabgpu.hash_lookup(c, some) == hashlib.sha256(some).digest()
        assert fabgpu.hash_memo_stats(c)["blocks_held"] == 1
    finally:
        c.close()


@pytest.mark.gpu
def test_a_big_block_through_the_threaded_upload(csp):
    """Blocks of 4 MiB and more travel through the pinned buffer in pieces copied by several threads - the buffer the digest memo then
    keeps.  A 13 MB block: a sample of its 10 400 messages, a flipped byte in each, and lookups from four threads at once while the next
    block's pass runs (what a channel's validators and its arrival hook do)."""
    import threading
    from test_device_walk import big_block
    rng = np.random.default_rng(205)
    blk = big_block(2600, rng)
    assert len(blk) > 8 << 20
    fabgpu.preverify_block(csp, blk)
    out = fabgpu.preverify_block2(csp, blk, block_seq=50, seed_memo=True)
    msgs = [m for i, m in _messages(out) if out["tuple_hashed"][i]][::13]
    assert len(msgs) >= 700
    errors = []

    def validators(part):
        try:
            for m in part:
                assert fabgpu.hash_lookup(csp, m) == hashlib.sha256(m).digest()
                bad = bytearray(m)
                bad[len(bad) // 3] ^= 4
                assert fabgpu.hash_lookup(csp, bytes(bad)) is None
        except Exception as e:   # noqa: BLE001
            errors.append(e)
    th = [threading.Thread(target=validators, args=(msgs[k::4],)) for k in range(4)]
    for t in th:
        t.start()
    nxt = fabgpu.preverify_block2(csp, blk, block_seq=51, seed_memo=True)    # the next block arrives meanwhile
    for t in th:
        t.join(timeout=300)
    assert not errors, errors[0]
    assert nxt["memo_seeded"] == out["memo_seeded"]
    assert fabgpu.hash_memo_stats(csp)["blocks_held"] == 2
