"""Ground truth for the BIG shapes of the device route: the 10 000-transaction blocks of BASELINE's second
metric - friendly, every creator a certificate nobody has met, every fifth creator an idemix pseudonym - and a 16 500-transaction block
(66 000 tuples: beyond the split submission - one launch, one lane per signature), each with 1 % of its transactions replaced by
CORRUPTED ones.  tx_flags are asserted against what the generator put in - not against the other route: until round 3 the big shapes
were covered by route equality and by bench.py only.

The blocks are the ones tools/make_bench_blocks.py pre-builds (.bench_blocks/, git-ignored, travels with the snapshot); without them they
are signed here (CPU oracle, tens of seconds each).  Transaction t with t % 100 == 37 is replaced by one signed by the fixture signers
with, by (t // 100) % 6:
  0  a creator signature over other bytes                  -> TX_BAD_CREATOR_SIGNATURE (1)
  1  the second endorsement's signature with its last byte flipped (the creator signed the payload as it stands) -> TX_BAD_ENDORSEMENT (2)
  2  the second endorsement's signature in high-S form     -> TX_BAD_ENDORSEMENT (2)
  3  a TxID that is not hex(SHA-256(nonce || creator))     -> TX_BAD_TXID (5)
  4  a proposal hash that is not the recomputed one        -> TX_BAD_PROPOSAL_HASH (6)
  5  a creator signature that does not unmarshal           -> TX_BAD_CREATOR_SIGNATURE (1)
(the reference's order of rejection: core/common/validation/msgvalidation.go:248-320, then VSCC)."""
import json
import os
import sys

import numpy as np
import pytest

import bccsp_sw_oracle as po
import blockbuilder as bb
import fabgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, ".bench_blocks")
WANT_OF_MODE = {0: fabgpu.TX_BAD_CREATOR_SIGNATURE, 1: fabgpu.TX_BAD_ENDORSEMENT, 2: fabgpu.TX_BAD_ENDORSEMENT, 3: fabgpu.TX_BAD_TXID,
                4: fabgpu.TX_BAD_PROPOSAL_HASH, 5: fabgpu.TX_BAD_CREATOR_SIGNATURE}


def _block(name, build):
    p = os.path.join(CACHE, name)
    if os.path.exists(p):
        return open(p, "rb").read()
    return build()


def corrupted_envelope(t, mode, rng, sign, fx):
    cid, cd = fx[4 + t % 2]
    picks = [fx[int(j)] for j in rng.choice(4, size=3, replace=False)]

    def ends(prp):
        out = []
        for j, (eid, ed) in enumerate(picks):
            sig = sign(ed, prp + eid)
            if j == 1 and mode == 1:
                sig = sig[:-1] + bytes([sig[-1] ^ 1])
            if j == 1 and mode == 2:
                r, s = po.unmarshal_ecdsa_signature(sig)
                sig = po.marshal_ecdsa_signature(r, po.N - s)
            out.append((eid, sig))
        return out
    payload, _ = bb.consistent_endorser_tx("mychannel", cid, bytes(rng.integers(0, 256, size=24, dtype=np.uint8)), bytes(rng.integers(0, 256, size=300, dtype=np.uint8)),
                                           bytes(rng.integers(0, 256, size=990, dtype=np.uint8)), ends, bad_txid=(mode == 3), bad_phash=(mode == 4))
    csig = sign(cd, payload + (b"!" if mode == 0 else b""))
    if mode == 5:
        csig = b"\x30\x81" + csig[1:]                                          # a long-form length below 128: asn1 refuses it
    return bb.envelope(payload, csig)


def with_one_percent_corrupted(envs, number=7, seed=404):
    """-> (block bytes, want flags): every transaction t with t % 100 == 37 replaced by a corrupted one"""
    import blockgen
    fx = blockgen.fixture_signers()
    sign = blockgen.make_signer(seed)
    rng = np.random.default_rng(seed + 1)
    envs = list(envs)
    want = np.zeros(len(envs), dtype=np.uint8)
    for t in range(37, len(envs), 100):
        mode = (t // 100) % 6
        envs[t] = corrupted_envelope(t, mode, rng, sign, fx)
        want[t] = WANT_OF_MODE[mode]
    return bb.block(number, envs), want


def test_the_corruption_generator_against_the_oracle():
    """(CPU) what the big-block tests call ground truth, checked on a small block by the pure-Python oracle: every tuple of a corrupted
    transaction verifies or fails exactly as its mode says, the hash checks of modes 3 / 4 fail, everything else is valid."""
    import hashlib

    import blockgen
    blk0, envs = blockgen.endorser_block(640, 12)
    blk, want = with_one_percent_corrupted(envs)
    assert sorted(np.nonzero(want)[0].tolist()) == list(range(37, 640, 100)) and set(want[want != 0]) == {1, 2, 5, 6}
    tuples, arena = fabgpu.block_tuples(blk)
    bad_tuple = {}
    for tp in tuples:
        tx = tp["tx"]
        if tx % 100 != 37:
            continue
        cut = lambda sp: arena[sp[0]:sp[0] + sp[1]]                              # noqa: E731
        ident, msg, sig = cut(tp["identity"]), cut(tp["prefix"]) + cut(tp["suffix"]), cut(tp["sig"])
        q = fabgpu.identity_to_p256(ident)
        try:
            ok = po.verify_ecdsa(int.from_bytes(q[:32], "big"), int.from_bytes(q[32:], "big"), sig, hashlib.sha256(msg).digest())
        except Exception:                                                       # noqa: BLE001  (bccsp/sw's (false, err))
            ok = False
        if not ok:
            bad_tuple.setdefault(tx, []).append(tp["kind"])
    assert bad_tuple == {37: [0], 137: [1], 237: [1], 537: [0], 637: [0]}
    bad_hash = {}
    for tx, kind, pieces, expect in fabgpu.block_hash_checks(blk):
        if tx % 100 != 37:
            continue
        d = hashlib.sha256(b"".join(blk[a:b] for a, b in pieces)).digest()
        exp = blk[expect[0]:expect[1]]
        if (d.hex().encode() if kind == 0 else d) != exp:
            bad_hash[tx] = kind
    assert bad_hash == {337: 0, 437: 1}


def _passes(csp, blk, want, n, seq0=0, **expect):
    csp.set_option("pass_stage_min_bytes", 1)
    before = fabgpu.pass_routes(csp)
    out = None
    for k in range(n):
        out = fabgpu.preverify_block2(csp, blk, block_seq=seq0 + k, lean=True)
        flags = np.asarray(out["tx_flags"])
        bad = np.nonzero(flags != want)[0]
        assert len(bad) == 0, "pass %d: transactions %r flagged %r, want %r" % (k, bad[:8].tolist(), flags[bad[:8]].tolist(), want[bad[:8]].tolist())
    after = fabgpu.pass_routes(csp)
    assert after["host_walks"] == before["host_walks"] and after["device_walks"] == before["device_walks"] + n, (before, after)
    for k, v in expect.items():
        assert out[k] == v, (k, out[k], v)
    return out


@pytest.mark.gpu
def test_friendly_block_of_10000_transactions_with_one_percent_corrupted():
    import blockgen
    blk = _block("friendly_10000.bin", lambda: blockgen.endorser_block(10000, 1)[0])
    _, envs = blockgen.split_envelopes(blk)
    bad, want = with_one_percent_corrupted(envs)
    assert int((want != 0).sum()) == 100
    csp = fabgpu.GPUCSP(device=0)
    try:
        _passes(csp, blk, np.zeros(10000, np.uint8), 4, n_tuples=40000, n_keyed=40000)    # the clean block: six signers earn their tables
        _passes(csp, bad, want, 2, seq0=10, n_tuples=40000, n_device_decoded=0)
        # with memo seeding: one entry per tuple the device hashed and decided - rejects included (their status is the reference's reject) -
        # and none for the signatures that were refused before any arithmetic (mode 2: high-S; mode 5: does not unmarshal)
        out = fabgpu.preverify_block2(csp, bad, block_seq=50, seed_memo=True, lean=True)
        assert (np.asarray(out["tx_flags"]) == want).all()
        n_refused = sum(1 for t in range(37, 10000, 100) if (t // 100) % 6 in (2, 5))
        assert out["memo_seeded"] == 40000 - n_refused, (out["memo_seeded"], n_refused)
        assert fabgpu.pass_routes(csp)["host_walks"] == 0
    finally:
        csp.close()


@pytest.mark.gpu
def test_10000_distinct_creator_certificates_with_one_percent_corrupted():
    import blockgen
    blk = _block("distinct_10000.bin", lambda: blockgen.endorser_block(10000, 3, creators=blockgen.fresh_identities(10000, 4))[0])
    _, envs = blockgen.split_envelopes(blk)
    bad, want = with_one_percent_corrupted(envs)
    csp = fabgpu.GPUCSP(device=0)
    try:
        csp._L.fabgpu_csp_identity_cache_limits(csp._h, 256, 256, 64)           # a cache too small to remember them: decoded on every pass
        out = _passes(csp, bad, want, 3, n_tuples=40000)
        assert out["n_device_decoded"] >= 9000
    finally:
        csp.close()


@pytest.mark.gpu
def test_10000_transactions_every_fifth_creator_an_idemix_pseudonym_with_one_percent_corrupted():
    import blockgen
    sys.path.insert(0, os.path.join(ROOT, "tools"))

    def build():
        import make_bench_blocks
        return make_bench_blocks.mixed_block("idemix", 10000, 5)
    blk = _block("idemix_10000_5.bin", build)
    _, envs = blockgen.split_envelopes(blk)
    bad, want = with_one_percent_corrupted(envs)
    # ... and three pseudonym signatures broken in place (the last byte of the nonce): "zero-knowledge proof is invalid" -> TX_BAD_CREATOR_SIGNATURE
    tuples, _ = fabgpu.block_tuples(bad)
    b = bytearray(bad)
    for tp in tuples:
        if tp["kind"] == 0 and tp["tx"] in (5, 4005, 9995):
            b[tp["sig"][0] + tp["sig"][1] - 1] ^= 1
            want[tp["tx"]] = fabgpu.TX_BAD_CREATOR_SIGNATURE
    raw_ipk = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]["MSP1OU1"]["ipk"])
    csp = fabgpu.GPUCSP(device=0)
    try:
        assert csp.idemix_msp_register("IdemixMSP1", raw_ipk) >= 0
        _passes(csp, bytes(b), want, 4, n_tuples=40000)
    finally:
        csp.close()


@pytest.mark.gpu
def test_16500_transactions_beyond_the_split_submission_with_one_percent_corrupted():
    """66 000 tuples: more than the creators-on-two-lanes + everybody-else-on-one split holds (65 536 lanes) - one launch, one lane per
    signature - and more than 2^16 of anything a 16-bit index could count."""
    import blockgen
    blk = _block("friendly_10000.bin", lambda: blockgen.endorser_block(10000, 1)[0])
    _, envs = blockgen.split_envelopes(blk)
    bad, want = with_one_percent_corrupted(envs + envs[:6500])
    assert len(want) == 16500 and int((want != 0).sum()) == 165
    csp = fabgpu.GPUCSP(device=0)
    try:
        _passes(csp, bad, want, 4, n_tuples=66000)
    finally:
        csp.close()
