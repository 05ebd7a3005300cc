"""The reference's OWN ledgers as golden vectors for the whole path.

tests/golden/ledger_blocks.json holds the 74 blocks of the sample ledgers the reference ships for its upgrade tests
(core/ledger/kvledger/tests/testdata/{v11,v13_statecouchdb,v20}/sample_ledgers*/ledgersData.zip), re-marshalled as common.Block by
tests/golden/gen_ledger_block_fixtures.py.  The v20 ledger was written by a real Fabric 2.0 network: its 15 endorser transactions and 5
config transactions carry 20 creator signatures, 21 endorsement signatures, 15 TxIDs, 15 proposal hashes and 19 orderer block
signatures — (certificate, message, DER signature) triples produced by the reference's own signing path and accepted by its own
validators (TRANSACTIONS_FILTER all VALID).  The v11 / v13 ledgers come from ledger-only test harnesses: identities are not
certificates, TxIDs are UUIDs — so they pin the "leave it to Go" and "TxID does not match" answers, and 84 more proposal hashes.

CPU tests: the C++ walker's tuples equal an independent Python decoder's (tests/fabric_decode.py) byte for byte, and the CPU oracle
accepts every reference-produced signature.  GPU tests: fabgpu_csp_block_preverify2 on every block — flags, statuses, digests, the
verdict memo — through the C ABI."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import bccsp_sw_oracle as po
import fabgpu
import fabric_decode as fd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "ledger_blocks.json")))["blocks"]
BLOCKS = [(b, base64.b64decode(b["block_b64"])) for b in FIX]
V20 = [(b, raw) for b, raw in BLOCKS if b["source"] == "v20"]


def expected_tuples(blk):
    """The SignedData the reference's validators build, in the walker's order: per tx creator then endorsements, block sigs last."""
    out = []
    for t, tx in enumerate(blk["txs"]):
        if not tx["creator"][0]:
            # no SignatureHeader.creator (the v11 / v13 harness genesis blocks): checkSignatureFromCreator fails on its nil-argument
            # check (core/common/validation/msgvalidation.go:28-31) before any signature is looked at - no tuple, transaction left to Go
            continue
        out.append((t, fabgpu.TUPLE_CREATOR) + tx["creator"])
        for a in tx["actions"]:
            for e in a["endorsements"]:
                out.append((t, fabgpu.TUPLE_ENDORSEMENT) + e)
    for s in blk["block_sigs"]:
        out.append((fabgpu.BLOCK_LEVEL_TX, fabgpu.TUPLE_BLOCK_SIG) + s)
    return out


def cut(arena, span):
    return arena[span[0]:span[0] + span[1]]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_fixture_file_is_what_the_generator_extracts_from_the_reference(tmp_path):
    import subprocess
    import sys
    gen = os.path.join(ROOT, "tests", "golden", "gen_ledger_block_fixtures.py")
    src = open(gen).read().replace('os.path.join(os.path.dirname(os.path.abspath(__file__)), "ledger_blocks.json")', repr(str(tmp_path / "out.json")))
    p = tmp_path / "gen.py"
    p.write_text(src)
    subprocess.run([sys.executable, str(p), REF], check=True, capture_output=True)
    assert json.load(open(tmp_path / "out.json")) == json.load(open(os.path.join(ROOT, "tests", "golden", "ledger_blocks.json")))


def test_fixture_inventory():
    assert len(BLOCKS) == 74 and len(V20) == 20
    n_env = sum(b["n_tx"] for b, _ in BLOCKS)
    assert n_env == 110
    for b, raw in BLOCKS:                                # the generator's own bookkeeping against the independent decoder
        blk = fd.decode_block(raw)
        assert blk["number"] == b["number"] and blk["data_hash"].hex() == b["data_hash"] and len(blk["envelopes"]) == b["n_tx"]
        assert blk["data_hash_ok"]                       # protoutil.BlockDataHash over the re-marshalled data == header.data_hash
        if b["tx_filter"] is not None:
            assert len(bytes.fromhex(b["tx_filter"])) == b["n_tx"]


def test_walker_tuples_equal_the_independent_decoder_on_every_ledger_block():
    n_tuples = 0
    for b, raw in BLOCKS:
        blk = fd.decode_block(raw)
        want = expected_tuples(blk)
        got, arena = fabgpu.block_tuples(raw)
        assert len(got) == len(want), (b["source"], b["chain"], b["number"])
        for g, (tx, kind, ident, msg, sig) in zip(got, want):
            assert (g["tx"], g["kind"]) == (tx, kind)
            assert cut(arena, g["identity"]) == ident
            assert cut(arena, g["prefix"]) + cut(arena, g["suffix"]) == msg
            assert cut(arena, g["sig"]) == sig
            if kind == fabgpu.TUPLE_ENDORSEMENT:         # the shared prefix is the proposal response payload itself
                assert g["prefix"][1] > 0 and cut(arena, g["suffix"]) == ident
        n_tuples += len(got)
        p = fabgpu.block_parse(raw)
        assert p["n_tx"] == b["n_tx"] and list(p["tx_type"]) == [tx["type"] for tx in blk["txs"]]
        assert p["channel_id"] == blk["txs"][0]["channel"]
    assert n_tuples == (110 - 6) + 105 + 19              # creators (6 genesis envelopes name none) + endorsements + orderer block signatures


def test_walker_hash_checks_equal_the_independent_decoder_on_every_ledger_block():
    n_txid = n_ph = n_txid_ok = 0
    for b, raw in BLOCKS:
        blk = fd.decode_block(raw)
        checks = fabgpu.block_hash_checks(raw)
        want = []
        for t, tx in enumerate(blk["txs"]):
            if tx["type"] != 3:
                continue
            want.append((t, 0, tx["txid_computed"], tx["tx_id"].encode("latin1")))
            for a in tx["actions"]:
                want.append((t, 1, a["proposal_hash_computed"], a["proposal_hash_expect"]))
        assert len(checks) == len(want)
        for (tx, kind, pieces, expect), (wt, wk, computed, claimed) in zip(checks, want):
            assert (tx, kind) == (wt, wk)
            digest = hashlib.sha256(b"".join(raw[s:e] for s, e in pieces)).digest()
            assert (digest.hex() if kind == 0 else digest) == computed
            assert raw[expect[0]:expect[1]] == claimed
            if kind == 0:
                n_txid += 1
                n_txid_ok += digest.hex().encode() == claimed
            else:
                n_ph += 1
                assert digest == claimed                 # all 99 proposal hashes of the reference's ledgers match
    assert (n_txid, n_ph, n_txid_ok) == (99, 99, 15)     # only the v20 network computed real TxIDs (v11 / v13 harnesses used UUIDs)


def test_oracle_accepts_every_reference_produced_signature():
    """20 creator + 21 endorsement + 19 orderer signatures of the v20 ledger: made by the reference, accepted by the restatement."""
    n = {0: 0, 1: 0, 2: 0}
    for b, raw in V20:
        blk = fd.decode_block(raw)
        for tx, kind, ident, msg, sig in expected_tuples(blk):
            mspid, pub = fd.identity_pubkey(ident)
            assert pub is not None and mspid.endswith("MSP"), (b["number"], kind, mspid)
            assert po.identity_verify(pub, msg, sig) is None
            assert po.identity_verify(pub, msg + b"\x00", sig) is not None
            n[kind] += 1
    assert (n[0], n[1], n[2]) == (20, 21, 19)
    assert all(bytes.fromhex(b["tx_filter"]) == b"\x00" * b["n_tx"] for b, _ in V20)     # the committing peer flagged every tx VALID


def test_walker_refuses_repeated_singular_fields():
    """golang/protobuf takes the LAST occurrence of a singular field; a walker that takes the first would verify other
    bytes than the Go validators see.  This one declares the envelope not understood (tx left to Go) and emits no tuple for it."""
    b, raw = next((b, raw) for b, raw in V20 if b["number"] == 6)
    blk = fd.decode_block(raw)
    env = blk["envelopes"][0]
    base_n = len(fabgpu.block_tuples(raw)[0])
    n_tx_tuples = 1 + sum(len(a["endorsements"]) for a in blk["txs"][0]["actions"])

    def reblock(new_env):
        import blockbuilder as bb
        header = fd.last(raw, 1)
        meta = fd.last(raw, 3)
        return bb.fbytes(1, header) + bb.fbytes(2, bb.fbytes(1, new_env)) + bb.fbytes(3, meta)

    import blockbuilder as bb
    payload, sig = fd.last(env, 1), fd.last(env, 2)
    assert len(fabgpu.block_tuples(reblock(env))[0]) == base_n               # the rebuild itself changes nothing
    cases = {
        "payload twice": bb.fbytes(1, b"\x0a\x00") + bb.fbytes(1, payload) + bb.fbytes(2, sig),
        "signature twice": bb.fbytes(1, payload) + bb.fbytes(2, b"junk") + bb.fbytes(2, sig),
        "payload as varint": bb.fvarint(1, 5) + bb.fbytes(1, payload) + bb.fbytes(2, sig),
    }
    hdr, data = fd.last(payload, 1), fd.last(payload, 2)
    cases["header twice"] = bb.fbytes(1, bb.fbytes(1, hdr) + bb.fbytes(1, hdr) + bb.fbytes(2, data)) + bb.fbytes(2, sig)
    chdr, shdr = fd.last(hdr, 1), fd.last(hdr, 2)
    cases["signature header twice"] = bb.fbytes(1, bb.fbytes(1, bb.fbytes(1, chdr) + bb.fbytes(2, shdr) + bb.fbytes(2, shdr)) + bb.fbytes(2, data)) + bb.fbytes(2, sig)
    cases["creator twice"] = bb.fbytes(1, bb.fbytes(1, bb.fbytes(1, chdr) + bb.fbytes(2, bb.fbytes(1, b"evil") + shdr)) + bb.fbytes(2, data)) + bb.fbytes(2, sig)
    cases["type twice"] = bb.fbytes(1, bb.fbytes(1, bb.fbytes(1, bb.fvarint(1, 1) + chdr) + bb.fbytes(2, shdr)) + bb.fbytes(2, data)) + bb.fbytes(2, sig)
    for name, e in cases.items():
        blk2 = reblock(e)
        got, _ = fabgpu.block_tuples(blk2)
        p = fabgpu.block_parse(blk2)
        assert len(got) == base_n - n_tx_tuples, name      # only the orderer's block signature is left
        assert all(g["kind"] == fabgpu.TUPLE_BLOCK_SIG for g in got), name
        assert fabgpu.block_hash_checks(blk2) == [], name
    # a repeated top-level field would MERGE in Go (two BlockData = concatenated transactions): the block is refused as a whole
    with pytest.raises(fabgpu.FabgpuError):
        fabgpu.block_tuples(raw + bb.fbytes(2, bb.fbytes(1, env)))


def test_block_header_bytes_against_the_orderer_signatures():
    """protoutil.BlockHeaderBytes is rebuilt by the walker (ASN.1, not in the block): the 19 orderer signatures only verify if number,
    previous_hash and data_hash are encoded exactly as Go's asn1.Marshal does — including number = 0..19 minimal INTEGERs."""
    for b, raw in V20:
        blk = fd.decode_block(raw)
        got, arena = fabgpu.block_tuples(raw)
        tail = [g for g in got if g["kind"] == fabgpu.TUPLE_BLOCK_SIG]
        assert len(tail) == len(blk["block_sigs"])
        hb = fd.block_header_bytes(blk["number"], blk["previous_hash"], blk["data_hash"])
        for g in tail:
            assert cut(arena, g["suffix"]).endswith(hb) and g["suffix"][0] >= (len(raw) + 63) // 64 * 64
    # large numbers: the INTEGER grows a leading zero when the top bit is set
    for number, enc in ((127, "02017f"), (128, "02020080"), (255, "020200ff"), (65535, "020300ffff"), (1 << 63, "0209008000000000000000")):
        assert fd.block_header_bytes(number, b"", b"").hex() == "30%02x" % (len(enc) // 2 + 4) + enc + "04000400"


# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def csp():
    c = fabgpu.GPUCSP()
    yield c
    c.close()


@pytest.mark.gpu
def test_preverify_pass_on_every_block_of_the_reference_ledgers(csp):
    """The whole pass, through the C ABI, on reference-produced bytes: flags == 0 exactly where the committing peer's
    TRANSACTIONS_FILTER says VALID (v20), every reference signature status 0, UUID-txid ledgers flagged BAD_TXID or left to Go."""
    seen = {0: 0, 1: 0, 2: 0}
    for b, raw in BLOCKS:
        blk = fd.decode_block(raw)
        want = expected_tuples(blk)
        r = fabgpu.preverify_block2(csp, raw, block_seq=b["number"])
        assert len(r["tuple_status"]) == len(want) and r["block_sigs_understood"]
        assert r["n_block_sigs"] == len(blk["block_sigs"])
        for i, (tx, kind, ident, msg, sig) in enumerate(want):
            assert (int(r["tuple_tx"][i]), int(r["tuple_kind"][i])) == (tx, kind)
            sp = [int(x) for x in r["tuple_spans"][i]]
            assert r["arena"][sp[0]:sp[0] + sp[1]] == ident and r["arena"][sp[6]:sp[6] + sp[7]] == sig
            assert r["arena"][sp[2]:sp[2] + sp[3]] + r["arena"][sp[4]:sp[4] + sp[5]] == msg
            mspid, pub = fd.identity_pubkey(ident)
            if pub is None:                                # ledger-harness identities: not certificates -> bccsp/sw decides
                assert r["tuple_status"][i] == fabgpu.TUPLE_ST_NEEDS_SW and not r["tuple_hashed"][i]
                continue
            assert bytes(r["tuple_qxy"][i]) == pub[0].to_bytes(32, "big") + pub[1].to_bytes(32, "big")
            assert r["tuple_hashed"][i] and bytes(r["tuple_digest"][i]) == hashlib.sha256(msg).digest()
            assert r["tuple_status"][i] == 0, (b["source"], b["number"], i, kind)
            assert po.identity_verify(pub, msg, sig) is None
            seen[kind] += 1
        flags = list(r["tx_flags"])
        if b["source"] == "v20":
            assert flags == list(bytes.fromhex(b["tx_filter"]))          # all zero: VALID where the reference said VALID
        else:
            for t, tx in enumerate(blk["txs"]):
                # creator identity is not a certificate -> bccsp/sw decides (outranks the UUID TxID); the genesis envelopes name no
                # creator at all -> left to the Go validators
                assert flags[t] == (fabgpu.TX_NEEDS_SW if tx["creator"][0] else fabgpu.TX_NOT_UNDERSTOOD)
    assert (seen[0], seen[1], seen[2]) == (20, 21, 19)


@pytest.mark.gpu
def test_tampering_with_reference_blocks_flips_exactly_the_right_flag(csp):
    b, raw = next((b, raw) for b, raw in V20 if b["number"] == 6)        # one ENDORSER_TRANSACTION with endorsements
    blk = fd.decode_block(raw)
    base = fabgpu.preverify_block2(csp, raw)
    assert list(base["tx_flags"]) == [0] and (base["tuple_status"] == 0).all()
    sp = base["tuple_spans"]
    kinds = list(base["tuple_kind"])

    def flip(off):
        m = bytearray(raw)
        m[off] ^= 1
        return bytes(m)

    i_end = kinds.index(fabgpu.TUPLE_ENDORSEMENT)
    i_bs = kinds.index(fabgpu.TUPLE_BLOCK_SIG)
    # a bit of the proposal response payload's extension (not its proposal_hash field): endorsements and the creator signature break
    prp = blk["txs"][0]["actions"][0]["prp"]
    ext_off = raw.index(prp) + len(prp) - 1
    r = fabgpu.preverify_block2(csp, flip(ext_off))
    assert list(r["tx_flags"]) == [fabgpu.TX_BAD_CREATOR_SIGNATURE]
    assert r["tuple_status"][0] == 1 and r["tuple_status"][i_end] == 1 and r["tuple_status"][i_bs] == 0
    # a bit inside an endorsement's signature value (last byte of s): that endorsement fails, and the creator (who signed the payload)
    s_off = int(sp[i_end][6]) + int(sp[i_end][7]) - 1
    r = fabgpu.preverify_block2(csp, flip(s_off))
    assert r["tuple_status"][i_end] in (1, 2) and list(r["tx_flags"]) == [fabgpu.TX_BAD_CREATOR_SIGNATURE]
    # a bit of the creator's signature only: creator fails, endorsements hold
    c_off = int(sp[0][6]) + int(sp[0][7]) - 1
    r = fabgpu.preverify_block2(csp, flip(c_off))
    assert r["tuple_status"][0] in (1, 2) and r["tuple_status"][i_end] == 0 and list(r["tx_flags"]) == [fabgpu.TX_BAD_CREATOR_SIGNATURE]
    # the data hash in the block header: only the orderer's signature over the header notices
    dh_off = raw.index(blk["data_hash"])
    r = fabgpu.preverify_block2(csp, flip(dh_off + 5))
    assert r["tuple_status"][i_bs] == 1 and list(r["tx_flags"]) == [0] and (r["tuple_status"][:i_bs] == 0).all()
    # block signatures can be left out of the pass
    r = fabgpu.preverify_block2(csp, raw, block_sigs=False)
    assert r["tuple_status"][i_bs] == fabgpu.TUPLE_ST_SKIPPED and (r["tuple_status"][:i_bs] == 0).all()


@pytest.mark.gpu
def test_verdict_memo_replays_bccsp_verify_lookups(csp):
    """SURVEY 8(f) rank 1, second half: after a pass with FABGPU_PASS_SEED_MEMO, the bccsp.Verify(k, sig, digest) calls the unchanged
    validators make (msp/identities.go:188) find their verdict; anything else misses (-> bccsp/sw); eviction is per block."""
    fabgpu.memo_evict_block(csp, 0)
    before = fabgpu.memo_stats(csp)
    total = 0
    for b, raw in V20:
        blk = fd.decode_block(raw)
        r = fabgpu.preverify_block2(csp, raw, block_seq=1000 + b["number"], seed_memo=True)
        want = expected_tuples(blk)
        assert r["memo_seeded"] == len(want)
        total += len(want)
        for tx, kind, ident, msg, sig in want:
            _, pub = fd.identity_pubkey(ident)
            qx, qy = pub[0].to_bytes(32, "big"), pub[1].to_bytes(32, "big")
            digest = hashlib.sha256(msg).digest()                         # what identity.Verify computes with bccsp.Hash
            assert fabgpu.memo_lookup(csp, qx, qy, sig, digest) == 0      # hit: the GPU's verdict
            assert fabgpu.memo_lookup(csp, qx, qy, sig, hashlib.sha256(msg + b"x").digest()) is None      # other digest: miss
            assert fabgpu.memo_lookup(csp, qy, qx, sig, digest) is None                                    # other key: miss
            assert fabgpu.memo_lookup(csp, qx, qy, sig + b"\x00", digest) is None                          # other signature bytes: miss
            assert fabgpu.memo_lookup(csp, qx, qy, sig + digest[:1], digest[1:]) is None                   # framing: no (sig||d[:k], d[k:]) collision
    st = fabgpu.memo_stats(csp)
    assert st["entries"] - before["entries"] == total == 60
    assert fabgpu.memo_evict_block(csp, 1006) == len(expected_tuples(fd.decode_block(V20[6][1])))
    assert fabgpu.memo_evict_block(csp, 1006) == 0
    for b, raw in V20:
        fabgpu.memo_evict_block(csp, 1000 + b["number"])
    assert fabgpu.memo_stats(csp)["entries"] == before["entries"]
    # invalid signatures are remembered as invalid (status 1), never as valid
    b, raw = V20[6]
    m = bytearray(raw)
    blk = fd.decode_block(raw)
    prp = blk["txs"][0]["actions"][0]["prp"]
    m[raw.index(prp) + len(prp) - 1] ^= 1
    r = fabgpu.preverify_block2(csp, bytes(m), block_seq=77, seed_memo=True)
    blk2 = fd.decode_block(bytes(m))
    ident, msg, sig = blk2["txs"][0]["actions"][0]["endorsements"][0]
    _, pub = fd.identity_pubkey(ident)
    assert fabgpu.memo_lookup(csp, pub[0].to_bytes(32, "big"), pub[1].to_bytes(32, "big"), sig, hashlib.sha256(msg).digest()) == 1
    fabgpu.memo_evict_block(csp, 77)


@pytest.mark.gpu
def test_pass_at_block_arrival_then_mcs_then_validators_then_evict(csp):
    """The sequence the Go side runs since round 3 (fabric-mod_amd/go/extensions/gossip/state/preverify_on_arrival.go + extensions/validation/
    preverify.go), replayed through the C ABI on the reference's own Fabric 2.0 blocks: several blocks ARRIVE (AddPayload: the marshalled
    bytes go through the pass, memo seeded under a per-block name) before the committer reaches the first; then, block by block, the
    validator wrapper finds the block's verdicts waiting (no second pass), the BlockValidation policy's lookups for the orderers'
    signatures hit, every creator / endorsement lookup of the validators hits, and the block's entries are evicted.  Hit counts are the
    reference's signature counts: 20 creator + 21 endorsement + 19 orderer."""
    seqs = {}
    before = fabgpu.memo_stats(csp)
    routes0 = fabgpu.pass_routes(csp)
    for b, raw in V20:                                                   # arrival: gossip hands over proto.Marshal(block) (gossip/state/state.go:592)
        seq = 0xA11CE000 + b["number"]
        assert fabgpu.memo_has_block(csp, seq) == 0
        r = fabgpu.preverify_block2(csp, raw, block_seq=seq, seed_memo=True, lean=True)
        seqs[b["number"]] = (seq, r["memo_seeded"])
        assert fabgpu.memo_has_block(csp, seq) == r["memo_seeded"] > 0
    passes = fabgpu.pass_routes(csp)
    assert passes["device_walks"] + passes["host_walks"] - routes0["device_walks"] - routes0["host_walks"] == len(V20)
    hits = {0: 0, 1: 0, 2: 0}
    for b, raw in V20:                                                   # the committer reaches the blocks, in order
        seq, seeded = seqs[b["number"]]
        assert fabgpu.memo_has_block(csp, seq) == seeded                 # Validate: HasBlock(seq) -> no marshal, no second pass
        blk = fd.decode_block(raw)
        want = expected_tuples(blk)
        assert len(want) == seeded
        h0 = fabgpu.memo_stats(csp)["hits"]
        for kind_wanted in (2, 0, 1):                                    # orderer signatures (block validation policy) first, then the validators
            for tx, kind, ident, msg, sig in want:
                if kind != kind_wanted:
                    continue
                _, pub = fd.identity_pubkey(ident)
                assert fabgpu.memo_lookup(csp, pub[0].to_bytes(32, "big"), pub[1].to_bytes(32, "big"), sig, hashlib.sha256(msg).digest()) == 0
                hits[kind] += 1
        assert fabgpu.memo_stats(csp)["hits"] - h0 == len(want)
        assert fabgpu.memo_evict_block(csp, seq) == seeded               # Validate returned
        assert fabgpu.memo_has_block(csp, seq) == 0
    assert (hits[0], hits[1], hits[2]) == (20, 21, 19)
    assert fabgpu.memo_stats(csp)["entries"] == before["entries"]
    assert fabgpu.pass_routes(csp)["device_walks"] + fabgpu.pass_routes(csp)["host_walks"] == passes["device_walks"] + passes["host_walks"]   # no pass since arrival


@pytest.mark.gpu
def test_memo_and_identity_cache_are_bounded(csp):
    L = fabgpu.load()
    L.fabgpu_csp_memo_set_capacity(csp._h, 8)
    try:
        for k, (b, raw) in enumerate(V20[5:12]):
            fabgpu.preverify_block2(csp, raw, block_seq=500 + k, seed_memo=True)
            assert fabgpu.memo_stats(csp)["entries"] <= 8 + 5              # at most the cap plus the newest block
    finally:
        L.fabgpu_csp_memo_set_capacity(csp._h, 1 << 18)
        for k in range(7):
            fabgpu.memo_evict_block(csp, 500 + k)
    import ctypes
    n = ctypes.c_uint64(0)
    L.fabgpu_csp_identity_cache_limits(csp._h, 2, 256, 64)
    try:
        for b, raw in V20[:8]:
            fabgpu.preverify_block2(csp, raw)
            L.fabgpu_csp_identity_cache_size(csp._h, ctypes.byref(n))
            assert n.value <= 2
            r = fabgpu.preverify_block2(csp, raw)
            assert (r["tuple_status"] == 0).all()                          # eviction never changes a verdict
    finally:
        L.fabgpu_csp_identity_cache_limits(csp._h, 4096, 256, 64)
