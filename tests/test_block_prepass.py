"""Block-level pre-verify pass: the C++ block walker + x509 key extraction on the CPU, the whole pass
on the GPU.  Synthetic blocks come from tests/blockbuilder.py (an independent encoder), identities from
tests/golden/block_identities.json (openssl-generated, test-only keys), signatures from the oracle."""
import glob
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import bccsp_sw_oracle as po
import blockbuilder as bb
import fabgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IDS = json.load(open(os.path.join(ROOT, "tests", "golden", "block_identities.json")))["identities"]
REF = "/root/reference"


def test_x509_key_extraction_on_generated_identities():
    for i in IDS:
        got = fabgpu.x509_p256_pubkey(i["pem"].encode())
        if i["curve"] == "prime256v1":
            assert got == (bytes.fromhex(i["qx"].rjust(64, "0")), bytes.fromhex(i["qy"].rjust(64, "0")))
            assert po.pt_mul(int(i["d"], 16), (po.GX, po.GY)) == (int(i["qx"], 16), int(i["qy"], 16))   # the fixture's key pair is consistent
        else:
            assert got is None                               # P-384: not ours
    assert fabgpu.x509_p256_pubkey(b"-----BEGIN CERTIFICATE-----\nAAAA\n-----END CERTIFICATE-----\n") is None
    assert fabgpu.x509_p256_pubkey(b"not a pem") is None


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_x509_key_extraction_equals_openssl_on_the_reference_certificates():
    """Every certificate fixture of the reference: the C++ SPKI walker and `openssl x509 -pubkey` agree on (curve, Qx, Qy)."""
    pems = sorted(glob.glob(REF + "/**/*.pem", recursive=True))
    checked = 0
    for path in pems:
        txt = open(path, "rb").read()
        if b"BEGIN CERTIFICATE" not in txt:
            continue
        r = subprocess.run(["openssl", "x509", "-in", path, "-noout", "-text"], capture_output=True, text=True)
        if r.returncode != 0:
            continue
        is_p256 = "ASN1 OID: prime256v1" in r.stdout
        got = fabgpu.x509_p256_pubkey(txt)
        assert (got is not None) == is_p256, path
        if is_p256:
            import re
            pub = re.sub(r"[^0-9a-f]", "", re.search(r"pub:\s*((?:[0-9a-f:]+\s*)+)ASN1 OID", r.stdout).group(1))
            assert got[0].hex() == pub[2:66] and got[1].hex() == pub[66:130], path
            checked += 1
    assert checked >= 100


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_block_walker_on_the_reference_block_fixtures():
    """The reference's own marshalled blocks pin the outer layers of the walker (Block, BlockData, Envelope, Payload, Header,
    ChannelHeader, SignatureHeader): one CONFIG envelope on the expected channel."""
    seen = 0
    for path, channel in ((REF + "/orderer/common/cluster/testdata/mychannel.block", "mychannel"),
                          (REF + "/orderer/consensus/etcdraft/testdata/mychannel.block", "mychannel")):
        if not os.path.exists(path):
            continue
        p = fabgpu.block_parse(open(path, "rb").read())
        assert p["n_tx"] == 1 and p["channel_id"] == channel and list(p["tx_type"]) == [1]      # HeaderType CONFIG
        assert p["n_prefixes"] == 0 and p["n_tuples"] <= 1
        seen += 1
    assert seen >= 1


def _sign(ident, msg, k):
    r, s = po.sign_raw(int(ident["d"], 16), hashlib.sha256(msg).digest(), k)
    return r, s


def build_block(n_tx, rng, corrupt=True):
    """n_tx endorser transactions x 3 endorsements by 4 endorsers, 2 creators, TxIDs and proposal hashes as the reference's
    validators recompute them; returns (block bytes, expected tx flags)."""
    p256 = [i for i in IDS if i["curve"] == "prime256v1"]
    p384 = [i for i in IDS if i["curve"] != "prime256v1"][0]
    sid = {i["cn"]: bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS}
    endorsers, creators = p256[:4], p256[4:6]
    envs, want = [], []
    for t in range(n_tx):
        creator = creators[t % 2]
        cbytes = sid[creator["cn"]]
        ext = bytes(rng.integers(0, 256, size=int(rng.integers(100, 1200)), dtype=np.uint8))
        ccpp = bytes(rng.integers(0, 256, size=200, dtype=np.uint8))
        nonce = bytes(rng.integers(0, 256, size=24, dtype=np.uint8))
        mode = (t % 13) if corrupt else 0
        picks = list(rng.choice(4, size=3, replace=False))
        flag = [fabgpu.TX_ALL_SIGNATURES_VALID]

        def sign_ends(prp, t=t, mode=mode, picks=picks, flag=flag):
            ends = []
            for j in picks:
                e = endorsers[j]
                ebytes = sid[e["cn"]]
                r, s = _sign(e, prp + ebytes, int(rng.integers(1, 1 << 62)))
                ends.append([ebytes, po.marshal_ecdsa_signature(r, s), r, s])
            if mode == 3:                               # a tampered endorsement (r + 1)
                ends[1][1] = po.marshal_ecdsa_signature(ends[1][2] + 1, ends[1][3]); flag[0] = fabgpu.TX_BAD_ENDORSEMENT
            if mode == 4:                               # a high-S endorsement: bccsp/sw rejects it with an error
                ends[0][1] = po.marshal_ecdsa_signature(ends[0][2], po.N - ends[0][3]); flag[0] = fabgpu.TX_BAD_ENDORSEMENT
            if mode == 5:                               # garbage DER
                ends[2][1] = b"\x30\x03\x02\x01"; flag[0] = fabgpu.TX_BAD_ENDORSEMENT
            if mode == 6:                               # an endorser with a P-384 identity: bccsp/sw must decide
                ends[2][0] = sid[p384["cn"]]; flag[0] = fabgpu.TX_NEEDS_SW
            if mode == 10:                              # bad creator AND bad endorsement: the creator outranks
                ends[0][1] = po.marshal_ecdsa_signature(ends[0][2] + 1, ends[0][3])
            return [(e[0], e[1]) for e in ends]
        payload, prp = bb.consistent_endorser_tx("mychannel", cbytes, nonce, ccpp, ext, sign_ends, bad_txid=(mode == 11), bad_phash=(mode == 12))
        if mode == 11:
            flag[0] = fabgpu.TX_BAD_TXID
        if mode == 12:
            flag[0] = fabgpu.TX_BAD_PROPOSAL_HASH
        if mode == 7:                                   # a CONFIG envelope: only the creator signature is checked (no TxID check either)
            payload = bb.endorser_tx_payload(1, "mychannel", "not-a-hash", cbytes, nonce, [(ccpp, prp, [])])
        if mode == 8:                                   # transaction bytes that are not a peer.Transaction
            hdr = bb.fbytes(1, bb.channel_header(3, "mychannel", bb.compute_txid(b"n", cbytes))) + bb.fbytes(2, bb.signature_header(cbytes, b"n"))
            payload = bb.fbytes(1, hdr) + bb.fbytes(2, b"\x0a\xff\xff\xff\xff\x0f" + b"junk")
            flag[0] = fabgpu.TX_NOT_UNDERSTOOD
        r, s = _sign(creator, payload, int(rng.integers(1, 1 << 62)))
        if mode == 9 or mode == 10:                     # creator signed something else
            r, s = _sign(creator, payload + b"!", 99); flag[0] = fabgpu.TX_BAD_CREATOR_SIGNATURE
        envs.append(bb.envelope(payload, po.marshal_ecdsa_signature(r, s)))
        want.append(flag[0])
    return bb.block(7, envs), np.array(want, dtype=np.uint8)


def test_block_walker_on_synthetic_blocks():
    rng = np.random.default_rng(5)
    blk, want = build_block(52, rng)
    p = fabgpu.block_parse(blk)
    assert p["n_tx"] == 52 and p["channel_id"] == "mychannel"
    n_cfg = sum(1 for t in range(52) if t % 13 == 7)
    n_bad = sum(1 for t in range(52) if t % 13 == 8)
    assert list(p["tx_type"]).count(1) == n_cfg and list(p["tx_type"]).count(3) == 52 - n_cfg
    assert p["n_tuples"] == (52 - n_bad) + 3 * (52 - n_cfg - n_bad) and p["n_prefixes"] == 52 - n_cfg - n_bad
    # the TxID / proposal-hash checks: recomputed here with hashlib from the spans the walker reports
    checks = fabgpu.block_hash_checks(blk)
    assert len(checks) == 2 * (52 - n_cfg - n_bad)
    bad = {}
    for tx, kind, pieces, (e0, e1) in checks:
        digest = hashlib.sha256(b"".join(blk[a:b] for a, b in pieces)).digest()
        ok = (blk[e0:e1] == digest.hex().encode()) if kind == 0 else (blk[e0:e1] == digest)
        if not ok:
            bad[tx] = kind
    assert bad == {t: (0 if t % 13 == 11 else 1) for t in range(52) if t % 13 in (11, 12)}
    with pytest.raises(fabgpu.FabgpuError):
        fabgpu.block_parse(b"\x12\xff\xff\xff\xff\x0f")              # BlockData length runs past the buffer


@pytest.mark.gpu
def test_preverify_pass_end_to_end(monkeypatch):
    csp = fabgpu.GPUCSP(device=0)
    csp.set_option("pass_stage_min_bytes", 1 << 40)         # the pass with the walk on the HOST (the device route: test_device_walk.py)
    rng = np.random.default_rng(6)
    blk, want = build_block(220, rng)
    before = csp.key_count()
    out = fabgpu.preverify_block(csp, blk)
    assert (out["tx_flags"] == want).all()
    assert csp.key_count() == before + 6                                  # 4 endorsers + 2 creators registered once, the P-384 one never
    out2 = fabgpu.preverify_block(csp, blk)                               # identities now come from the cache
    assert (out2["tx_flags"] == want).all() and (out2["tuple_status"] == out["tuple_status"]).all() and csp.key_count() == before + 6
    # per-tuple detail: exactly the corrupted tuples are non-zero
    st = out["tuple_status"]
    assert set(np.unique(st)) <= {0, 1, 2, 5, 6}
    assert (st == fabgpu.TUPLE_ST_NEEDS_SW).sum() == sum(1 for t in range(220) if t % 13 == 6)
    assert (st == 2).sum() == sum(1 for t in range(220) if t % 13 == 4)
    assert (st == fabgpu.TUPLE_ST_BAD_DER).sum() == sum(1 for t in range(220) if t % 13 == 5)
    assert (out["tx_flags"] == fabgpu.TX_BAD_TXID).sum() == sum(1 for t in range(220) if t % 13 == 11)
    assert (out["tx_flags"] == fabgpu.TX_BAD_PROPOSAL_HASH).sum() == sum(1 for t in range(220) if t % 13 == 12)
    # a clean block: every transaction valid
    blk3, want3 = build_block(64, rng, corrupt=False)
    assert (fabgpu.preverify_block(csp, blk3)["tx_flags"] == 0).all() and (want3 == 0).all()
    csp.close()


@pytest.mark.gpu
def test_preverify_pass_with_known_and_new_identities(monkeypatch):
    """A block signed by identities of which only some have a device table (newcomers among known ones): the answer does not depend
    on which path - per-key tables or keys carried along - the tuples took.  (Host walk: one launch for the whole block, so "some" means
    the fresh-key kernel for everybody; the device route decides per launch class, test_device_walk.py.)"""
    csp = fabgpu.GPUCSP(device=0)
    csp.set_option("pass_stage_min_bytes", 1 << 40)
    L = csp._L
    before = csp.key_count()
    rng = np.random.default_rng(61)
    blk, want = build_block(300, rng)
    L.fabgpu_csp_identity_cache_limits(csp._h, 4096, 3, 1)               # three tables at most, earned at the first sight
    first = fabgpu.preverify_block2(csp, blk, block_seq=1)                # nobody has a table yet
    assert (first["tx_flags"] == want).all() and first["n_keyed"] == 0
    assert csp.key_count() == before + 3
    second = fabgpu.preverify_block2(csp, blk, block_seq=2)               # three of the six signers have one: still the fresh-key path
    n_sub = int(second["tuple_hashed"].sum())
    assert second["n_keyed"] == 0 and (second["tx_flags"] == want).all()
    L.fabgpu_csp_identity_cache_limits(csp._h, 4096, 256, 1)              # room for everybody: this pass registers the rest ...
    fabgpu.preverify_block2(csp, blk, block_seq=3)
    fourth = fabgpu.preverify_block2(csp, blk, block_seq=4)               # ... and the next is all tables
    assert fourth["n_keyed"] == n_sub and (fourth["tx_flags"] == want).all()
    for k in ("tuple_status", "tuple_hashed", "tuple_digest", "tuple_tx", "tuple_kind"):
        assert (second[k] == first[k]).all() and (fourth[k] == first[k]).all(), k
    csp.close()


# ---- blocks whose creators are idemix identities (BASELINE config 5 at the block level) -----------------------------------
def build_mixed_block(n_tx, rng, idemix_every=5):
    """As build_block(corrupt=False), but every idemix_every-th transaction is created by an idemix identity of IdemixMSP1 whose
    envelope signature is a NymSignature; some of those are tampered.  Returns (block, expected tx flags, #idemix creators)."""
    import random as pyrandom
    import idemix_oracle as io
    from idemix_common import be32, fixtures
    fx = fixtures()
    ipk, sk = fx["MSP1OU1"]["ipk"], fx["MSP1OU1"]["signer"].sk
    prng = pyrandom.Random(int(rng.integers(1, 1 << 30)))
    p256 = [i for i in IDS if i["curve"] == "prime256v1"]
    sid = {i["cn"]: bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS}
    endorsers, creators = p256[:4], p256[4:6]
    envs, want, n_idemix = [], [], 0
    for t in range(n_tx):
        ext = bytes(rng.integers(0, 256, size=int(rng.integers(100, 1200)), dtype=np.uint8))
        ccpp = bytes(rng.integers(0, 256, size=200, dtype=np.uint8))
        picks = list(rng.choice(4, size=3, replace=False))

        def sign_ends(prp, picks=picks):
            ends = []
            for j in picks:
                e = endorsers[j]
                r, s = _sign(e, prp + sid[e["cn"]], int(rng.integers(1, 1 << 62)))
                ends.append((sid[e["cn"]], po.marshal_ecdsa_signature(r, s)))
            return ends
        flag = fabgpu.TX_ALL_SIGNATURES_VALID
        if t % idemix_every == 0:
            n_idemix += 1
            nym, r_nym = io.make_nym(sk, ipk, prng)
            which = (t // idemix_every) % 6
            mspid = "IdemixMSP1" if which != 4 else "UnknownIdemixMSP"
            cbytes = bb.serialized_idemix_identity(mspid, be32(nym[0]), be32(nym[1]))
            payload, _ = bb.consistent_endorser_tx("mychannel", cbytes, b"nonce%d" % t, ccpp, ext, sign_ends)
            sig = io.nym_sign(sk, nym, r_nym, ipk, payload, prng)
            if which == 2:                                # signed another payload
                sig = io.nym_sign(sk, nym, r_nym, ipk, payload + b"!", prng); flag = fabgpu.TX_BAD_CREATOR_SIGNATURE
            if which == 3:                                # a 31-byte field: amcl-internal, left to bccsp/idemix
                sig = dict(sig, nonce=sig["nonce"][:31]); flag = fabgpu.TX_NEEDS_SW
            if which == 4:                                # an idemix MSP nobody registered
                flag = fabgpu.TX_NEEDS_SW
            if which == 5:                                # s-value >= r: outside the device's domain
                sig = dict(sig, proof_s_sk=be32(io.R + 5)); flag = fabgpu.TX_NEEDS_SW
            envs.append(bb.envelope(payload, io.nym_signature_marshal(sig)))
        else:
            creator = creators[t % 2]
            payload, _ = bb.consistent_endorser_tx("mychannel", sid[creator["cn"]], b"nonce%d" % t, ccpp, ext, sign_ends)
            r, s = _sign(creator, payload, int(rng.integers(1, 1 << 62)))
            envs.append(bb.envelope(payload, po.marshal_ecdsa_signature(r, s)))
        want.append(flag)
    return bb.block(9, envs), np.array(want, dtype=np.uint8), n_idemix


def test_walker_sees_idemix_creators_as_ordinary_tuples():
    rng = np.random.default_rng(8)
    blk, want, n_idemix = build_mixed_block(30, rng)
    p = fabgpu.block_parse(blk)
    assert p["n_tx"] == 30 and p["n_tuples"] == 30 * 4 and n_idemix == 6


@pytest.mark.gpu
def test_preverify_pass_with_idemix_creators():
    import json
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    raw_ipk = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]["MSP1OU1"]["ipk"])
    csp = fabgpu.GPUCSP(device=0)
    rng = np.random.default_rng(9)
    blk, want, n_idemix = build_mixed_block(120, rng)
    # before the MSP is registered every idemix creator is left to bccsp/idemix
    out0 = fabgpu.preverify_block(csp, blk)
    idemix_tx = np.arange(120) % 5 == 0
    assert (out0["tx_flags"][idemix_tx] == fabgpu.TX_NEEDS_SW).all() and (out0["tx_flags"][~idemix_tx] == 0).all()
    assert csp.idemix_msp_register("IdemixMSP1", raw_ipk) >= 0
    out = fabgpu.preverify_block(csp, blk)
    assert (out["tx_flags"] == want).all(), list(zip(np.nonzero(out["tx_flags"] != want)[0], out["tx_flags"][out["tx_flags"] != want]))
    creators = out["tuple_kind"] == fabgpu.TUPLE_CREATOR if hasattr(fabgpu, "TUPLE_CREATOR") else out["tuple_kind"] == 0
    assert (out["tuple_status"][~creators] == 0).all()
    assert (out["tx_flags"] == fabgpu.TX_BAD_CREATOR_SIGNATURE).sum() == sum(1 for t in range(0, 120, 5) if (t // 5) % 6 == 2)
    assert fabgpu.preverify_block(csp, blk)["tx_flags"].tolist() == want.tolist()          # repeatable; pseudonyms are not cached
    # the verdict memo also carries the pseudonym signatures: key = Nym.x || Nym.y, digest = SHA-256(message) - what the Go
    # NymVerifier wrapper recomputes from ITS arguments (fabric-mod_amd/go/bccsp/idemixgpu/nymverifier.go)
    r2 = fabgpu.preverify_block2(csp, blk, block_seq=9, seed_memo=True)
    assert (r2["tx_flags"] == want).all()
    from idemix_common import fixtures
    ipk_hash, other_hash = bytes(fixtures()["MSP1OU1"]["ipk"].hash), bytes(fixtures()["MSP2OU1"]["ipk"].hash)
    assert len(ipk_hash) == 32 and ipk_hash != other_hash
    n_nym = 0
    for i in np.nonzero(r2["tuple_kind"] == 0)[0]:
        tx = int(r2["tuple_tx"][i])
        if tx % 5 != 0 or r2["tuple_status"][i] not in (0, 1):
            continue
        sp = [int(x) for x in r2["tuple_spans"][i]]
        msg, sig = r2["arena"][sp[4]:sp[4] + sp[5]], r2["arena"][sp[6]:sp[6] + sp[7]]
        assert bytes(r2["tuple_digest"][i]) == hashlib.sha256(msg).digest() and r2["tuple_hashed"][i]
        qxy = bytes(r2["tuple_qxy"][i])
        # ... bound to the issuer key it was verified under (ipk.Hash), and in a key domain of its own: the ECDSA lookup with the same
        # bytes misses, another issuer's hash misses (two channels may name their idemix MSPs alike - ADVICE r2)
        assert fabgpu.memo_lookup_nym(csp, ipk_hash, qxy[:32], qxy[32:], sig, hashlib.sha256(msg).digest()) == int(r2["tuple_status"][i])
        assert fabgpu.memo_lookup_nym(csp, ipk_hash, qxy[:32], qxy[32:], sig, hashlib.sha256(msg + b"!").digest()) is None
        assert fabgpu.memo_lookup_nym(csp, other_hash, qxy[:32], qxy[32:], sig, hashlib.sha256(msg).digest()) is None
        assert fabgpu.memo_lookup(csp, qxy[:32], qxy[32:], sig, hashlib.sha256(msg).digest()) is None
        n_nym += 1
    assert n_nym >= 10
    assert fabgpu.memo_evict_block(csp, 9) == r2["memo_seeded"]
    # the same MSP id registered with ANOTHER issuer key (a second channel's "IdemixMSP1"): the pass can no longer know under which key a
    # creator of that MSP id verifies - it leaves them to bccsp/idemix
    raw_ipk2 = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]["MSP2OU1"]["ipk"])
    assert csp.idemix_msp_register("IdemixMSP1", raw_ipk2) >= 0
    amb = fabgpu.preverify_block(csp, blk)
    assert (amb["tx_flags"][idemix_tx] == fabgpu.TX_NEEDS_SW).all() and (amb["tx_flags"][~idemix_tx] == 0).all()
    csp.close()
    # With channels named, a channel's latest key replaces its earlier one: a second channel that disagrees makes the MSP id ambiguous,
    # and the ambiguity ends when that channel's config moves to the same key (or the first channel rotates to the other's)
    csp = fabgpu.GPUCSP(device=0)
    assert csp.idemix_msp_register("IdemixMSP1", raw_ipk, channel="ch1") >= 0
    assert (fabgpu.preverify_block(csp, blk)["tx_flags"] == want).all()
    assert csp.idemix_msp_register("IdemixMSP1", raw_ipk2, channel="ch2") >= 0
    amb = fabgpu.preverify_block(csp, blk)
    assert (amb["tx_flags"][idemix_tx] == fabgpu.TX_NEEDS_SW).all() and (amb["tx_flags"][~idemix_tx] == 0).all()
    assert csp.idemix_msp_register("IdemixMSP1", raw_ipk, channel="ch2") >= 0
    assert (fabgpu.preverify_block(csp, blk)["tx_flags"] == want).all()
    # rotation inside the one channel left: its creators now verify under the NEW key - these, signed under the old one, fail
    assert csp.idemix_msp_register("IdemixMSP1", raw_ipk2, channel="ch1") >= 0 and csp.idemix_msp_register("IdemixMSP1", raw_ipk2, channel="ch2") >= 0
    rot = fabgpu.preverify_block(csp, blk)
    # (... those the nym kernel decides; the ones the block leaves to bccsp/idemix stay there)
    assert (rot["tx_flags"][idemix_tx] == np.where(want[idemix_tx] == fabgpu.TX_NEEDS_SW, fabgpu.TX_NEEDS_SW, fabgpu.TX_BAD_CREATOR_SIGNATURE)).all()
    assert (rot["tx_flags"][~idemix_tx] == 0).all()
    # The Hash field of a marshalled issuer key is not trusted (the reference recomputes it: idemix/issuerkey.go:171-182): a key whose
    # field 10 says something else is not accelerated
    assert raw_ipk[-34] == 0x52 and raw_ipk[-33] == 0x20
    forged = raw_ipk[:-1] + bytes([raw_ipk[-1] ^ 1])
    assert csp.idemix_msp_register("IdemixMSP9", forged, channel="ch1") == -1
    csp.close()


@pytest.mark.gpu
def test_preverify_pass_with_the_block_uploaded_ahead(monkeypatch):
    """Blocks of 4 MiB and more travel to the device on a helper thread while they are parsed (fabgpu_arena_stage); the test
    blocks are smaller, so the threshold is lowered: same flags either way, corrupted transactions included."""
    csp = fabgpu.GPUCSP(device=0)
    rng = np.random.default_rng(16)
    blk, want = build_block(130, rng)
    plain = fabgpu.preverify_block(csp, blk)
    csp.set_option("pass_stage_min_bytes", 1)
    staged = fabgpu.preverify_block(csp, blk)
    assert (plain["tx_flags"] == want).all() and (staged["tx_flags"] == want).all()
    assert (staged["tuple_status"] == plain["tuple_status"]).all()
    for _ in range(3):                                                       # uploads replace each other; every pass still answers
        assert (fabgpu.preverify_block(csp, blk)["tx_flags"] == want).all()
    csp.close()


def test_threaded_walk_on_a_multi_megabyte_block():
    """BlockData of 1 MiB and more is listed on the calling thread while worker threads parse the envelopes chunk by chunk (256 per
    chunk): 700 transactions (3 chunks, 3.5 MB), every count and every hash check must come out as for a small block."""
    rng = np.random.default_rng(23)
    sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS if i["curve"] == "prime256v1"]
    fake = b"\x30\x44\x02\x20" + b"\x11" * 32 + b"\x02\x20" + b"\x22" * 32          # signatures are not looked at by the walker
    envs = []
    for t in range(700):
        payload, _ = bb.consistent_endorser_tx("mychannel", sid[4 + t % 2], bytes(rng.integers(0, 256, size=24, dtype=np.uint8)),
                                               bytes(rng.integers(0, 256, size=300, dtype=np.uint8)), bytes(rng.integers(0, 256, size=900, dtype=np.uint8)),
                                               lambda prp: [(sid[j], fake) for j in (0, 1, 2)], bad_txid=(t % 97 == 5), bad_phash=(t % 89 == 7))
        envs.append(bb.envelope(payload, fake))
    blk = bb.block(3, envs)
    assert len(blk) > 3 << 20
    p = fabgpu.block_parse(blk)
    assert p["n_tx"] == 700 and p["n_tuples"] == 2800 and p["n_prefixes"] == 700 and p["channel_id"] == "mychannel"
    assert (np.asarray(p["tx_type"]) == 3).all()
    checks = fabgpu.block_hash_checks(blk)
    assert [c[0] for c in checks] == [t for t in range(700) for _ in range(2)]           # in transaction order: TxID, then proposal hash
    bad = set()
    for tx, kind, pieces, (e0, e1) in checks:
        digest = hashlib.sha256(b"".join(blk[a:b] for a, b in pieces)).digest()
        if not ((blk[e0:e1] == digest.hex().encode()) if kind == 0 else (blk[e0:e1] == digest)):
            bad.add((tx, kind))
    assert bad == {(t, 0) for t in range(700) if t % 97 == 5} | {(t, 1) for t in range(700) if t % 89 == 7}


def test_concurrent_walks_share_the_worker_pool_or_fall_back():
    """Two channels validate at once: the walk's persistent worker pool (csrc/worker_pool.h) serves one job at a time, the second caller
    runs on threads of its own.  Four Python threads (ctypes releases the GIL) walk multi-megabyte blocks concurrently, many times."""
    import threading
    rng = np.random.default_rng(5)
    sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in IDS if i["curve"] == "prime256v1"]
    fake = b"\x30\x44\x02\x20" + b"\x11" * 32 + b"\x02\x20" + b"\x22" * 32
    blocks = []
    for ntx in (300, 450, 600, 750):
        envs = []
        for t in range(ntx):
            payload, _ = bb.consistent_endorser_tx("mychannel", sid[4 + t % 2], bytes(rng.integers(0, 256, size=24, dtype=np.uint8)),
                                                   bytes(rng.integers(0, 256, size=300, dtype=np.uint8)), bytes(rng.integers(0, 256, size=900, dtype=np.uint8)),
                                                   lambda prp: [(sid[j], fake) for j in (0, 1, 2)])
            envs.append(bb.envelope(payload, fake))
        blocks.append((ntx, bb.block(7, envs)))
    assert all(len(b) > 1 << 20 for _, b in blocks)                    # the threaded walk
    errors = []

    def work(ntx, blk):
        try:
            for _ in range(25):
                p = fabgpu.block_parse(blk)
                assert p["n_tx"] == ntx and p["n_tuples"] == 4 * ntx and p["n_prefixes"] == ntx
        except Exception as e:                                        # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work, args=b) for b in blocks]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors and not any(t.is_alive() for t in th)


@pytest.mark.gpu
def test_concurrent_passes_on_one_provider():
    """A peer with several channels validates their blocks at once through ONE BCCSP: three threads run the pass (with memo seeding and
    eviction) over different blocks on the same provider, many times; every answer must be the block's own."""
    import threading
    csp = fabgpu.GPUCSP(device=0)
    rng = np.random.default_rng(77)
    blocks = [build_block(n, rng) for n in (90, 260, 1100)]            # the last one is big enough for the threaded walk and gates
    solo = [fabgpu.preverify_block2(csp, blk, block_seq=10 + i) for i, (blk, _) in enumerate(blocks)]
    for (blk, want), ref in zip(blocks, solo):
        assert (ref["tx_flags"] == want).all()
    errors = []

    def work(i):
        blk, want = blocks[i]
        try:
            for k in range(12):
                seq = 1000 * (i + 1) + k
                out = fabgpu.preverify_block2(csp, blk, block_seq=seq, seed_memo=True)
                assert (out["tx_flags"] == want).all()
                for f in ("tuple_status", "tuple_digest", "tuple_hashed", "tuple_tx"):
                    assert (out[f] == solo[i][f]).all(), f
                # one verdict of this block, looked up the way bccsp.Verify would
                j = int(np.flatnonzero(out["tuple_hashed"])[k % int(out["tuple_hashed"].sum())])
                sp = out["tuple_spans"][j]
                sig = out["arena"][int(sp[6]):int(sp[6]) + int(sp[7])]
                got = fabgpu.memo_lookup(csp, bytes(out["tuple_qxy"][j][:32]), bytes(out["tuple_qxy"][j][32:]), sig, bytes(out["tuple_digest"][j]))
                assert got == int(out["tuple_status"][j]), (got, int(out["tuple_status"][j]))
                fabgpu.memo_evict_block(csp, seq)
        except Exception as e:                                        # noqa: BLE001
            import traceback
            errors.append(traceback.format_exc())
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors[0]
    assert not any(t.is_alive() for t in th)
    csp.close()


def test_walkers_survive_mutated_input():
    """The host-side parsers read untrusted network bytes: a few thousand mutants of a valid block and of a valid certificate must
    neither crash the process nor report a span outside the buffer.  (The thorough version runs under ASan/UBSan: tools/fuzz/run.sh.)"""
    rng = np.random.default_rng(41)
    blk, _ = build_block(12, rng, corrupt=False)
    base = np.frombuffer(blk, dtype=np.uint8)
    for it in range(1500):
        m = base.copy()
        for _ in range(int(rng.integers(1, 8))):
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
        raw = m.tobytes()
        try:
            p = fabgpu.block_parse(raw)
        except fabgpu.FabgpuError:
            continue                                       # outer framing broken: refused, fine
        assert p["n_tuples"] <= 4 * p["n_tx"] + 4
        for tx, kind, pieces, (e0, e1) in fabgpu.block_hash_checks(raw):
            assert all(0 <= a <= b <= len(raw) for a, b in pieces) and 0 <= e0 <= e1 <= len(raw)
    pem = IDS[0]["pem"].encode()
    for it in range(1500):
        b = bytearray(pem)
        for _ in range(3):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        got = fabgpu.x509_p256_pubkey(bytes(b))
        assert got is None or (len(got[0]) == 32 and len(got[1]) == 32)
