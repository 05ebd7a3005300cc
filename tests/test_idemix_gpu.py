"""GPU parity of the idemix pseudonym-signature kernel (fabgpu_idemix_nym_verify_batch) against oracle/idemix_oracle.py:
status-exact on seeded batches that mix valid signatures with every kind of invalid / out-of-domain input, over two
issuers of the reference's fixtures, every message-length class of the fused SHA-256, and a 30 000-signature batch checked
through replication (the verdict of a copy is the verdict of its original)."""
import os

import numpy as np
import pytest

import fabgpu
import idemix_oracle as io
from idemix_common import be32, fixtures, make_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "four-lanes-one-stream", "four-lanes-fused", "two-lanes", "one-lane"])
def env(request):
    """auto: batches up to 16 384 run on the four-lanes-per-signature kernel (bn_quad29.h), up to 32 768 on the two-lanes one, larger
    ones on the one-lane kernel; two-lanes = FABGPU_FLAG_NO_QUAD, one-lane = FABGPU_FLAG_ONE_LANE_ONLY.  The four-lane form runs in two
    phases since round 5 (commitments, then the challenges with eight lanes on a message); four-lanes-fused =
    FABGPU_FLAG_NYM_FUSED_HASH is round 4's single kernel; by default the fixed-base terms run as a launch of their own on a second stream
    beside the commitments, four-lanes-one-stream = FABGPU_FLAG_NYM_NO_SIDE_STREAM keeps them inside.  Every case goes through all five."""
    fx = fixtures()
    extra = {"auto": 0, "four-lanes-one-stream": fabgpu.FLAG_NYM_NO_SIDE_STREAM, "four-lanes-fused": fabgpu.FLAG_NYM_FUSED_HASH, "two-lanes": fabgpu.FLAG_NO_QUAD, "one-lane": fabgpu.FLAG_ONE_LANE_ONLY}[request.param]
    ctx = fabgpu.Context(device=0, flags=fabgpu.FLAG_TIME_KERNELS | extra)
    issuers = []
    for name in ("MSP1OU1", "MSP2OU1"):
        ipk = fx[name]["ipk"]
        iid = ctx.idemix_issuer_register((be32(ipk.h_sk[0]), be32(ipk.h_sk[1])), (be32(ipk.h_rand[0]), be32(ipk.h_rand[1])), ipk.hash)
        assert iid == len(issuers)
        issuers.append((ipk, fx[name]["signer"].sk))
    yield ctx, issuers
    ctx.close()


def _run(ctx, batch, with_issuer=True):
    arena, off, iid, cols, expect = batch.arrays()
    ok, st = ctx.idemix_nym_verify_batch(arena, off, *cols, issuer_id=iid if with_issuer else None)
    return ok, st, expect


def test_issuer_registration_is_idempotent_and_gated(env):
    ctx, issuers = env
    ipk = issuers[0][0]
    again = ctx.idemix_issuer_register((be32(ipk.h_sk[0]), be32(ipk.h_sk[1])), (be32(ipk.h_rand[0]), be32(ipk.h_rand[1])), ipk.hash)
    assert again == 0 and ctx.idemix_issuer_count() == 2
    with pytest.raises(fabgpu.FabgpuError):
        ctx.idemix_issuer_register((be32(ipk.h_sk[0]), be32((ipk.h_sk[1] + 1) % io.P)), (be32(ipk.h_rand[0]), be32(ipk.h_rand[1])), ipk.hash)


@pytest.mark.parametrize("n,seed", [(1, 1), (63, 2), (64, 3), (65, 4), (257, 5), (700, 6)])
def test_status_exact_against_oracle(env, n, seed):
    ctx, issuers = env
    b = make_batch(issuers, n, seed)
    ok, st, expect = _run(ctx, b)
    bad = np.nonzero(st != expect)[0]
    assert bad.size == 0, [(int(i), b.what[i], int(st[i]), int(expect[i]), len(b.msgs[i])) for i in bad[:8]]
    assert np.array_equal(ok, expect == 0)


def test_default_issuer_and_unknown_issuer(env):
    ctx, issuers = env
    b = make_batch(issuers[:1], 40, 9, tamper=False)
    ok, st, expect = _run(ctx, b, with_issuer=False)          # NULL issuer ids = issuer 0
    assert np.array_equal(st, expect) and ok.all()
    arena, off, iid, cols, _ = b.arrays()
    iid[::3] = 7                                              # not registered
    ok, st = ctx.idemix_nym_verify_batch(arena, off, *cols, issuer_id=iid)
    assert (st[::3] == io.NYM_NEEDS_SW).all() and not ok[::3].any()
    mask = np.ones(len(st), bool)
    mask[::3] = False
    assert (st[mask] == 0).all()


def test_every_message_length_class(env):
    """header (166 bytes) + message: exercises the third block shared between header tail and message, block boundaries, padding"""
    import random
    ctx, issuers = env
    ipk, sk = issuers[0]
    rng = random.Random(21)
    from idemix_common import NymBatch
    b = NymBatch()
    nym, r_nym = io.make_nym(sk, ipk, rng)
    for ln in list(range(0, 140)) + [154, 155, 217, 218, 219, 1000, 4608]:
        msg = bytes(rng.getrandbits(8) for _ in range(ln))
        b.add(0, ipk, nym, io.nym_sign(sk, nym, r_nym, ipk, msg, rng), msg)
    ok, st, expect = _run(ctx, b)
    assert (expect == 0).all()
    assert np.array_equal(st, expect), [len(b.msgs[i]) for i in np.nonzero(st != expect)[0]]


def test_side_launch_that_comes_too_late_changes_nothing(env, request):
    """Three launches (round 5): the fixed-base terms run on a side stream beside the commitments; a commitment wavefront that does not
    find its records gives up (flag 0 -> 2) and computes the terms itself, and a side wavefront that finds the 2 skips its rows.  The
    test hook orders the side launch BEHIND the commitment launch, so every wavefront takes exactly that path - statuses must not move."""
    ctx, issuers = env
    ctx.test_nym_side_after(True)
    try:
        for n, seed in ((1, 11), (65, 12), (700, 13), (6000, 14)):
            if n <= 700:
                b = make_batch(issuers, n, seed)
                ok, st, expect = _run(ctx, b)
            else:
                base = make_batch(issuers, 200, seed)
                arena, off, iid, cols, exp0 = base.arrays()
                pick = np.random.default_rng(n).integers(0, 200, size=n)
                off2 = np.zeros(n + 1, dtype=np.uint32)
                off2[1:] = np.cumsum((off[1:] - off[:-1])[pick])
                arena2 = np.concatenate([arena[off[i]:off[i + 1]] for i in pick])
                ok, st = ctx.idemix_nym_verify_batch(arena2, off2, *[c[pick] for c in cols], issuer_id=iid[pick])
                expect = exp0[pick]
            assert np.array_equal(st, expect) and np.array_equal(ok, expect == 0), n
    finally:
        ctx.test_nym_side_after(False)


def test_block_sized_batch_by_replication(env):
    """the four-lane limit 16 384 and its neighbours, 30 000 (a block), the two-lane limit 32 768 and its neighbours, several rounds"""
    ctx, issuers = env
    base = make_batch(issuers, 300, 17)
    arena, off, iid, cols, expect = base.arrays()
    for n in (5000, 16383, 16384, 16385, 30000, 32767, 32768, 32769, 70000):
        rng = np.random.default_rng(n)
        pick = rng.integers(0, 300, size=n)
        lens = (off[1:] - off[:-1])[pick]
        off2 = np.zeros(n + 1, dtype=np.uint32)
        off2[1:] = np.cumsum(lens)
        arena2 = np.concatenate([arena[off[i]:off[i + 1]] for i in pick])
        ok, st = ctx.idemix_nym_verify_batch(arena2, off2, *[c[pick] for c in cols], issuer_id=iid[pick])
        assert np.array_equal(st, expect[pick]), n
        assert np.array_equal(ok, expect[pick] == 0), n
    assert ctx.last_kernel_ms() > 0


def test_pseudonym_signatures_ride_in_an_identity_batch(env):
    """fabgpu_identity_batch.n_nym (ABI v4): the idemix creators of a block in the same submission as its ECDSA signatures - messages
    addressed as spans of the one arena, the nym kernel on a second stream.  Statuses must be those of the stand-alone entry point
    (which the tests above hold against the oracle), the ECDSA answers must not notice the riders."""
    import hashlib
    import bccsp_sw_oracle as po
    ctx, issuers = env
    nb = make_batch(issuers, 150, 31)
    narena, noff, iid, cols, expect = nb.arrays()
    rng = np.random.default_rng(32)
    d = 0x1234567
    qx, qy = po.pt_mul(d, (po.GX, po.GY))
    msgs = [bytes(rng.integers(0, 256, size=int(rng.integers(1, 900)), dtype=np.uint8)) for _ in range(90)]
    sigs = [po.sign_raw(d, hashlib.sha256(m).digest(), 5000 + j) for j, m in enumerate(msgs)]
    for j in range(0, 90, 7):
        msgs[j] = msgs[j] + b"?"                                   # tampered after signing
    # one arena: [pad][ECDSA messages][pad][pseudonym messages], everything addressed by (start, end)
    pad = b"\xee" * 37
    parts, espans = [pad], []
    at = len(pad)
    for m in msgs:
        espans.append((at, at + len(m)))
        parts.append(m)
        at += len(m)
    parts.append(pad)
    at += len(pad)
    base = at
    nspans = np.stack([noff[:-1] + base, noff[1:] + base], axis=1).astype(np.uint32)
    arena = np.frombuffer(b"".join(parts) + narena.tobytes() + b"\0" * 8, dtype=np.uint8)
    col = lambda v: np.frombuffer(b"".join(x.to_bytes(32, "big") for x in v), dtype=np.uint8).reshape(-1, 32)
    r, s_ = col([a for a, _ in sigs]), col([b for _, b in sigs])
    kq = (col([qx] * 90), col([qy] * 90))
    alone_bits, alone_st = ctx.identity_verify_batch(arena, np.array(espans, dtype=np.uint32), r, s_, qx=kq[0], qy=kq[1], spans=True)
    bits, st, (nok, nst) = ctx.identity_verify_batch(arena, np.array(espans, dtype=np.uint32), r, s_, qx=kq[0], qy=kq[1], spans=True,
                                                     nym=(nspans, iid, cols))
    assert (bits == alone_bits).all() and (st == alone_st).all() and bits.sum() == 90 - len(range(0, 90, 7))
    assert np.array_equal(nst, expect), [(int(i), nb.what[i], int(nst[i]), int(expect[i])) for i in np.nonzero(nst != expect)[0][:8]]
    assert np.array_equal(nok, expect == 0)
    # with the gathered digests in the same call, and with the arena staged ahead
    g = np.zeros((150, 6), dtype=np.uint32)
    g[:, 0:2] = nspans
    tok = ctx.arena_stage(arena)
    bits2, st2, dig, (nok2, nst2) = ctx.identity_verify_batch(arena, np.array(espans, dtype=np.uint32), r, s_, qx=kq[0], qy=kq[1], spans=True,
                                                              gather_spans=g, stage_token=tok, nym=(nspans, iid, cols))
    assert (bits2 == bits).all() and np.array_equal(nst2, expect)
    assert [bytes(x) for x in dig] == [hashlib.sha256(m).digest() for m in nb.msgs]


# ---- the host mirror of bccsp/idemix/handlers (NymVerifier.Verify), driven with the reference's vocabulary ------------------
def test_nym_verifier_mirror_semantics():
    import json
    import os
    import random
    from idemix_common import ROOT
    fx = fixtures()
    raw_ipk = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]["MSP1OU1"]["ipk"])
    ipk, sk = fx["MSP1OU1"]["ipk"], fx["MSP1OU1"]["signer"].sk
    csp = fabgpu.GPUCSP()
    try:
        iid = csp.idemix_issuer_import(raw_ipk)
        assert iid >= 0
        assert csp.idemix_issuer_import(raw_ipk) == iid                       # idempotent
        with pytest.raises(fabgpu.BCCSPError, match="invalid raw, it must not be nil"):
            csp.idemix_issuer_import(b"")
        rng = random.Random(31)
        nym, r_nym = io.make_nym(sk, ipk, rng)
        key = be32(nym[0]) + be32(nym[1])
        msg = b"creator payload bytes"
        sig = io.nym_sign(sk, nym, r_nym, ipk, msg, rng)
        good = io.nym_signature_marshal(sig)
        flipped = io.nym_signature_marshal(dict(sig, nonce=be32(int.from_bytes(sig["nonce"], "big") ^ 1)))
        short = io.nym_signature_marshal(dict(sig, proof_s_sk=sig["proof_s_sk"][:31]))
        reordered = (io.pb_bytes_field(4, sig["nonce"]) + io.pb_bytes_field(2, sig["proof_s_sk"]) + io.pb_bytes_field(1, sig["proof_c"])
                     + io.pb_bytes_field(3, sig["proof_s_r_nym"]))           # field order is free in protobuf
        cases = [
            (key, good, msg),            # 0 valid
            (key, good, msg + b"!"),     # 1 other message
            (key, flipped, msg),         # 2 tampered nonce
            (key, b"", msg),             # 3 empty signature
            (key, b"\xff\xff\xff", msg), # 4 bytes the walker does not accept: bccsp/idemix decides (and words the error)
            (key, short, msg),           # 5 a 31-byte field: amcl-internal -> bccsp/idemix decides
            (key[:63], good, msg),       # 6 odd-sized nym key: bccsp/idemix decides
            (b"", good, msg),            # 7 empty nym key: KeyImport error
            (key, reordered, msg),       # 8 valid
            (key, good + io.pb_bytes_field(9, b"unknown field"), msg),   # 9 unknown fields are skipped by proto.Unmarshal
        ]
        res = csp.idemix_nym_verify_batch(iid, [c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases])
        assert res[0] == (True, False, None)
        assert res[1] == (False, False, "pseudonym signature invalid: zero-knowledge proof is invalid")
        assert res[2] == (False, False, "pseudonym signature invalid: zero-knowledge proof is invalid")
        assert res[3] == (False, False, "invalid signature, it must not be empty")
        assert res[4] == (False, True, None)
        assert res[5] == (False, True, None)
        assert res[6] == (False, True, None)
        assert res[7] == (False, False, "invalid raw, it must not be nil")
        assert res[8] == (True, False, None)
        assert res[9] == (True, False, None)
        # an issuer the device does not know: everything is left to bccsp/idemix
        res = csp.idemix_nym_verify_batch(-1, [key], [good], [msg])
        assert res[0] == (False, True, None)
    finally:
        csp.close()
