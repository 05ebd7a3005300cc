"""ONE provider over SEVERAL device contexts (fabgpu_csp_new2; fabric-mod_amd/csrc/bccsp_host.h GPUCSP::devs_, pass_route.h).

The reference has one process-global BCCSP (bccsp/factory/factory.go:41-55), handed to every channel's validator
(core/peer/peer.go:337-355), and validates channels side by side (core/committer/txvalidator/v20/validator.go:194-210): whatever
drives more than one GPU sits behind that one object.  The GPU box of the test pool has ONE MI355X, so the pool here is three
contexts on device 0 - everything but the PCIe links is as on an 8-GPU node: three sets of streams, staging slots, identity tables,
comb tables and predictions, one identity cache and one verdict memo on the host."""
import base64
import json
import os
import threading

import numpy as np
import pytest

import fabgpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ["tx_flags", "tx_type", "tuple_tx", "tuple_kind", "tuple_status", "tuple_spans", "tuple_digest", "tuple_hashed", "tuple_qxy"]


def _same(a, b, keys=KEYS):
    for k in keys:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_provider_options_struct_is_the_headers():
    """fabgpu_csp_opts as ctypes lays it out = as the C compiler lays it out (a program compiled against include/fabgpu_bccsp.h prints
    its sizeof and the offsets of its fields)."""
    import ctypes
    import subprocess
    import tempfile
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "fabgpu_bccsp.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(fabgpu_csp_opts), offsetof(fabgpu_csp_opts, size), offsetof(fabgpu_csp_opts, n_devices),
           offsetof(fabgpu_csp_opts, devices), offsetof(fabgpu_csp_opts, ctx_flags), offsetof(fabgpu_csp_opts, concurrent_passes),
           offsetof(fabgpu_csp_opts, expect_block_bytes), offsetof(fabgpu_csp_opts, expect_tuples), offsetof(fabgpu_csp_opts, pass_device_walk),
           offsetof(fabgpu_csp_opts, pass_stage_min_bytes), offsetof(fabgpu_csp_opts, pass_device_memo), offsetof(fabgpu_csp_opts, pass_host_counts),
           offsetof(fabgpu_csp_opts, pass_timing), offsetof(fabgpu_csp_opts, pass_hash_memo), offsetof(fabgpu_csp_opts, hash_memo_blocks));
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "o.c"), "w").write(src)
        subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(d, "o.c"), "-o", os.path.join(d, "o")], check=True)
        got = [int(x) for x in subprocess.run([os.path.join(d, "o")], capture_output=True, text=True, check=True).stdout.split()]
    O = fabgpu._CspOpts
    want = [ctypes.sizeof(O)] + [getattr(O, f).offset for f, _ in O._fields_]
    assert got == want, (got, want)


@pytest.fixture()
def pool():
    c = fabgpu.GPUCSP(devices=[0, 0, 0])
    c.set_option("pass_stage_min_bytes", 1)
    yield c
    c.close()


@pytest.mark.gpu
def test_passes_land_on_every_device_and_the_memo_does_not_care_which(pool):
    """Nine passes one after the other (nothing in flight: device = block_seq mod 3), then six callers at once: every context of the pool
    serves passes, registered keys have their tables on all three, answers equal a single-device provider's on the same blocks - corrupted
    transactions included - and a bccsp.Verify(k, sig, digest) lookup finds its verdict whichever device computed it."""
    import blockgen
    from test_device_walk import clean_modes_block
    assert pool.device_count() == 3 and pool.get_option("n_devices") == 3
    friendly, _ = blockgen.endorser_block(200, 7)
    one = fabgpu.GPUCSP(device=0)
    one.set_option("pass_stage_min_bytes", 1)
    try:
        for k in range(9):
            assert pool.route_block(k) == k % 3
            out = fabgpu.preverify_block2(pool, friendly, block_seq=k)
            assert (out["tx_flags"] == 0).all() and (out["tuple_status"] == 0).all()
        assert pool.passes_per_device() == [3, 3, 3]
        # the six signers earned their comb tables (64 namings) on the first passes: every device holds all six, and the passes of the
        # third round ran through them wherever they landed
        assert [pool.key_count(d) for d in range(3)] == [6, 6, 6]
        for k in range(9, 12):
            out = fabgpu.preverify_block2(pool, friendly, block_seq=k)
            assert out["n_keyed"] == 800 and out["n_device_decoded"] == 0, (k, out["n_keyed"], out["n_device_decoded"])
        # a block with every corruption the device decides: the pool's answer on each device = the single-device provider's
        rng = np.random.default_rng(101)
        blk, want = clean_modes_block(220, rng)
        ref = None
        for k in range(3):
            ref = fabgpu.preverify_block2(one, blk, block_seq=100 + k)          # (learning passes)
        assert (ref["tx_flags"] == want).all()
        before = pool.passes_per_device()
        for k in range(12, 18):
            got = fabgpu.preverify_block2(pool, blk, block_seq=k, seed_memo=(k >= 15))
            assert (got["tx_flags"] == want).all()
            if k >= 15:                                                          # (by then the identities are known everywhere: same launch classes)
                _same(ref, got)
        after = pool.passes_per_device()
        assert [a - b for a, b in zip(after, before)] == [2, 2, 2]
        # the memo: blocks 15, 16, 17 were verified on devices 0, 1, 2 - every lookup is answered from the one table
        assert all(fabgpu.memo_has_block(pool, k) > 0 for k in (15, 16, 17))
        hits = 0
        for i in np.nonzero(got["tuple_hashed"])[0][:60]:
            sp = [int(x) for x in got["tuple_spans"][i]]
            sig = got["arena"][sp[6]:sp[6] + sp[7]]
            st = fabgpu.memo_lookup(pool, bytes(got["tuple_qxy"][i][:32]), bytes(got["tuple_qxy"][i][32:]), sig, bytes(got["tuple_digest"][i]))
            assert st == int(got["tuple_status"][i])
            hits += 1
        assert hits == 60
        for k in (15, 16):
            assert fabgpu.memo_evict_block(pool, k) > 0
        assert fabgpu.memo_has_block(pool, 17) > 0 and fabgpu.memo_has_block(pool, 15) == 0
        # six callers at once (the channels of a peer): every pass answers, the load spreads
        before = pool.passes_per_device()
        errs = []

        def caller(t):
            try:
                for k in range(4):
                    r = fabgpu.preverify_block2(pool, bytes(bytearray(blk)), block_seq=1000 * (t + 1) + k, seed_memo=True, lean=True)
                    assert (r["tx_flags"] == want).all()
                    fabgpu.memo_evict_block(pool, 1000 * (t + 1) + k)
            except Exception as e:                                               # noqa: BLE001
                errs.append(repr(e))
        th = [threading.Thread(target=caller, args=(t,)) for t in range(6)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        served = [a - b for a, b in zip(pool.passes_per_device(), before)]
        assert sum(served) == 24 and min(served) >= 2, served   # (least busy first: about eight each; the bound leaves room for a slow context)
    finally:
        one.close()


@pytest.mark.gpu
def test_flat_batches_and_idemix_issuers_on_every_device_of_the_pool(pool):
    """What is not a block pass - CSP.Verify batches, identity.Verify batches, KeyImport, idemix issuers - goes round the pool: the same
    answers from every device, issuers and imported keys under one id everywhere."""
    import bccsp_sw_oracle as po
    from idemix_common import fixtures
    from test_block_prepass import build_mixed_block
    b = fabgpu.synth_batch(600, seed=11, invalid_permille=100)
    keys = [pool.key_import((int.from_bytes(b["qx"][i].tobytes(), "big"), int.from_bytes(b["qy"][i].tobytes(), "big"))) for i in range(4)]
    assert [pool.key_count(d) for d in range(3)] == [4, 4, 4]                    # KeyImport builds the table once and uploads it three times
    sigs = [po.marshal_ecdsa_signature(int.from_bytes(b["r"][i].tobytes(), "big"), int.from_bytes(b["s"][i].tobytes(), "big")) for i in range(600)]
    ks = [fabgpu.ECDSAPublicKey(int.from_bytes(b["qx"][i].tobytes(), "big"), int.from_bytes(b["qy"][i].tobytes(), "big")) for i in range(600)]
    want = [bool(k == 0) for k in b["kind"]]
    for _ in range(4):                                                            # four batches: round the ring, every context serves at least one
        got = pool.verify_batch(ks, sigs, [b["e"][i].tobytes() for i in range(600)])
        assert [v for v, _ in got] == want
    # an idemix MSP: its issuer tables on every device, creators verified wherever the pass lands
    raw_ipk = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]["MSP1OU1"]["ipk"])
    assert pool.idemix_msp_register("IdemixMSP1", raw_ipk) == 0
    L = pool._L
    assert [L.fabgpu_idemix_issuer_count(L.fabgpu_csp_ctx_of(pool._h, d)) for d in range(3)] == [1, 1, 1]
    rng = np.random.default_rng(9)
    blk, want_flags, n_idemix = build_mixed_block(120, rng)
    outs = []
    for k in range(6):                                                            # two passes per device (the first one there catches up with the nym launch)
        outs.append(fabgpu.preverify_block2(pool, blk, block_seq=k, seed_memo=True))
        assert (outs[-1]["tx_flags"] == want_flags).all(), k
    assert pool.passes_per_device() == [2, 2, 2]
    for o in outs[1:]:
        _same(outs[0], o)
    # the digest memo is the provider's, whichever device kept the block's copy: every message a pass hashed - the idemix creators'
    # payloads among them (their entries carry SHA-256(message) too) - is answered with hashlib's digest, six blocks' copies held on three devices
    import hashlib
    st = fabgpu.hash_memo_stats(pool)
    assert st["blocks_held"] == 6 and st["refused"] == 0
    o, asked = outs[-1], 0
    for i in range(len(o["tuple_status"])):
        if not o["tuple_hashed"][i] or o["tuple_status"][i] > 3:
            continue
        sp = [int(x) for x in o["tuple_spans"][i]]
        msg = o["arena"][sp[2]:sp[2] + sp[3]] + o["arena"][sp[4]:sp[4] + sp[5]]
        assert fabgpu.hash_lookup(pool, msg) == hashlib.sha256(msg).digest() == bytes(o["tuple_digest"][i])
        asked += 1
    assert asked > 300
    ipk_hash = bytes(fixtures()["MSP1OU1"]["ipk"].hash)
    o = outs[-1]                                                                   # (device 2's pass)
    n = 0
    for i in np.nonzero(o["tuple_kind"] == 0)[0]:
        if int(o["tuple_tx"][i]) % 5 != 0 or o["tuple_status"][i] not in (0, 1):
            continue
        sp = [int(x) for x in o["tuple_spans"][i]]
        sig = o["arena"][sp[6]:sp[6] + sp[7]]
        q = bytes(o["tuple_qxy"][i])
        assert fabgpu.memo_lookup_nym(pool, ipk_hash, q[:32], q[32:], sig, bytes(o["tuple_digest"][i])) == int(o["tuple_status"][i])
        n += 1
    assert n > 10


@pytest.mark.gpu
def test_what_overlapping_passes_need_is_there_when_the_provider_is_made():
    """ProviderOptions::concurrent_passes (GPUOpts.ConcurrentPasses): staging slots, pinned memo tables and pass arrays are allocated at
    construction - the first overlapping, memo-seeding passes of a provider cost what later ones cost (round 3: 5-16 ms each for the
    first three or four, once per provider; a peer that joins a channel gets no untimed rounds)."""
    import time

    import blockgen
    blk, _ = blockgen.endorser_block(1500, 21)
    csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=2, expect_block_bytes=len(blk) + 4096, expect_tuples=6200)
    try:
        for k in range(6):                                                        # one caller: the signers are learned and earn their tables
            fabgpu.preverify_block2(csp, blk, block_seq=k, lean=True)
        per = [[], []]

        def caller(t):
            for k in range(6):
                b = bytes(bytearray(blk))
                c0 = time.perf_counter()
                r = fabgpu.preverify_block2(csp, b, block_seq=100 * (t + 1) + k, seed_memo=True, lean=True)
                per[t].append((time.perf_counter() - c0) * 1e3)
                assert (r["tx_flags"] == 0).all() and r["memo_seeded"] == 6000
                fabgpu.memo_evict_block(csp, 100 * (t + 1) + k)
        th = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        first = max(per[0][0], per[1][0])
        steady = float(np.median(per[0][2:] + per[1][2:]))
        # (a loose bound: the boxes are shared; what it catches is the 5-16 ms of a provider that allocates when passes first overlap)
        assert first < 3.0 * steady + 1.0, (first, steady, per)
    finally:
        csp.close()
