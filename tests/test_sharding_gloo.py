"""The N>1 path on CPU: world_size-2 gloo processes shard a flattened batch (fabgpu.sharding), each computes its
shard's verdict words (the CPU oracle stands in for the kernel here - this test is about the partition and the
exchange, not the arithmetic), one all-gather merges them, and every rank must hold the single-process bitmap."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pack(valid: np.ndarray, nwords: int) -> np.ndarray:
    bits = np.zeros(nwords * 64, dtype=np.uint8)
    bits[: valid.size] = valid
    return np.packbits(bits, bitorder="little").view("<u8").astype(np.uint64)


def _worker(rank, world, port, n, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "fabric-mod_amd")):
        sys.path.insert(0, p)
    import coracle
    from fabgpu import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    b = coracle.make_batch(n, seed=4242, invalid_frac=0.3)            # same block on every rank
    lo, hi = sharding.shard_range(n, rank, world)
    assert lo % 64 == 0
    sw = sharding.shard_words(n, world)
    st = coracle.verify_batch(b["qx"][lo:hi], b["qy"][lo:hi], b["e"][lo:hi], b["r"][lo:hi], b["s"][lo:hi]) if hi > lo else np.zeros(0, np.uint8)
    local = torch.from_numpy(_pack(st == 0, sw).view(np.int64).copy())
    merged = sharding.allgather_verdicts(local, n, world)
    q.put((rank, merged.numpy().view(np.uint64).copy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1000, 64, 130, 3750 * 2 + 17])
def test_two_rank_shard_and_allgather_matches_single_process(n):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = coracle.make_batch(n, seed=4242, invalid_frac=0.3)
    want = _pack(coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"]) == 0, (n + 63) // 64)
    ranges = sorted(r[2] for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b_[0] for a, b_ in zip(ranges, ranges[1:]))
    for rank, merged, _ in res:
        assert (merged == want).all(), rank


def test_shard_geometry():
    sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))
    from fabgpu import sharding
    # BASELINE cfg 3: 30 000 tuples over 8 GPUs -> 59 words per rank, shards of <= 3776 tuples
    assert sharding.shard_words(30000, 8) == 59
    cover = []
    for r in range(8):
        lo, hi = sharding.shard_range(30000, r, 8)
        assert lo % 64 == 0 and hi - lo <= 59 * 64
        cover.append((lo, hi))
    assert cover[0][0] == 0 and cover[-1][1] == 30000 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    for n in (0, 1, 63, 64, 65, 1000):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, w) for r in range(w)]
            assert sum(h - l for l, h in spans) == n


# ---- BASELINE config 5: a mixed batch (ECDSA + idemix pseudonym signatures) across ranks -------------------------------------
def _mixed_worker(rank, world, port, n_ec, n_nym, q):
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import coracle
    import idemix_oracle as io
    from fabgpu import sharding
    from idemix_common import fixtures, make_batch
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    b = coracle.make_batch(n_ec, seed=515, invalid_frac=0.2)              # the same block on every rank
    fx = fixtures()
    nb = make_batch([(fx[m]["ipk"], fx[m]["signer"].sk) for m in ("MSP1OU1", "MSP2OU1")], n_nym, 516)
    merged = []
    # the two sub-batches are sharded independently by the same rule: each kernel gets a contiguous, 64-aligned range
    lo, hi = sharding.shard_range(n_ec, rank, world)
    st = coracle.verify_batch(b["qx"][lo:hi], b["qy"][lo:hi], b["e"][lo:hi], b["r"][lo:hi], b["s"][lo:hi]) if hi > lo else np.zeros(0, np.uint8)
    merged.append(sharding.allgather_verdicts(torch.from_numpy(_pack(st == 0, sharding.shard_words(n_ec, world)).view(np.int64).copy()), n_ec, world))
    lo2, hi2 = sharding.shard_range(n_nym, rank, world)
    st2 = np.array([nb.expect[i] for i in range(lo2, hi2)], dtype=np.uint8)   # the oracle's verdicts stand in for the nym kernel
    merged.append(sharding.allgather_verdicts(torch.from_numpy(_pack(st2 == 0, sharding.shard_words(n_nym, world)).view(np.int64).copy()), n_nym, world))
    q.put((rank, [m.numpy().view(np.uint64).copy() for m in merged], (lo, hi), (lo2, hi2)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mixed_batch_of_config_5():
    for p in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import coracle
    from idemix_common import fixtures, make_batch
    n_ec, n_nym = 800, 200                                                 # 80 % / 20 %
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mixed_worker, args=(r, world, port, n_ec, n_nym, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = coracle.make_batch(n_ec, seed=515, invalid_frac=0.2)
    want_ec = _pack(coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"]) == 0, (n_ec + 63) // 64)
    fx = fixtures()
    nb = make_batch([(fx[m]["ipk"], fx[m]["signer"].sk) for m in ("MSP1OU1", "MSP2OU1")], n_nym, 516)
    want_nym = _pack(np.array(nb.expect) == 0, (n_nym + 63) // 64)
    for rank, (m_ec, m_nym), r1, r2 in res:
        assert (m_ec == want_ec).all() and (m_nym == want_nym).all(), rank
    assert sorted(r[2] for r in res)[0][0] == 0 and sorted(r[3] for r in res)[-1][1] == n_nym
