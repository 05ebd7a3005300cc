"""One batch over G devices (BASELINE.json configs[2]): the product's multi-device dispatcher (fabgpu_multi_*).

CPU: the shard plan (count mode == fabgpu.sharding's ranges; bytes mode balances message bytes), and the FAKE backend - G host threads
running the kernel's verification core + a memcpy all-gather (libfabgpu_hosttest.so) - against the oracle.
GPU (one MI355X): G = 1 through RCCL (communicator of one rank, ncclAllGather), and G = 3 shards on the same device with the host
merge (RCCL refuses one device twice), both bit-exact against the oracle and the single-context C ABI.  The 8-GPU run is the driver's."""
import ctypes
import os

import numpy as np
import pytest

import coracle
import fabgpu
from fabgpu import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8p = ctypes.POINTER(ctypes.c_uint8)


@pytest.mark.parametrize("n,G", [(30000, 8), (30000, 1), (64, 2), (65, 2), (1, 8), (130, 3), (3750 * 2 + 17, 2), (300000, 8), (0, 4)])
def test_count_plan_equals_the_torch_sharding_module(n, G):
    ranges, wpr = fabgpu.multi_plan(n, G)
    assert wpr == sharding.shard_words(n, G)
    assert ranges == [sharding.shard_range(n, g, G) for g in range(G)]
    assert all(lo % 64 == 0 or lo == n for lo, _ in ranges)
    assert sum(hi - lo for lo, hi in ranges) == n
    if n == 30000 and G == 8:
        assert ranges[0] == (0, 3776) and wpr == 59      # "shards of 3 750" rounded up to whole words


def test_bytes_plan_balances_message_bytes_at_word_granularity():
    rng = np.random.default_rng(3)
    n, G = 30000, 8
    # 1 in 10 messages is 20x longer (creator payloads among endorsement messages)
    lens = np.where(rng.random(n) < 0.1, 40000, 2000).astype(np.uint64)
    lens[:3000] = 300                                                     # and a light head
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    ranges, wpr = fabgpu.multi_plan(n, G, off)
    assert ranges[0][0] == 0 and ranges[-1][1] == n
    for (a, b), (c, d) in zip(ranges, ranges[1:]):
        assert b == c and a % 64 == 0 and (b % 64 == 0 or b == n)
    per = [int(off[hi]) - int(off[lo]) for lo, hi in ranges]
    byc = [int(off[hi]) - int(off[lo]) for lo, hi in fabgpu.multi_plan(n, G)[0]]
    assert max(per) < 1.1 * sum(per) / G and min(per) > 0.9 * sum(per) / G     # within a word or so of the even share ...
    assert max(byc) > 1.1 * sum(byc) / G and min(byc) < 0.5 * sum(byc) / G     # ... where equal counts leave one device a third of the work
    assert all((hi - lo + 63) // 64 <= wpr for lo, hi in ranges)
    # uniform lengths: bytes mode covers the batch with no shard above the cap
    off2 = (np.arange(n + 1, dtype=np.uint64) * 1856).astype(np.uint32)
    r2, w2 = fabgpu.multi_plan(n, G, off2)
    assert sum(hi - lo for lo, hi in r2) == n and all((hi - lo + 63) // 64 <= w2 for lo, hi in r2)
    # degenerate: every byte in the last message
    off3 = np.zeros(n + 1, np.uint32)
    off3[-1] = 10 ** 6
    r3, w3 = fabgpu.multi_plan(n, G, off3)
    assert sum(hi - lo for lo, hi in r3) == n and all((hi - lo + 63) // 64 <= w3 for lo, hi in r3)


@pytest.mark.parametrize("n,G,by_bytes", [(500, 1, False), (500, 2, False), (777, 8, False), (64, 3, False), (900, 4, True)])
def test_fake_backend_g_host_threads_and_memcpy_allgather(n, G, by_bytes):
    L = ctypes.CDLL(os.path.join(ROOT, "fabric-mod_amd", "lib", "libfabgpu_hosttest.so"))
    b = coracle.make_batch(n, seed=99 + n, invalid_frac=0.25)
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    bits = np.zeros((n + 63) // 64, np.uint64)
    st = np.zeros(n, np.uint8)
    same = ctypes.c_int(0)
    off = None
    if by_bytes:
        lens = np.random.default_rng(n).integers(1, 5000, size=n).astype(np.uint64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    p = lambda a: a.ctypes.data_as(u8p)
    rc = L.hosttest_multi_verify(ctypes.c_size_t(n), G, off.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)) if off is not None else None, p(b["qx"]), p(b["qy"]),
                                 p(b["e"]), p(b["r"]), p(b["s"]), bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), p(st), ctypes.byref(same))
    assert rc == 0 and same.value == 1
    assert (st == want).all() and (fabgpu.unpack_bits(bits, n) == (want == 0)).all()


# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("devices,host_merge", [([0], False), ([0, 0, 0], True), ([0], True), ([0, 0], False)])
def test_multi_dispatcher_equals_the_oracle_and_the_single_context_abi(devices, host_merge):
    m = fabgpu.MultiContext(devices, host_merge=host_merge)
    try:
        assert m.device_count() == len(devices)
        ranks, why = m.collective()
        if host_merge:
            assert ranks == 0 and "HOST_MERGE" in why
        elif len(set(devices)) < len(devices):
            # a repeated ordinal cannot form a communicator: the library merges on the host instead of failing (round 5)
            assert ranks == 0 and "repeated" in why
            host_merge = True
        else:
            # the one-rank communicator of this box: RCCL's all-gather passed its one-word self-check at init - or the library says why not
            assert ranks in (0, len(devices)) and why
            assert ranks == len(devices), "RCCL unusable on this box: %s" % why
        for n in (1, 63, 64, 65, 1000, 30000):
            b = coracle.make_batch(n, seed=5150 + n, invalid_frac=0.2 if n > 4 else 0.0) if n < 30000 else fabgpu.synth_batch(n, seed=20260921, invalid_permille=10)
            want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
            bits, st = m.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
            assert (st == want).all() and (bits == (want == 0)).all(), (n, devices)
            bits2, none = m.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"], want_status=False)
            assert none is None and (bits2 == bits).all()
        # hash mode, ragged messages: shards are cut by bytes
        n = 2000
        rng = np.random.default_rng(8)
        lens = rng.integers(0, 3000, size=n)
        lens[::7] = 0
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        arena = rng.integers(0, 256, size=int(off[-1]) + 8, dtype=np.uint8)
        dig = coracle.sha256_batch(arena, off)
        c = fabgpu.synth_batch(n, seed=6, invalid_permille=100, e_in=dig)
        want = coracle.sha256_verify_batch(arena, off, c["qx"], c["qy"], c["r"], c["s"])
        bits, st = m.sha256_p256_verify_batch(arena, off, c["qx"], c["qy"], c["r"], c["s"])
        assert (st == want).all() and (bits == (want == 0)).all()
        if not host_merge:
            assert m.merged_bitmap_dev(0) != 0               # the all-gathered bitmap stays resident on the device
        # argument errors are infrastructure errors, never verdicts
        z = np.zeros((4, 32), np.uint8)
        bad_off = np.array([0, 5, 3, 9, 12], np.uint32)
        with pytest.raises(fabgpu.FabgpuError):
            m.sha256_p256_verify_batch(np.zeros(16, np.uint8), bad_off, z, z, z, z)
    finally:
        m.close()


@pytest.mark.gpu
def test_multi_init_argument_errors():
    with pytest.raises(fabgpu.FabgpuError):
        fabgpu.MultiContext([9999])
    with pytest.raises(fabgpu.FabgpuError):
        fabgpu.MultiContext([])
