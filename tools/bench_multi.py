#!/usr/bin/env python3
"""The product library's own multi-GPU path (fabgpu_multi_*: ONE process, G contexts, RCCL all-gather of the verdict bitmaps) on
BASELINE.json configs[2] - "same 10k x 3 block sharded across 8 MI355X" - and on a batch ten times larger, where sharding can pay.
Host pointers in, bitmap out: every figure is PCIe-inclusive (staging + H2D + kernels + all-gather + D2H), wall clock around the
blocking C-ABI call.  Prints ONE JSON line.  bench.py (rank 0) runs this in a subprocess when the driver launches N > 1 ranks, so
that the in-process dispatcher is measured on the same 8-GPU node; it is also the tool for a manual run.  The oracle is not used:
verdicts are checked against the generator's ground truth (the oracle-vs-generator equality is a GPU test)."""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--host-merge", action="store_true")
    ap.add_argument("--shard-of", type=int, default=0, help="also time rank 0's shard of the 30000-tuple block as fabgpu_multi_plan cuts it for this many devices "
                    "(bench.py's shard_of_8 leg: what one GPU of eight would be handed, on the devices there are)")
    ap.add_argument("--devices", default=None, help="comma-separated ordinals (default 0..gpus-1); a dry run on a smaller box repeats ordinals")
    args = ap.parse_args()
    import numpy as np

    import fabgpu
    t0 = time.perf_counter()
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    m = fabgpu.MultiContext(devices, host_merge=args.host_merge)
    init_s = time.perf_counter() - t0
    ranks, why = m.collective()
    out = {"tool": "tools/bench_multi.py", "n_gpus": len(devices), "devices": devices,
           "merge": "RCCL ncclAllGather (in-process, ncclCommInitAll)" if ranks else "host D2H x G",
           "collective": "rccl" if ranks else "host_merge", "rccl_ranks": ranks, "collective_why": why,
           "init_s": init_s, "legs": []}
    for n, label in ((30000, "BASELINE.json configs[2]: one 10k x 3 block (30000 tuples) cut into %d shards" % len(devices)),
                     (300000, "300000 tuples (ten blocks' worth) cut into %d shards" % len(devices))):
        b = fabgpu.synth_batch(n, seed=20260921, invalid_permille=10)
        for _ in range(3):
            bits, _ = m.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"], want_status=False)
        wall = []
        for _ in range(args.iters):
            c0 = time.perf_counter()
            bits, _ = m.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"], want_status=False)
            wall.append((time.perf_counter() - c0) * 1e3)
        assert (bits == (b["kind"] == 0)).all(), "merged bitmap differs from the generator's ground truth"
        med = statistics.median(wall)
        out["legs"].append({"workload": label, "tuples": n, "value": n / (med * 1e-3), "unit": "verifies/s", "median_ms": med,
                            "p95_ms": sorted(wall)[int(0.95 * (len(wall) - 1))], "min_ms": min(wall), "iters": len(wall),
                            "shards": fabgpu.multi_plan(n, len(devices))[0][:2] + ["..."], "parity": "bit-identical to the ground truth"})
    if args.shard_of > 1:
        n = 30000
        (lo, hi) = fabgpu.multi_plan(n, args.shard_of)[0][0]
        b = fabgpu.synth_batch(n, seed=20260921, invalid_permille=10)
        f = [b[k][lo:hi] for k in ("qx", "qy", "e", "r", "s")]
        for _ in range(3):
            bits, _ = m.p256_verify_batch(*f, want_status=False)
        wall = []
        for _ in range(args.iters):
            c0 = time.perf_counter()
            bits, _ = m.p256_verify_batch(*f, want_status=False)
            wall.append((time.perf_counter() - c0) * 1e3)
        assert (bits == (b["kind"][lo:hi] == 0)).all(), "shard bitmap differs from the generator's ground truth"
        med = statistics.median(wall)
        out["legs"].append({"workload": "rank 0's shard of the 30000-tuple block cut for %d devices, through this process's %d device(s)" % (args.shard_of, len(devices)),
                            "shard_of": args.shard_of, "tuples": int(hi - lo), "value": (hi - lo) / (med * 1e-3), "unit": "verifies/s", "median_ms": med,
                            "min_ms": min(wall), "iters": len(wall), "parity": "bit-identical to the ground truth"})
    m.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
