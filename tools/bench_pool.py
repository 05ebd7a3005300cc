#!/usr/bin/env python3
"""BASELINE's second metric - validated tx/s per block - on EVERY GPU of the node through ONE provider (fabgpu_csp_new2: what the
reference's one process-global BCCSP becomes, bccsp/factory/factory.go:41-55): one GPUCSP over G device contexts, 2 G callers (the
channels of a peer with the arrival pipeline: two passes in flight per device) each submitting fresh copies of the 10 000-transaction
friendly block with memo seeding + eviction - exactly what go/extensions/gossip/state/preverify_on_arrival.go and
go/extensions/validation/preverify.go do.  Prints ONE JSON line: aggregate validated tx/s, passes per device (the provider's routing),
the same with ONE device beside it (scaling inside one process), first pass latencies.  bench.py (rank 0) runs this in a subprocess
when the driver launches N > 1 ranks; --devices 0,0,0 runs three contexts on one GPU (the 1-GPU test boxes)."""
import argparse
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]


def run(fabgpu, np, blk, devices, callers, per_caller, n_tx, memo=True):
    csp = fabgpu.GPUCSP(devices=devices, concurrent_passes=2, expect_block_bytes=len(blk) + (1 << 20), expect_tuples=4 * n_tx + 256)
    try:
        for k in range(3 * len(devices) + 3):                   # the six signers are learned and earn their tables (on every device)
            r = fabgpu.preverify_block2(csp, blk, block_seq=k, lean=True)
        assert (np.asarray(r["tx_flags"]) == 0).all() and r["n_keyed"] == 4 * n_tx, (r["n_keyed"], int((np.asarray(r["tx_flags"]) != 0).sum()))
        before = csp.passes_per_device()
        copies = [[bytes(bytearray(blk)) for _ in range(per_caller)] for _ in range(callers)]
        per = [[] for _ in range(callers)]
        errs = []

        def caller(t):
            try:
                for k in range(per_caller):
                    seq = (t + 1) * 0x9E3779B97F4A7C15 + k * 0x632BE59BD9B4E019 & ((1 << 64) - 1)   # names like MemoSeq's: hashes
                    c0 = time.perf_counter()
                    r_ = fabgpu.preverify_block2(csp, copies[t][k], block_seq=seq, seed_memo=memo, lean=True)
                    per[t].append((time.perf_counter() - c0) * 1e3)
                    assert (np.asarray(r_["tx_flags"]) == 0).all()
                    if memo:
                        assert r_["memo_seeded"] == 4 * n_tx
                        fabgpu.memo_evict_block(csp, seq)
            except Exception as e:                              # noqa: BLE001
                errs.append(repr(e)[:200])
        th = [threading.Thread(target=caller, args=(t,)) for t in range(callers)]
        c0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - c0
        if errs:
            return {"error": errs[0]}
        served = [a - b for a, b in zip(csp.passes_per_device(), before)]
        lat = [x for p in per for x in p]
        return {"validated_tx_per_s": n_tx * callers * per_caller / wall, "ms_per_block_aggregate": wall / (callers * per_caller) * 1e3, "blocks": callers * per_caller,
                "callers": callers, "device_contexts": len(devices), "devices": devices, "passes_per_device": served,
                "pass_latency_ms": {"median": statistics.median(lat), "max": max(lat), "first": max(p[0] for p in per)},
                "routes": {k: v for k, v in fabgpu.pass_routes(csp).items() if k in ("device_walks", "host_walks")}}
    finally:
        csp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--devices", default="", help="comma-separated HIP ordinals (may repeat); default 0 .. gpus-1")
    ap.add_argument("--per-caller", type=int, default=8)
    ap.add_argument("--tx", type=int, default=10000)
    args = ap.parse_args()
    import numpy as np

    import fabgpu
    devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(args.gpus))
    path = os.path.join(ROOT, ".bench_blocks", "friendly_%d.bin" % args.tx)
    if os.path.exists(path):
        blk = open(path, "rb").read()
    else:
        import blockgen
        blk = blockgen.endorser_block(args.tx, 1)[0]
    G = len(devices)
    out = {"tool": "tools/bench_pool.py", "metric": "validated tx/s per block, ONE provider over %d device context(s), memo seeding + eviction, fresh copy of the block per pass" % G,
           "block_bytes": len(blk), "n_tx": args.tx}
    out.update(run(fabgpu, np, blk, devices, 2 * G, args.per_caller, args.tx))
    if G > 1:
        one = run(fabgpu, np, blk, devices[:1], 2, args.per_caller, args.tx)
        out["one_device_two_callers"] = one
        if "validated_tx_per_s" in one and "validated_tx_per_s" in out:
            out["speedup_over_one_device"] = out["validated_tx_per_s"] / one["validated_tx_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
