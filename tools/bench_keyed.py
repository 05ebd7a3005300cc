#!/usr/bin/env python3
"""Exploration bench for the REGISTERED-KEY path (SURVEY.md 8(d) "realistic" variant: endorsers drawn from a pool of 16
keys, msp/cache/cache.go:14-18): n tuples signed by a pool of keys whose comb tables are resident on the device, one
fabgpu_p256_verify_batch_keyed_dev launch per step.  Not the driver's bench (bench.py measures configs[1], a fresh keypair
per signature).  The oracle is used only as the checker."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30000)
    ap.add_argument("--keys", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tables16", action="store_true", help="FABGPU_FLAG_KEY_TABLES_16BIT: 16-bit combs for the registered keys (round 6)")
    args = ap.parse_args()
    import numpy as np
    import torch

    import coracle
    import fabgpu
    n = args.n
    b = coracle.make_pool_batch(n, seed=20260921, nkeys=args.keys, invalid_frac=0.01)
    ctx = fabgpu.Context(device=0, max_batch=n, flags=fabgpu.FLAG_KEY_TABLES_16BIT if args.tables16 else 0)
    t0 = time.perf_counter()
    ids = np.array([ctx.key_register(b["pool_qx"][j].tobytes(), b["pool_qy"][j].tobytes()) for j in range(args.keys)], dtype=np.uint32)
    reg_ms = (time.perf_counter() - t0) * 1e3 / args.keys
    tables16_ms = None
    if args.tables16:
        ready = ctx.test_key_tables16(int(ids[0]))                  # waits for the builds queued behind the registrations
        tables16_ms = (time.perf_counter() - t0) * 1e3
        assert ready == min(args.keys, 64), ready
    kid = torch.from_numpy(ids[b["key_index"]].view(np.int32)).cuda()
    dev = {k: torch.from_numpy(b[k]).cuda() for k in ("e", "r", "s")}
    words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        ctx.p256_verify_batch_keyed_dev(n, kid.data_ptr(), dev["e"].data_ptr(), dev["r"].data_ptr(), dev["s"].data_ptr(), words.data_ptr(), 0,
                                        stream.cuda_stream)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    got = fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n)
    want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
    assert (got == (want == 0)).all(), "keyed verdicts differ from the oracle"
    print(json.dumps({"metric": "ECDSA P-256 verifies/sec, registered keys", "value": n / dt, "unit": "verifies/s", "ms_per_step": dt * 1e3,
                      "config": {"workload": "%d tuples signed by a pool of %d registered keys, 1%% invalid, 1 GPU" % (n, args.keys)},
                      "key_register_ms_per_key": reg_ms, "tables_16bit": bool(args.tables16), "all_16bit_tables_built_and_one_checked_after_ms": tables16_ms, "parity": "verdict bitmap bit-identical to the CPU oracle"}))
    ctx.close()


if __name__ == "__main__":
    main()
