#!/usr/bin/env python3
"""Randomised parity soak: for --seconds, draw batch sizes across both kernel geometries (two lanes per signature up to 32 768, one lane
beyond), mutation mixes and entry points (verify-only, fused hash + verify with ragged messages, registered keys) and compare every
status byte and verdict bit with the C oracle.  Test infrastructure: tests/test_soak.py runs it for FABGPU_SOAK_SECONDS (default 15) under
`pytest -m gpu`; `python tests/soak_parity.py --seconds 240` is the long run recorded in profiles/r02_soak_parity.json."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
for p in ("fabric-mod_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))


def soak(seconds, seed, tables16=False, keyed_only=False):
    import types
    args = types.SimpleNamespace(seconds=seconds, seed=seed)
    import numpy as np

    import coracle
    import fabgpu
    rng = np.random.default_rng(args.seed)
    # tables16: FABGPU_FLAG_KEY_TABLES_16BIT - the keyed batches then verify while their keys' 16-bit tables are still being built behind
    # the registrations (a wavefront sees each table or not yet: both must give the oracle's statuses), and beyond the 64th key without one
    ctx = fabgpu.Context(device=0, flags=fabgpu.FLAG_KEY_TABLES_16BIT if tables16 else 0)
    t_end = time.time() + args.seconds
    stats = {"batches": 0, "tuples": 0, "invalid": 0, "verify_only": 0, "fused": 0, "keyed": 0, "pair_geometry": 0, "one_lane_geometry": 0}
    it = 0
    while time.time() < t_end:
        it += 1
        n = int(rng.choice([rng.integers(1, 300), rng.integers(300, 33000), rng.integers(33000, 90000)], p=[0.2, 0.5, 0.3]))
        mode = 2 if keyed_only else it % 3
        b = fabgpu.synth_batch(n, seed=int(rng.integers(1, 1 << 40)), invalid_permille=int(rng.choice([0, 10, 200, 500])))
        if mode == 0:
            want = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
            bits, st = ctx.p256_verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
            stats["verify_only"] += 1
        elif mode == 1:
            lens = rng.integers(0, 400, size=n)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
            arena = rng.integers(0, 256, size=int(off[-1]) + 64, dtype=np.uint8)
            dig = ctx.sha256_batch(arena, off)
            b = fabgpu.synth_batch(n, seed=int(rng.integers(1, 1 << 40)), invalid_permille=100, e_in=dig)
            want = coracle.sha256_verify_batch(arena, off, b["qx"], b["qy"], b["r"], b["s"])
            bits, st = ctx.sha256_p256_verify_batch(arena, off, b["qx"], b["qy"], b["r"], b["s"])
            stats["fused"] += 1
        else:
            nk = int(rng.integers(1, 24))
            pb = coracle.make_pool_batch(min(n, 40000), seed=int(rng.integers(1, 1 << 30)), nkeys=nk, invalid_frac=0.2)
            n = pb["qx"].shape[0]
            ids = np.array([ctx.key_register(pb["pool_qx"][j].tobytes(), pb["pool_qy"][j].tobytes()) for j in range(nk)], dtype=np.uint32)
            want = coracle.verify_batch(pb["qx"], pb["qy"], pb["e"], pb["r"], pb["s"])
            bits, st = ctx.p256_verify_batch_keyed(ids[pb["key_index"]], pb["e"], pb["r"], pb["s"])
            stats["keyed"] += 1
        if not ((st == want).all() and (bits == (want == 0)).all()):
            bad = np.nonzero(st != want)[0][:5]
            return {"soak": "MISMATCH", "iteration": it, "n": n, "mode": mode, "first_bad": bad.tolist()}
        stats["batches"] += 1
        stats["tuples"] += n
        stats["invalid"] += int((want != 0).sum())
        stats["pair_geometry" if n <= 32768 else "one_lane_geometry"] += 1
    ctx.close()
    return {"soak": "ok", "seconds": args.seconds, "seed": args.seed, **stats,
            "parity": "every status byte and verdict bit equal to the C oracle (oracle/p256_oracle.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--tables16", action="store_true", help="a context with FABGPU_FLAG_KEY_TABLES_16BIT (round 6)")
    ap.add_argument("--keyed-only", action="store_true", help="registered-key batches only")
    a = ap.parse_args()
    r = soak(a.seconds, a.seed, a.tables16, a.keyed_only)
    r["tables16"], r["keyed_only"] = a.tables16, a.keyed_only
    print(json.dumps(r))
    sys.exit(0 if r["soak"] == "ok" else 1)


if __name__ == "__main__":
    main()
