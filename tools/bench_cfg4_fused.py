#!/usr/bin/env python3
"""Exploration bench for BASELINE.json configs[3]: "SHA-256 of proposal-response bytes fused ahead of verify (hash+verify
kernel), 100k tx, 1 GPU".  n = tx x 3 messages of 1856 bytes (prp 1024 B per tx + endorser 832 B per endorsement,
SURVEY.md 8(d)), one fused launch per step through fabgpu_sha256_p256_verify_batch_dev.  Not the driver's bench (bench.py
measures configs[1]); prints one JSON line with the same timing protocol.  The oracle is used only as the checker."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tx", type=int, default=100000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prefixed", action="store_true", help="hand the prp of each tx over once as a shared prefix (mid-state reuse)")
    ap.add_argument("--keys", type=int, default=0, help="sign with a pool of this many REGISTERED keys instead of a fresh key per signature")
    args = ap.parse_args()
    import numpy as np
    import torch

    import fabgpu
    n = args.tx * 3
    rng = np.random.default_rng(20260921)
    prp = rng.integers(0, 256, size=(args.tx, 1024), dtype=np.uint8)
    endorser = rng.integers(0, 256, size=(n, 832), dtype=np.uint8)
    arena = np.empty((n, 1856), dtype=np.uint8)
    arena[:, :1024] = np.repeat(prp, 3, axis=0)
    arena[:, 1024:] = endorser
    off = (np.arange(n + 1, dtype=np.uint64) * 1856).astype(np.uint32)
    dig = np.frombuffer(b"".join(hashlib.sha256(arena[i].tobytes()).digest() for i in range(n)), dtype=np.uint8).reshape(n, 32)
    ctx = fabgpu.Context(device=0, max_batch=n)
    if args.keys:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import coracle
        b = coracle.make_pool_batch(n, seed=20260921, nkeys=args.keys, invalid_frac=0.01, digests=dig)
        ids = np.array([ctx.key_register(b["pool_qx"][j].tobytes(), b["pool_qy"][j].tobytes()) for j in range(args.keys)], dtype=np.uint32)
        kid = torch.from_numpy(ids[b["key_index"]].view(np.int32)).cuda()
    else:
        b = fabgpu.synth_batch(n, seed=20260921, invalid_permille=10, e_in=dig)
    if args.prefixed:   # arena = all prp back to back, then all endorser suffixes
        flat = np.concatenate([prp.reshape(-1), endorser.reshape(-1)])
        pre_off = (np.arange(args.tx + 1, dtype=np.uint64) * 1024).astype(np.uint32)
        pre_idx = np.repeat(np.arange(args.tx, dtype=np.uint32), 3)
        off = (args.tx * 1024 + np.arange(n + 1, dtype=np.uint64) * 832).astype(np.uint32)
    else:
        flat = arena.reshape(-1)
    t = {k: torch.from_numpy(v).cuda() for k, v in dict(arena=flat, off=off.view(np.int32), qx=b["qx"], qy=b["qy"], r=b["r"], s=b["s"]).items()}
    words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()
    desc = fabgpu._IdBatch()
    desc.n, desc.arena, desc.arena_bytes, desc.off = n, t["arena"].data_ptr(), t["arena"].numel(), t["off"].data_ptr()
    desc.r, desc.s, desc.verdict_bits = t["r"].data_ptr(), t["s"].data_ptr(), words.data_ptr()
    if args.keys:
        desc.key_id = kid.data_ptr()
    else:
        desc.qx, desc.qy = t["qx"].data_ptr(), t["qy"].data_ptr()
    mid = None
    if args.prefixed:
        tp = {"pre_off": torch.from_numpy(pre_off.view(np.int32)).cuda(), "pre_idx": torch.from_numpy(pre_idx.view(np.int32)).cuda()}
        mid = torch.zeros(args.tx * 8, dtype=torch.int32, device="cuda")
        desc.n_prefixes, desc.pre_off, desc.pre_idx = args.tx, tp["pre_off"].data_ptr(), tp["pre_idx"].data_ptr()

    def step():
        ctx.identity_verify_batch_dev(desc, mid.data_ptr() if mid is not None else 0, stream.cuda_stream)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    got = fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n)
    # kind 1 mutates e_out only: in hash mode the message decides, so those tuples stay valid
    want = (b["kind"] == 0) | (b["kind"] == 1)
    if args.keys:
        import coracle
        w = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"])
        want = (w == 0) | (b["kind"] == 1)
    assert (got == want).all(), "fused verdicts differ from the generator's ground truth"
    print(json.dumps({"metric": "fused SHA-256 + ECDSA P-256 verifies/sec", "value": n / dt, "unit": "verifies/s", "ms_per_step": dt * 1e3,
                      "config": {"workload": "BASELINE.json configs[3]: %d tx x 3 messages of 1856 B, fused hash+verify, 1 GPU%s%s" % (
                          args.tx, ", prp handed over once per tx (mid-state reuse)" if args.prefixed else "",
                          ", pool of %d registered keys" % args.keys if args.keys else ""), "tuples": n,
                                 "message_bytes": 1856}, "hashed_GB_per_s": n * 1856 / dt / 1e9, "validated_tx_per_s": args.tx / dt,
                      "parity": "verdict bitmap equals the generator's ground truth"}))
    ctx.close()


if __name__ == "__main__":
    main()
