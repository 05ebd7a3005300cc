#!/usr/bin/env python3
"""Exploration bench for BASELINE config 5 on ONE GPU: a mixed batch, 80 % ECDSA P-256 tuples (fresh keys, as bench.py) and
20 % idemix pseudonym signatures (FP256BN; creators only - docs/source/idemix.rst:171-176), device-resident inputs, the two
kernels launched on two HIP streams per step.  Also times the idemix kernel alone.  Not the driver's bench (bench.py measures
configs[1]).  The oracles are used only as checkers: every timed input's verdict vector is compared with theirs.
(The 8-GPU form of config 5 shards both sub-batches by contiguous ranges exactly as bench.py --gpus N does for P-256.)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fabric-mod_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=30000, help="total signatures per step")
    ap.add_argument("--idemix-share", type=float, default=0.2)
    ap.add_argument("--msg-len", type=int, default=4608, help="creator message bytes (SURVEY 8(d): 4 608)")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=12, help="untimed launches per timed leg (the chip needs ~10 ms of work to leave its idle clocks)")
    ap.add_argument("--no-quad", action="store_true", help="FABGPU_FLAG_NO_QUAD: the idemix share on the two-lanes-per-signature kernel (A/B)")
    ap.add_argument("--fused-hash", action="store_true", help="FABGPU_FLAG_NYM_FUSED_HASH: round 4's single four-lane kernel instead of the two phases (A/B)")
    ap.add_argument("--no-side-stream", action="store_true", help="FABGPU_FLAG_NYM_NO_SIDE_STREAM: the fixed-base terms inside the commitment kernel (A/B)")
    ap.add_argument("--prio", default="", help="stream priorities 'ec,nym' (e.g. -1,0: the ECDSA stream high; HIP: lower number = higher priority) - exploration")
    ap.add_argument("--marker", choices=("", "timing", "plain"), default="", help="the caller records an event on the ECDSA stream after every ECDSA launch (exploration)")
    ap.add_argument("--no-time-kernels", action="store_true", help="do not bracket launches with timing events (as bench.py's context)")
    ap.add_argument("--base", type=int, default=192, help="distinct oracle-signed pseudonym signatures that the batch replicates")
    args = ap.parse_args()
    import random

    import numpy as np
    import torch

    import fabgpu
    import idemix_oracle as io
    from idemix_common import NymBatch, be32, fixtures

    fx = fixtures()
    ctx = fabgpu.Context(device=0, max_batch=args.n, flags=(0 if args.no_time_kernels else fabgpu.FLAG_TIME_KERNELS) | (fabgpu.FLAG_NO_QUAD if args.no_quad else 0) |
                         (fabgpu.FLAG_NYM_FUSED_HASH if args.fused_hash else 0) | (fabgpu.FLAG_NYM_NO_SIDE_STREAM if args.no_side_stream else 0))
    issuers = []
    t0 = time.perf_counter()
    for name in ("MSP1OU1", "MSP2OU1"):
        ipk = fx[name]["ipk"]
        ctx.idemix_issuer_register((be32(ipk.h_sk[0]), be32(ipk.h_sk[1])), (be32(ipk.h_rand[0]), be32(ipk.h_rand[1])), ipk.hash)
        issuers.append((ipk, fx[name]["signer"].sk))
    reg_ms = (time.perf_counter() - t0) * 1e3 / 2

    n_nym = int(round(args.n * args.idemix_share))
    n_ec = args.n - n_nym
    # idemix part: `base` signatures signed by the oracle (1 % tampered), replicated to n_nym
    rng = random.Random(20260921)
    nb = NymBatch()
    for i in range(args.base):
        k = i % 2
        ipk, sk = issuers[k]
        nym, r_nym = io.make_nym(sk, ipk, rng)
        msg = bytes(rng.getrandbits(8) for _ in range(args.msg_len))
        sig = io.nym_sign(sk, nym, r_nym, ipk, msg, rng)
        if i % 100 == 99:
            msg = msg[:-1] + bytes([msg[-1] ^ 1])
        nb.add(k, ipk, nym, sig, msg)
    arena, off, iid, cols, expect = nb.arrays()
    pick = np.random.default_rng(1).integers(0, args.base, size=n_nym)
    lens = (off[1:] - off[:-1])[pick]
    off2 = np.zeros(n_nym + 1, dtype=np.uint32)
    off2[1:] = np.cumsum(lens)
    arena2 = np.concatenate([arena[off[i]:off[i + 1]] for i in pick])
    d_arena = torch.from_numpy(arena2).cuda()
    d_off = torch.from_numpy(off2.view(np.int32)).cuda()
    d_iid = torch.from_numpy(iid[pick].view(np.int32)).cuda()
    d_cols = [torch.from_numpy(c[pick]).cuda() for c in cols]
    d_words_nym = torch.zeros((n_nym + 63) // 64, dtype=torch.int64, device="cuda")
    want_nym = expect[pick] == 0

    # ECDSA part
    import coracle
    b = fabgpu.synth_batch(n_ec, seed=20260921, invalid_permille=10)
    d_ec = {k: torch.from_numpy(b[k]).cuda() for k in ("qx", "qy", "e", "r", "s")}
    d_words_ec = torch.zeros((n_ec + 63) // 64, dtype=torch.int64, device="cuda")
    want_ec = coracle.verify_batch(b["qx"], b["qy"], b["e"], b["r"], b["s"]) == 0

    if args.prio:
        p1, p2 = [int(x) for x in args.prio.split(",")]
        s1, s2 = torch.cuda.Stream(priority=p1), torch.cuda.Stream(priority=p2)
    else:
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def step_nym(st):
        ctx.idemix_nym_verify_batch_dev(n_nym, d_arena.data_ptr(), d_arena.numel(), d_off.data_ptr(), d_iid.data_ptr(), *[c.data_ptr() for c in d_cols],
                                        d_words_nym.data_ptr(), 0, st.cuda_stream)

    markers = [torch.cuda.Event(enable_timing=(args.marker == "timing")) for _ in range(64)] if args.marker else []
    mk = [0]

    def step_ec(st):
        ctx.p256_verify_batch_dev(n_ec, d_ec["qx"].data_ptr(), d_ec["qy"].data_ptr(), d_ec["e"].data_ptr(), d_ec["r"].data_ptr(), d_ec["s"].data_ptr(),
                                  d_words_ec.data_ptr(), 0, st.cuda_stream)
        if markers:
            markers[mk[0] % 64].record(st)
            mk[0] += 1

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    dt_ec = timed(lambda: step_ec(s1))
    dt_nym = timed(lambda: step_nym(s1))
    ker_ms = ctx.last_kernel_ms()
    dt_mix = timed(lambda: (step_ec(s1), step_nym(s2)))
    dt_ec2 = timed(lambda: step_ec(s1))      # again after the dense idemix runs: the chip is power-limited
    got_nym = fabgpu.unpack_bits(d_words_nym.cpu().numpy().view(np.uint64), n_nym)
    got_ec = fabgpu.unpack_bits(d_words_ec.cpu().numpy().view(np.uint64), n_ec)
    assert (got_nym == want_nym).all(), "idemix verdicts differ from the oracle"
    assert (got_ec == want_ec).all(), "ECDSA verdicts differ from the oracle"
    print(json.dumps({
        "metric": "signature verifies/sec, mixed batch (config 5 on one GPU)", "value": args.n / dt_mix, "unit": "verifies/s",
        "ms_per_step": dt_mix * 1e3,
        "config": {"workload": "%d ECDSA P-256 tuples (fresh keys) + %d idemix pseudonym signatures (%d-byte messages, 2 issuers), 1 %% invalid, two streams"
                   % (n_ec, n_nym, args.msg_len)},
        "idemix_alone": {"n": n_nym, "verifies_per_s": n_nym / dt_nym, "ms_per_step": dt_nym * 1e3, "kernel_ms": ker_ms},
        "ecdsa_alone": {"n": n_ec, "verifies_per_s": n_ec / dt_ec, "ms_per_step": dt_ec * 1e3, "ms_per_step_after_idemix_runs": dt_ec2 * 1e3},
        "issuer_register_ms": reg_ms, "parity": "both verdict bitmaps bit-identical to the CPU oracles"}))
    ctx.close()


if __name__ == "__main__":
    main()
