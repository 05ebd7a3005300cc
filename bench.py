#!/usr/bin/env python3
"""bench.py - BASELINE.json's metric on BASELINE.json's config, one process per GPU.

metric   : ECDSA P-256 verifies/sec (whole node); validated tx/sec per block is reported beside it.
workload : BASELINE.json configs[1] "Block of 10k tx x 3 endorsements, batched P-256 verify on 1 MI355X":
           n = 30 000 (Qx,Qy,e,r,s) tuples per GPU, synthetic (seed 20260921 + rank, fresh P-256 keypair per
           signature, low-S, 1 % invalid mix - SURVEY.md 8(d)), resident in HBM when the timed region starts.
step     : one pass of the hot path over one block: fabgpu_p256_verify_batch_dev (the C ABI the cgo provider
           binds) on torch's current stream; with N > 1 ranks each rank verifies its own block (weak scaling,
           signatures are independent) and one RCCL all-gather merges the per-rank verdict bitmaps over xGMI
           (SURVEY.md 8(e)); no other data-path collective exists.
Timing   : W warm-up steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize(); max over ranks.
Extras   : roofline (HIP events on the launch stream), valu_roofline (the integer-ALU fraction north_star asks for),
           cpu_baseline (rank 0, N = 1 only): OpenSSL libcrypto driven like bccsp/sw on all host cores - the
           reference's own Go path cannot be built here (no Go toolchain), see DESIGN.md.
The oracle (oracle/) is used only as the checker and as the cpu_baseline leg, never inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))

N_TX = 10000
N_ENDORSE = 3
SEED = 20260921
ALGO_BYTES_PER_VERIFY = 160.125          # 5 x 32 B in, 1 bit out
MAC_PER_VERIFY = 3.1e5                   # SURVEY.md 8(d) canonical u32 multiply-accumulate count per verify
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: 8 TB/s spec
# integer-MAC ceiling, MEASURED on MI355X by fabric-mod_amd/csrc/ubench.hip (profiles/r01_ubench_instruction_costs.txt):
# independent v_mad_u64_u32 streams at 4 waves/SIMD on all 1024 SIMDs retire one wave-instruction per 1.902 ns per SIMD
# (wall clock, i.e. at whatever frequency the chip sustains for a pure multiplier stream) = 64 lanes / 1.902 ns x 1024 SIMDs.
VALU_PEAK_MAC = 64 / 1.902e-9 * 1024
# v_mad_i64_i32 the verify kernels actually execute per signature (static count x trip counts):
# 263 dbl x 792 + 53 add x 1728 + 23 madd x 1179 + ~5 k = 3.32e5 with one lane per signature; the two-lane kernel executes the
# same products (2 lanes x (263 x 396 + 53 x 864 + 23 x 666)) plus the scalar part on both lanes = 3.4e5.
EXECUTED_MAC_PER_VERIFY = 3.4e5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tx", type=int, default=N_TX, help="tx per block (default = BASELINE configs[1]; other values are exploration only)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import fabgpu
    from fabgpu import sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    n_tx = args.tx
    n = n_tx * N_ENDORSE
    ctx = fabgpu.Context(device=local_rank, max_batch=n)
    block = fabgpu.synth_batch(n, seed=SEED + rank, invalid_permille=10)
    dev = {k: torch.from_numpy(block[k]).cuda() for k in ("qx", "qy", "e", "r", "s")}
    words_n = (n + 63) // 64
    words = torch.zeros(words_n, dtype=torch.int64, device="cuda")
    merged = torch.zeros(words_n * world, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        ctx.p256_verify_batch_dev(n, dev["qx"].data_ptr(), dev["qy"].data_ptr(), dev["e"].data_ptr(), dev["r"].data_ptr(),
                                  dev["s"].data_ptr(), words.data_ptr(), 0, stream.cuda_stream)
        if world > 1:
            dist.all_gather_into_tensor(merged, words)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stream_ms_per_step = ev0.elapsed_time(ev1) / args.steps      # HIP events on the launch stream over the timed region

    # per-launch kernel duration from the library's own HIP events (outside the timed region), for the roofline line
    kms = []
    for _ in range(min(10, args.steps)):
        ctx.p256_verify_batch_dev(n, dev["qx"].data_ptr(), dev["qy"].data_ptr(), dev["e"].data_ptr(), dev["r"].data_ptr(),
                                  dev["s"].data_ptr(), words.data_ptr(), 0, stream.cuda_stream)
        kms.append(ctx.last_kernel_ms())
    kernel_ms = float(np.mean(kms))

    # parity of the timed input: verdict bitmap vs the generator's ground truth (every rank) ...
    got = fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n)
    assert (got == (block["kind"] == 0)).all(), "verdict bitmap differs from the generator's ground truth"
    if world > 1:
        m = merged.cpu().numpy().view(np.uint64).reshape(world, words_n)
        assert (fabgpu.unpack_bits(m[rank], n) == got).all(), "all-gathered bitmap differs from the local one"

    if rank == 0:
        # HBM/fabric bytes per launch from the rocprofv3 PMC passes of this same command (profiles/r01_pmc_traffic.json:
        # 2 x FETCH_SIZE per MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE), only valid for the BASELINE workload
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if n_tx == N_TX and os.path.exists(tpath):
            traffic = json.load(open(tpath)).get("traffic_gb_per_launch") * 1e9   # bytes per launch
        ms_per_step = dt / args.steps * 1e3
        total = n * world
        value = total / (dt / args.steps)
        kernel_s = (stream_ms_per_step if world == 1 else kernel_ms) * 1e-3
        achieved = ALGO_BYTES_PER_VERIFY * n / kernel_s / 1e9
        out = {
            "metric": "ECDSA P-256 verifies/sec (whole node)", "value": value, "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64",
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]" if n_tx == N_TX else "EXPLORATION (not the BASELINE config)") + ": block of %d tx x 3 endorsements = %d P-256 tuples per GPU, " % (n_tx, n) +
                                   "verify-only kernel via the C ABI, fresh keypair per signature, 1% invalid",
                       "tuples_per_gpu": n, "tx_per_block": n_tx, "endorsements_per_tx": N_ENDORSE, "seed": SEED,
                       "parallelism": "1 block per GPU%s" % (" + RCCL all-gather of verdict bitmaps" if world > 1 else "")},
            "validated_tx_per_s": n_tx * world / (dt / args.steps),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "bytes/launch (algorithmic: %d)" % int(ALGO_BYTES_PER_VERIFY * n),
                         "kernel": "p256_verify_pair_kernel<256> (two lanes per signature)" if n <= 32768 else "p256_verify_kernel<256>", "kernel_ms": kernel_s * 1e3,
                         "kernel_ms_lib_events": kernel_ms,
                         "note": "integer-VALU-bound, not HBM-bound (SURVEY 8(d)): see valu_roofline"},
            "valu_roofline": {"bound": "u32-mac", "achieved": n / kernel_s * MAC_PER_VERIFY, "peak": VALU_PEAK_MAC, "unit": "MAC/s",
                              "frac": n / kernel_s * MAC_PER_VERIFY / VALU_PEAK_MAC,
                              "model": "3.1e5 u32 MACs per verify (SURVEY 8(d) canonical count); peak = measured v_mad_u64_u32 ceiling of csrc/ubench.hip",
                              "executed_mac_per_verify": EXECUTED_MAC_PER_VERIFY,
                              "executed_frac": n / kernel_s * EXECUTED_MAC_PER_VERIFY / VALU_PEAK_MAC},
            "parity": "verdict bitmap bit-identical to generator ground truth on the timed input",
        }
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import coracle
            want = coracle.verify_batch(block["qx"], block["qy"], block["e"], block["r"], block["s"])      # the oracle checks ...
            assert (got == (want == 0)).all(), "GPU verdicts differ from the oracle"
            best = None
            for _ in range(3):                                                                                # ... and OpenSSL is timed
                c0 = time.perf_counter()
                st = coracle.ossl_verify_batch(block["qx"], block["qy"], block["e"], block["r"], block["s"])
                c1 = time.perf_counter()
                best = c1 - c0 if best is None else min(best, c1 - c0)
            assert (st == want).all()
            cores = len(os.sched_getaffinity(0))
            out["cpu_baseline"] = {"value": n / best, "unit": "verifies/s", "cores": cores, "kind": "port",
                                   "sample": "the same 30000-tuple block, all host cores (OpenMP), best of 3; OpenSSL 3 nistz256 "
                                             "ECDSA_do_verify + low-S gate = proxy for bccsp/sw (Go toolchain absent)"}
            out["parity"] = "verdict bitmap bit-identical to the CPU oracle and to OpenSSL on the timed input"
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
