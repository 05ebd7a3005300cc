#!/usr/bin/env python3
"""The three kernels bench.py's rooflines are about, a few launches each, for a rocprofv3 --pmc pass (tools/gpu_pmc_sq.sh):
  p256_verify_pair_lds_kernel<256>      30 000 tuples (BASELINE configs[1]; the default above 16 384 tuples)
  p256_verify_pair_kernel<256>          30 000 tuples, table in the global workspace (--pair-table global: the form round 2 profiled)
  sha256_p256_verify_kernel<256>        300 000 messages of 1 856 B (BASELINE configs[3]; one lane per signature)
  idemix_nym_verify_quad_kernel<256>    6 000 pseudonym signatures over 4 608-byte messages (the idemix share of BASELINE configs[4])
Synthetic inputs, verdicts checked against the generator's ground truth.  No timing here: counters perturb it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np   # noqa: E402
import torch   # noqa: E402

import fabgpu   # noqa: E402

LAUNCHES = int(os.environ.get("PMC_LAUNCHES", "6"))
SEED = 20260921


def main():
    which = set(sys.argv[1:]) or {"verify", "verify_global", "fused", "nym"}
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    if "verify" in which or "verify_global" in which:
        n = 30000
        b = fabgpu.synth_batch(n, seed=SEED, invalid_permille=10)
        d = {k: torch.from_numpy(b[k]).cuda() for k in ("qx", "qy", "e", "r", "s")}
        words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        for name, flags in (("verify", 8), ("verify_global", 16)):
            if name not in which:
                continue
            ctx = fabgpu.Context(device=0, max_batch=n, flags=flags)
            for _ in range(LAUNCHES):
                ctx.p256_verify_batch_dev(n, d["qx"].data_ptr(), d["qy"].data_ptr(), d["e"].data_ptr(), d["r"].data_ptr(), d["s"].data_ptr(), words.data_ptr(), 0, st)
            torch.cuda.synchronize()
            assert (fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n) == (b["kind"] == 0)).all()
            ctx.close()
    if "fused" in which:
        n, L = 300000, 1856
        rng = np.random.default_rng(SEED)
        arena = rng.integers(0, 256, size=n * L, dtype=np.uint8)
        off = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
        ctx = fabgpu.Context(device=0)
        t_arena, t_off = torch.from_numpy(arena).cuda(), torch.from_numpy(off.view(np.int32)).cuda()
        dig_d = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        ctx.sha256_batch_dev(n, t_arena.data_ptr(), t_arena.numel(), t_off.data_ptr(), dig_d.data_ptr(), st)
        torch.cuda.synchronize()
        b = fabgpu.synth_batch(n, seed=SEED, invalid_permille=10, e_in=dig_d.cpu().numpy().reshape(n, 32))
        t = {k: torch.from_numpy(b[k]).cuda() for k in ("qx", "qy", "r", "s")}
        words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        for _ in range(max(2, LAUNCHES // 2)):
            ctx.sha256_p256_verify_batch_dev(n, t_arena.data_ptr(), t_arena.numel(), t_off.data_ptr(), t["qx"].data_ptr(), t["qy"].data_ptr(), t["r"].data_ptr(),
                                             t["s"].data_ptr(), words.data_ptr(), 0, st)
        torch.cuda.synchronize()
        assert (fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n) == ((b["kind"] == 0) | (b["kind"] == 1))).all()
        ctx.close()
    if "nym" in which:
        import random

        import idemix_oracle as io
        from idemix_common import NymBatch, be32, fixtures
        fx = fixtures()
        ctx = fabgpu.Context(device=0)
        issuers = []
        for name in ("MSP1OU1", "MSP2OU1"):
            ipk = fx[name]["ipk"]
            ctx.idemix_issuer_register((be32(ipk.h_sk[0]), be32(ipk.h_sk[1])), (be32(ipk.h_rand[0]), be32(ipk.h_rand[1])), ipk.hash)
            issuers.append((ipk, fx[name]["signer"].sk))
        rng = random.Random(SEED)
        nb = NymBatch()
        base, n_nym = 64, 6000
        for i in range(base):
            ipk, sk = issuers[i % 2]
            nym, r_nym = io.make_nym(sk, ipk, rng)
            msg = bytes(rng.getrandbits(8) for _ in range(4608))
            nb.add(i % 2, ipk, nym, io.nym_sign(sk, nym, r_nym, ipk, msg, rng), msg)
        arena, off, iid, cols, expect = nb.arrays()
        pick = np.random.default_rng(1).integers(0, base, size=n_nym)
        lens = (off[1:] - off[:-1])[pick]
        off2 = np.zeros(n_nym + 1, dtype=np.uint32)
        off2[1:] = np.cumsum(lens)
        arena2 = np.concatenate([arena[off[i]:off[i + 1]] for i in pick])
        d_arena, d_off = torch.from_numpy(arena2).cuda(), torch.from_numpy(off2.view(np.int32)).cuda()
        d_iid = torch.from_numpy(np.ascontiguousarray(iid[pick]).view(np.int32)).cuda()
        d_cols = [torch.from_numpy(np.ascontiguousarray(c[pick])).cuda() for c in cols]
        words = torch.zeros((n_nym + 63) // 64, dtype=torch.int64, device="cuda")
        for _ in range(LAUNCHES):
            ctx.idemix_nym_verify_batch_dev(n_nym, d_arena.data_ptr(), d_arena.numel(), d_off.data_ptr(), d_iid.data_ptr(), *[c.data_ptr() for c in d_cols],
                                            words.data_ptr(), 0, st)
        torch.cuda.synchronize()
        assert (fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n_nym) == (expect[pick] == 0)).all()
        ctx.close()
    print("pmc kernels ok:", sorted(which))


if __name__ == "__main__":
    main()
