#!/usr/bin/env python3
"""A/B of the two homes of the pair kernel's per-signature table: global workspace (16 entries, 52 signed 5-bit
windows) against LDS (8 entries, 65 signed 4-bit windows) on BASELINE configs[1] (30 000 tuples, inputs resident in HBM), launch time
by HIP events, verdicts checked against the generator's ground truth.  One JSON line."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import fabgpu  # noqa: E402

torch.cuda.set_device(0)
out = {}
for n in (30000, 10000, 1000):
    b = fabgpu.synth_batch(n, seed=20260921, invalid_permille=10)
    dev = {k: torch.from_numpy(b[k]).cuda() for k in ("qx", "qy", "e", "r", "s")}
    words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()
    for mode, flag in (("global", fabgpu.FLAG_PAIR_TABLE_GLOBAL), ("lds", fabgpu.FLAG_PAIR_TABLE_LDS), ("global_again", fabgpu.FLAG_PAIR_TABLE_GLOBAL)):
        ctx = fabgpu.Context(device=0, max_batch=n, flags=flag)

        def verify():
            ctx.p256_verify_batch_dev(n, dev["qx"].data_ptr(), dev["qy"].data_ptr(), dev["e"].data_ptr(), dev["r"].data_ptr(), dev["s"].data_ptr(),
                                      words.data_ptr(), 0, stream.cuda_stream)
        for _ in range(10):
            verify()
        torch.cuda.synchronize()
        each = []
        for _ in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            verify()
            e1.record(stream)
            e1.synchronize()
            each.append(e0.elapsed_time(e1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(40):
            verify()
        e1.record(stream)
        e1.synchronize()
        got = fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n)
        assert (got == (b["kind"] == 0)).all(), mode
        out["%d_%s" % (n, mode)] = {"median_ms": statistics.median(each), "min_ms": min(each), "back_to_back_ms": e0.elapsed_time(e1) / 40}
        ctx.close()
print(json.dumps(out))
