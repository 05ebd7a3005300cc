#!/usr/bin/env python3
"""Default-sized blocks through the pass (device route): median wall and device phase for 100 / 500 / 1 000 transactions, with the
eight-lane two-phase keyed kernels (default) and without (FABGPU_FLAG_NO_WIDE).   usage: gpu_probe_small2.py [sizes ...] [--one N]
--one N: a single configuration (wide, N transactions), eight passes - for a rocprofv3 --kernel-trace timeline."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np   # noqa: E402

import blockgen   # noqa: E402
import fabgpu   # noqa: E402

if os.environ.get("PROBE_LIB"):          # A/B against another build of the library (same call, same box)
    fabgpu._LIB_PATH = os.environ["PROBE_LIB"]


def run(ntx, flags, passes, memo=False):
    blk, _ = blockgen.endorser_block(ntx, 31 + ntx)
    csp = fabgpu.GPUCSP(devices=[0], flags=flags, concurrent_passes=2, expect_block_bytes=len(blk) + (1 << 20), expect_tuples=4 * ntx + 256)
    try:
        for k in range(16):                      # (an identity earns its comb table after 64 namings: a few blocks of 100)
            r = fabgpu.preverify_block2(csp, blk, block_seq=k, lean=True)
            if k >= 3 and r["n_keyed"] == 4 * ntx:
                break
        assert (np.asarray(r["tx_flags"]) == 0).all() and r["n_keyed"] == 4 * ntx, (r["n_keyed"], 4 * ntx)
        per, dev = [], []
        for k in range(passes):
            b = bytes(bytearray(blk))
            c0 = time.perf_counter()
            r = fabgpu.preverify_block2(csp, b, block_seq=100 + k, seed_memo=memo, lean=True)
            per.append((time.perf_counter() - c0) * 1e3)
            dev.append(r["ms_stage"][2])
            assert (np.asarray(r["tx_flags"]) == 0).all()
            if memo:
                fabgpu.memo_evict_block(csp, 100 + k)
        return statistics.median(per), statistics.median(dev), min(per)
    finally:
        csp.close()


def main():
    if "--one" in sys.argv:
        n = int(sys.argv[sys.argv.index("--one") + 1])
        print("one", n, run(n, 0, 8))
        return
    sizes = [int(x) for x in sys.argv[1:]] or [100, 500, 1000]
    for n in sizes:
        for name, flags in (("wide", 0), ("no-wide", fabgpu.FLAG_NO_WIDE)):
            for memo in (False, True):
                med, dev, mn = run(n, flags, 30, memo)
                print("%5d tx  %-8s %-5s  pass median %.3f ms (min %.3f)  device phase %.3f ms" % (n, name, "memo" if memo else "flags", med, mn, dev), flush=True)


if __name__ == "__main__":
    main()
