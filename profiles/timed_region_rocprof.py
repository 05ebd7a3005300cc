#!/usr/bin/env python3
"""From a rocprofv3 (rocpd sqlite) kernel trace of `python bench.py --gpus 1 --steps K --warmup W`: the average duration of the dominant
kernel's launches INSIDE the contract's timed region.  bench.py launches p256_verify_pair_lds_kernel<256> in this order: --clock-warmup
(60) launches, W warm-up steps, the K timed steps, then the dispersion leg and the other legs (whose launches of the same kernel - the
mixed leg's run beside the idemix kernels - are in the --stats average but have nothing to do with `roofline.kernel_ms`).
usage: timed_region_rocprof.py <results.db> [clock_warmup=60 W=5 K=20]     clock_warmup may be the path of the bench line of the same run:
its config.clock_warmup_launches is then used (round 6: the warm-up lasts until the launches stop getting faster, so the number varies)."""
import sqlite3
import sys

db = sys.argv[1]
args = sys.argv[2:5] + ["60", "5", "20"][len(sys.argv) - 2:]
if not args[0].isdigit():
    import json
    args[0] = str(json.loads(open(args[0]).read().strip().splitlines()[-1])["config"]["clock_warmup_launches"])
cw, w, k = (int(x) for x in args)
c = sqlite3.connect(db)
rows = sorted((s, e) for name, s, e in c.execute("select name, start, end from kernels") if "p256_verify_pair_lds_kernel" in name)
d = [(e - s) / 1e3 for s, e in rows]
if len(d) < cw + w + k:
    sys.exit("only %d launches of the kernel in the trace" % len(d))
mean = lambda x: sum(x) / len(x)
print("p256_verify_pair_lds_kernel<256>: %d launches in the trace" % len(d))
print("  clock warm-up launches 1-25 : %.1f us average" % mean(d[:25]))
print("  clock warm-up launches 26-%d: %.1f us average" % (cw, mean(d[25:cw])))
print("  the W = %d warm-up steps     : %.1f us average" % (w, mean(d[cw:cw + w])))
print("  the K = %d TIMED steps      : %.1f us average  (min %.1f, max %.1f)   <- compare with roofline.kernel_ms" % (k, mean(d[cw + w:cw + w + k]), min(d[cw + w:cw + w + k]), max(d[cw + w:cw + w + k])))
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(cw + w, cw + w + k - 1)]
print("  gaps between the timed launches: %.1f us average" % mean(gaps))
print("  all launches (what --stats averages): %.1f us" % mean(d))
