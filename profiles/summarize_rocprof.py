#!/usr/bin/env python3
"""Turn rocprofv3 (ROCm 7.2, rocpd sqlite output) results into the text summaries committed under profiles/.
usage: summarize_rocprof.py <results.db> [...]   -> prints kernel stats (top_kernels view) and per-kernel PMC averages."""
import sqlite3
import sys
from collections import defaultdict

for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    print("== %s" % path)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    if rows:
        print("%-60s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for n, calls, tot, avg, pct in rows:
            print("%-60s %8d %14.1f %12.3f %8.3f" % (n[:60], calls, tot, avg, pct))
    try:
        pm = list(c.execute("select kernel_name,counter_name,value,vgpr_count,accum_vgpr_count,sgpr_count,lds_block_size,scratch_size,grid_size,workgroup_size from counters_collection"))
    except sqlite3.Error:
        pm = []
    agg = defaultdict(list)
    meta = {}
    for k, cn, v, vg, ag, sg, lds, scr, grid, wg in pm:
        agg[(k, cn)].append(v)
        meta[k] = (vg, ag, sg, lds, scr, grid, wg)
    for k, m in meta.items():
        print("kernel %s\n  vgpr %s agpr %s sgpr %s lds %s scratch %s grid %s wg %s" % ((k[:80],) + m))
    for (k, cn), vs in sorted(agg.items()):
        print("  %-28s n=%-4d mean=%.6g  (%s)" % (cn, len(vs), sum(vs) / len(vs), k[:40]))
