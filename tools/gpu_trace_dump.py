#!/usr/bin/env python3
"""Dump a rocprofv3 (rocpd sqlite) trace as one merged timeline of kernels and memory copies between two times.
usage: gpu_trace_dump.py <results.db> [t0_ms t1_ms]   (times relative to the first kernel of the trace)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
rows = []
if "kernels" in names:
    for name, q, s, e in c.execute("select name, queue_id, start, end from kernels"):
        rows.append((s, e, "K q%-2s %s" % (q, name[:90])))
mc = [n for n in names if n.startswith("memory_cop")]
for n in mc[:1]:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % n)]
    sel = [x for x in ("name", "start", "end", "size", "src_agent_type", "dst_agent_type") if x in cols]
    for r in c.execute("select %s from %s" % (",".join(sel), n)):
        d = dict(zip(sel, r))
        rows.append((d["start"], d["end"], "C %s %s bytes %s->%s" % (d.get("name", ""), d.get("size", "?"), d.get("src_agent_type", "?"), d.get("dst_agent_type", "?"))))
rows.sort()
if not rows:
    sys.exit("empty trace; tables: %s" % names)
t00 = rows[0][0]
lo = float(sys.argv[2]) if len(sys.argv) > 3 else -1e18
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
prev_end = None
for s, e, what in rows:
    a, b = (s - t00) / 1e6, (e - t00) / 1e6
    if a < lo or a > hi:
        continue
    print("%10.3f %10.3f  %8.1f us  %s" % (a, b, (e - s) / 1e3, what))
