#!/usr/bin/env python3
"""One pass of the block pre-verify pass as a timeline, out of a rocprofv3 --kernel-trace result (rocpd sqlite): the kernels between
the LAST walk_count kernel and the end of the trace, start / end in us from that kernel's start, with the HSA queue (= stream).
usage: timeline_rocprof.py <results.db> [which]     which: -1 = last pass (default), -2 = the one before ..."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
rows = list(c.execute("select name, queue_id, start, end from kernels order by start"))
firsts = [i for i, r in enumerate(rows) if "walk_count" in r[0]]
if not firsts:
    sys.exit("no walk_count kernel in the trace")
lo = firsts[which]
hi = firsts[which + 1] if which != -1 and which + 1 < 0 else len(rows)
t0 = rows[lo][2]
print("one device-route pass under rocprofv3 --kernel-trace (us from the first kernel; q = HSA queue = stream)")
for name, q, s, e in rows[lo:hi]:
    print("%8.1f %8.1f  q%-2d %s" % ((s - t0) / 1e3, (e - t0) / 1e3, q, name[:100]))
