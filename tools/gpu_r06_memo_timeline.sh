#!/bin/bash
# round 6: kernel AND copy timeline of one memo-seeding pass over the 10 000-transaction block (where the memo's early half ends
# relative to the verify launches)  ->  gpurun_out/r06_timeline_10000tx_memo.txt
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_tl
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_tl -- python $R/tools/bench_block.py --block-file $R/.bench_blocks/friendly_10000.bin --steps 6 --register-after 8 --memo > /dev/null 2>&1 )
f=$(find /tmp/prof_tl -name "*.db" | head -1)
python3 - "$f" > $R/gpurun_out/r06_timeline_10000tx_memo.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
k = list(c.execute("select name, queue_id, start, end from kernels order by start"))
firsts = [i for i, r in enumerate(k) if "walk_count" in r[0]]
t0 = k[firsts[-1]][2]
ev = [((s - t0) / 1e3, (e - t0) / 1e3, "q%-2d %s" % (q, n[:90])) for n, q, s, e in k[firsts[-1]:]]
try:
    cols = [r[1] for r in c.execute("pragma table_info(memory_copies)")]
    for row in c.execute("select * from memory_copies order by start"):
        d = dict(zip(cols, row))
        if d["start"] >= t0 - 2.5e6:
            ev.append(((d["start"] - t0) / 1e3, (d["end"] - t0) / 1e3, "COPY %s %s bytes" % (d.get("name"), d.get("size"))))
except Exception as ex:
    ev.append((0, 0, "no copy table: %r" % (ex,)))
print("one memo-seeding device-route pass (us from the count kernel; copies before it are the block's upload)")
for s, e, what in sorted(ev):
    print("%9.1f %9.1f  %s" % (s, e, what))
PY
cut -c1-140 $R/gpurun_out/r06_timeline_10000tx_memo.txt | tail -45
