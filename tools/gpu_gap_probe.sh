#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python tools/make_walk_block.py /tmp/blk.bin
for t in 16 8 4 1; do echo "r01 walker threads=$t"; fabric-mod_amd/lib/walk_r01 /tmp/blk.bin $t | tail -3; echo "r02 walker threads=$t"; fabric-mod_amd/lib/walk_r02 /tmp/blk.bin $t | tail -3; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_gap
rocprofv3 --kernel-trace -d /tmp/prof_gap -- python $R/bench.py --steps 20 --warmup 5 --no-extras > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/prof_gap/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tabs if 'kernel_dispatch' in x.lower() or x == 'kernels']
print(t[:10])
for name in ('kernels',) + tuple(t):
    try:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % name)]
        if 'start' in cols and 'end' in cols:
            rows = list(c.execute("select start, end from %s order by start" % name))
            rows = [(s, e) for s, e in rows if e - s > 300000]
            gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
            durs = [e - s for s, e in rows]
            print(name, 'kernels', len(rows), 'dur us', sum(durs) / len(durs) / 1e3, 'gaps us', sorted(g / 1e3 for g in gaps)[:5], sorted(g / 1e3 for g in gaps)[len(gaps) // 2], sorted(g / 1e3 for g in gaps)[-5:])
            break
    except Exception as ex:
        print(name, ex)
PY
