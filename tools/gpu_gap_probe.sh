#!/bin/bash
# gaps between back-to-back launches of the dominant kernel, from the rocprofv3 kernel trace of the bench command
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_gap
rocprofv3 --kernel-trace -d /tmp/prof_gap -- python $R/bench.py --steps 20 --warmup 5 --no-extras > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/prof_gap/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = [(s, e) for s, e in c.execute("select start, end from kernels order by start") if e - s > 300000]
gaps = sorted((rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1))
durs = [e - s for s, e in rows]
print('kernels', len(rows), 'mean duration us', sum(durs) / len(durs) / 1e3, 'gaps us: min', gaps[:3], 'median', gaps[len(gaps) // 2], 'max', gaps[-3:])
PY
