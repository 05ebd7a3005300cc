#!/bin/bash
# round 6: the driver's bench command under rocprofv3 --kernel-trace --stats (twice: box-to-box and run-to-run spread of the timed region),
# summarised like tools/gpu_round_evidence.sh does, plus the dominant kernel's launches 56-95 one by one
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
for i in 1 2; do
  rm -rf /tmp/prof_bench
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_rocprof_run${i}.out 2> $O/r06_rocprof_run${i}.err )
  tail -n1 $O/r06_rocprof_run${i}.out > $O/r06_rocprof_run${i}_line.json
  f=$(find /tmp/prof_bench -name "*.db" -printf "%s %p\n" 2>/dev/null | sort -n | tail -1 | cut -d" " -f2)
  python $R/profiles/summarize_rocprof.py "$f" > $O/r06_rocprof_run${i}_stats.txt 2>&1
  python $R/profiles/timed_region_rocprof.py "$f" $O/r06_rocprof_run${i}_line.json 5 20 >> $O/r06_rocprof_run${i}_stats.txt 2>&1
  tail -8 $O/r06_rocprof_run${i}_stats.txt
  python3 - "$f" <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = sorted((s, e) for name, s, e in c.execute("select name, start, end from kernels") if "p256_verify_pair_lds_kernel" in name)
print("launches 56-95 (us):", " ".join("%.0f" % ((e - s) / 1e3) for s, e in rows[55:95]))
PY
  python3 -c "import json; d=json.loads(open('$O/r06_rocprof_run${i}_line.json').read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
