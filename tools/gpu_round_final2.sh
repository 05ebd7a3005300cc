#!/bin/bash
# end-of-round evidence after the device-side block walk: the GPU test suite, the driver's bench command plain and under rocprofv3
# (kernel trace + stats).  Summaries land in gpurun_out/ and are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; cd $R; mkdir -p $OUT
( time timeout 900 python -m pytest tests -m gpu -q ) > $OUT/r02f_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/r02f_pytest.log | head -4
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/r02f_bench.json 2> $OUT/r02f_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('$OUT/r02f_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['valu_roofline']['frac'], d['cpu_baseline']['value']); print(json.dumps(d.get('block_pass'))[:1500])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r02f_bench_under_rocprof.json 2>/dev/null
python $R/profiles/summarize_rocprof.py $(find /tmp/prof_final -name "*.db") > $OUT/r02f_rocprof.txt 2>&1; head -12 $OUT/r02f_rocprof.txt | cut -c1-120
