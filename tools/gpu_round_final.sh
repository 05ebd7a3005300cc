#!/bin/bash
# end-of-round evidence in one gpurun call: GPU test suite, the driver's bench command plain and under rocprofv3 (kernel trace + stats),
# the block pass (flags only / memo seeding / idle gaps), config 5, registered keys, the in-process multi
# dispatcher.  Summaries land in gpurun_out/ and are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $OUT/r02z_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r02z_pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/r02z_bench.json 2> $OUT/r02z_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/r02z_bench.json')); print('bench', d['value'], d['ms_per_step'], d['dispersion']['median_ms'], d['pcie_inclusive']['value'], d['configs3_fused']['value'], d['valu_roofline']['frac'], d['valu_roofline']['executed_frac'], d['cpu_baseline']['value'], d['cpu_baseline']['single_thread']['value'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats -d /tmp/prof_final -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r02z_bench_under_rocprof.json 2>/dev/null
python $R/profiles/summarize_rocprof.py $(find /tmp/prof_final -name "*.db") > $OUT/r02z_rocprof_final.txt 2>&1; head -8 $OUT/r02z_rocprof_final.txt
cd $R
for t in 10000 3000 1000 100; do python tools/bench_block.py --tx $t --steps 8 > $OUT/r02z_block_${t}_flags.json 2>/dev/null; python tools/bench_block.py --tx $t --steps 8 --memo > $OUT/r02z_block_${t}_memo.json 2>/dev/null; python -c "
import json
a=json.load(open('$OUT/r02z_block_${t}_flags.json')); b=json.load(open('$OUT/r02z_block_${t}_memo.json')); print('block', $t, 'flags', a['ms_per_block'], 'memo', b['ms_per_block'], b['memo_lookup_us_via_ctypes'])"; done
python tools/bench_block.py --tx 1000 --steps 10 --idle-ms 250 --memo 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('idle250 1k', d['ms_per_block'], d['ms_min'], d['ms_max'])"
python tools/bench_block.py --steps 8 --idle-ms 250 --memo 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('idle250 10k', d['ms_per_block'], d['ms_min'], d['ms_max'])"
python tools/bench_cfg5_mixed.py > $OUT/r02z_cfg5.json 2>/dev/null; cut -c1-300 $OUT/r02z_cfg5.json
python tools/bench_keyed.py > $OUT/r02z_keyed.json 2>/dev/null; cut -c1-200 $OUT/r02z_keyed.json
python tools/bench_multi.py --gpus 1 2>/dev/null | grep "^{" > $OUT/r02z_multi_g1.json; cut -c1-500 $OUT/r02z_multi_g1.json
