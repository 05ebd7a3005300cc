#!/bin/bash
# end-of-round evidence in one gpurun call: GPU test suite, the driver's bench command under rocprofv3 (kernel trace + stats) and plain,
# the block pass (flags only / memo seeding), config 5.  Summaries land in gpurun_out/ and are copied to profiles/ by hand.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -q > $OUT/r02g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r02g_pytest.log
python bench.py --steps 20 --warmup 5 > $OUT/r02g_bench.json 2> $OUT/r02g_bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$OUT/r02g_bench.json')); print('bench', d['value'], d['ms_per_step'], d['dispersion'], d['pcie_inclusive']['value'], d['configs3_fused']['value'], d['configs3_fused']['median_ms'], d['valu_roofline']['frac'], d['valu_roofline']['executed_frac'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats -d /tmp/prof_final -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r02g_bench_under_rocprof.json 2>/dev/null
python $R/profiles/summarize_rocprof.py $(find /tmp/prof_final -name "*.db") > $OUT/r02g_rocprof_final.txt 2>&1; head -12 $OUT/r02g_rocprof_final.txt
cd $R
python tools/bench_block.py --steps 8 > $OUT/r02g_block_flags.json 2>/dev/null; cut -c1-420 $OUT/r02g_block_flags.json
python tools/bench_block.py --steps 8 --memo > $OUT/r02g_block_memo.json 2>/dev/null; cut -c1-520 $OUT/r02g_block_memo.json
python tools/bench_block.py --tx 1000 --steps 8 --memo > $OUT/r02g_block_memo_1k.json 2>/dev/null; cut -c1-420 $OUT/r02g_block_memo_1k.json
python tools/bench_cfg5_mixed.py > $OUT/r02g_cfg5.json 2>/dev/null; cut -c1-600 $OUT/r02g_cfg5.json
python tools/bench_keyed.py > $OUT/r02g_keyed.json 2>/dev/null; cut -c1-500 $OUT/r02g_keyed.json
