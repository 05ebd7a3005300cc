#!/bin/bash
# where the block pass spends its time (FABGPU_PASS_TIMING), memo seeding cost, idle-clock latency with and without the warm-up kernel
R=$GRAFT_REPO_ROOT
cd $R
echo "== back to back, flags only"; FABGPU_PASS_TIMING=1 python tools/bench_block.py --steps 6 2>&1 | tail -8 | cut -c1-900
echo "== back to back, memo"; python tools/bench_block.py --steps 6 --memo 2>/dev/null | cut -c1-900
for warm in 0 500 1500 3000; do
  echo "== 10k tx, 250 ms idle, warm $warm us"; FABGPU_PASS_WARM_US=$warm python tools/bench_block.py --steps 8 --idle-ms 250 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_block'], d['ms_min'], d['ms_max'])"
done
for warm in 0 300 1000; do
  echo "== 1k tx, 250 ms idle, warm $warm us"; FABGPU_PASS_WARM_US=$warm python tools/bench_block.py --tx 1000 --steps 8 --idle-ms 250 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_block'], d['ms_min'], d['ms_max'])"
done
echo "== 1k tx back to back"; python tools/bench_block.py --tx 1000 --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_block'], d['ms_min'], d['ms_max'])"
python bench.py --steps 20 --warmup 5 > gpurun_out/r02f_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02f_bench.json')); print('bench', d['value'], d['ms_per_step'], d['dispersion']['median_ms'], d['pcie_inclusive']['value'], d['configs3_fused']['value'], d['configs3_fused']['median_ms'])"
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fault or cfg4 or arena" 2>&1 | tail -3
