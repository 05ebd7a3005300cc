#!/bin/bash
set -u
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python3 -m pytest tests/test_idemix_gpu.py tests/test_gpu_parity.py tests/test_device_walk.py -m gpu -x -q -k "auto or sha or hash or coop or walk or small" 2>&1 | tail -4
for v in "" ; do
  echo "== bench_cfg5_mixed $v"
  timeout 300 python3 tools/bench_cfg5_mixed.py $v 2>&1 | tail -n 1 | tee -a gpurun_out/r05_idemix_ab2.jsonl
done
rm -rf /tmp/kt
( cd /tmp && PMC_LAUNCHES=40 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/gpu_pmc_kernels.py nym > /tmp/kt.log 2>&1 )
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
echo "== kernel stats ($f)"; [ -n "$f" ] && cut -c1-160 "$f" | tee $R/gpurun_out/r05_idemix_kernel_stats.csv | head -8 || tail -5 /tmp/kt.log
