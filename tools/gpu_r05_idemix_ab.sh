#!/bin/bash
# round 5: the idemix four-lane form - parity of the default (three launches) and A/B against its predecessors, alone and in the mixed batch
set -u
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python3 -m pytest tests/test_idemix_gpu.py tests/test_idemix_nym_kats.py -m gpu -x -q -k "${PYTEST_K:-auto or one-stream or kats}" 2>&1 | tail -4
for v in "" "--no-side-stream" "--fused-hash"; do
  echo "== bench_cfg5_mixed $v"
  timeout 300 python3 tools/bench_cfg5_mixed.py $v 2>&1 | tail -n 1 | tee -a gpurun_out/r05_idemix_ab3.jsonl | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['idemix_alone'], d['ms_per_step'])"
done
rm -rf /tmp/kt
( cd /tmp && PMC_LAUNCHES=40 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/gpu_pmc_kernels.py nym > /tmp/kt.log 2>&1 )
f=$(find /tmp/kt -name "*.db" | head -1)
[ -n "$f" ] && python3 $R/profiles/summarize_rocprof.py "$f" 2>&1 | grep -v rocclr | head -8 | cut -c1-150 | tee $R/gpurun_out/r05_idemix_kernel_stats.txt
