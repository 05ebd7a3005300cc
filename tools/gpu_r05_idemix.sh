#!/bin/bash
# round 5: the idemix two-phase form - parity on every variant, then A/B against round 4's fused kernel, alone and in the mixed batch
set -u
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_idemix_gpu.py tests/test_idemix_nym_kats.py -m gpu -x -q 2>&1 | tail -5
for v in "" "--fused-hash" "--no-quad"; do
  echo "== bench_cfg5_mixed $v"
  timeout 300 python3 tools/bench_cfg5_mixed.py $v 2>&1 | tail -n 1 | tee -a gpurun_out/r05_idemix_ab.jsonl
done
