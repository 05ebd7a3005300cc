#!/bin/bash
# round 5: the driver's command (N=1) and the N=2 gloo dry run of the multi-rank code path; only stdout's LAST line is the bench line
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_n1.out 2> gpurun_out/r05_bench_n1.err
echo "n1 rc=$? bytes=$(tail -n1 gpurun_out/r05_bench_n1.out | wc -c)"
tail -n1 gpurun_out/r05_bench_n1.out
FABGPU_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r05_bench_n2_gloo.out 2> gpurun_out/r05_bench_n2_gloo.err
echo "n2 rc=$? bytes=$(tail -n1 gpurun_out/r05_bench_n2_gloo.out | wc -c)"
tail -n1 gpurun_out/r05_bench_n2_gloo.out
