#!/bin/bash
# round 5: per-kernel durations (rocprofv3 --kernel-trace --stats) and SQ counters of the idemix two-phase form, 6 000 signatures
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf /tmp/kt
( cd /tmp && PMC_LAUNCHES=40 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/tools/gpu_pmc_kernels.py nym > /tmp/kt.log 2>&1 )
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
echo "== kernel stats ($f)"; [ -n "$f" ] && cat "$f" | cut -c1-200 | tee $R/gpurun_out/r05_idemix_kernel_stats.csv | head -12 || tail -5 /tmp/kt.log
TAG=r05_idemix KERNELS=nym bash $R/tools/gpu_pmc_sq.sh
cat $R/gpurun_out/r05_idemix_pmc_sq.txt | cut -c1-260 | head -60
