#!/bin/bash
# one gpurun call: a block through the pass with stage timing, then a rocprofv3 kernel trace of the same.  usage: gpu_probe_distinct.sh <block file> <tag>
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B=${1:-.bench_blocks/distinct_10000.bin}
T=${2:-distinct}
cd $R
timeout 120 python tools/bench_block.py --timing --block-file $B --steps 6 --register-after 64 > gpurun_out/probe_$T.json 2> gpurun_out/probe_$T.err
tail -3 gpurun_out/probe_$T.err | cut -c1-250
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -- python $R/tools/bench_block.py --block-file $R/$B --steps 6 --register-after 64 > /dev/null 2>&1
f=$(find /tmp/prof_$T -name "*.db" 2>/dev/null | head -1)
if [ -n "$f" ]; then python $R/profiles/summarize_rocprof.py "$f" > $R/gpurun_out/probe_${T}_kernels.txt 2>&1; python $R/profiles/timeline_rocprof.py "$f" > $R/gpurun_out/probe_${T}_timeline.txt 2>&1; cut -c1-150 $R/gpurun_out/probe_${T}_timeline.txt | head -40; else echo "no db"; fi
