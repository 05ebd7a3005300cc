#!/bin/bash
# round 5: kernel timeline of the first passes of a fresh provider (which kernels pass 2 runs that pass 3 does not; gaps vs durations)
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export LD_LIBRARY_PATH=$R/fabric-mod_amd/lib:${LD_LIBRARY_PATH:-}
rm -rf /tmp/ft
( cd /tmp && GO_REPLAY_PASS_TIMING=1 timeout 200 rocprofv3 --kernel-trace -d /tmp/ft -- $R/fabric-mod_amd/lib/go_call_replay $R/.bench_blocks/friendly_10000.bin 4 16 1 > /tmp/ft.out 2> /tmp/ft.err )
grep "fabgpu pass2" /tmp/ft.err | head -8
f=$(find /tmp/ft -name "*.db" | head -1)
python3 $R/tools/gpu_trace_dump.py "$f" > $R/gpurun_out/r05_fresh_trace.txt 2>&1
wc -l $R/gpurun_out/r05_fresh_trace.txt
grep -v "walk_warm\|probe" $R/gpurun_out/r05_fresh_trace.txt | cut -c1-150 | tail -n +1 | head -150
