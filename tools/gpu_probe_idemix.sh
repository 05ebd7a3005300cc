#!/bin/bash
# the 10 000-tx block with every 5th creator idemix through the pass on both routes (stage timing), then a kernel timeline of the device route
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
B=.bench_blocks/idemix_10000_5.bin
timeout 200 python tools/bench_block.py --timing --block-file $B --idemix --steps 8 --register-after 64 > gpurun_out/probe_idemix_dev.json 2> gpurun_out/probe_idemix_dev.err
tail -2 gpurun_out/probe_idemix_dev.err | cut -c1-220; cut -c1-260 gpurun_out/probe_idemix_dev.json
timeout 200 python tools/bench_block.py --host-walk --block-file $B --idemix --steps 8 --register-after 64 2>/dev/null | cut -c1-200
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_idemix -- python $R/tools/bench_block.py --block-file $R/$B --idemix --steps 6 --register-after 64 > /dev/null 2>&1
f=$(find /tmp/prof_idemix -name "*.db" 2>/dev/null | head -1)
if [ -n "$f" ]; then python $R/profiles/timeline_rocprof.py "$f" > $R/gpurun_out/probe_idemix_timeline.txt 2>&1; cut -c1-150 $R/gpurun_out/probe_idemix_timeline.txt | head -30; else echo "no db"; fi
