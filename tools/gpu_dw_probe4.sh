#!/bin/bash
# as gpu_dw_probe3.sh, the bench twice (box noise), trace without the stats table: the timelines in profiles/r02_device_walk_timeline.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_device_walk.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/dw_tests.log; tail -1 gpurun_out/dw_tests.log
B=.bench_blocks/ecdsa_10000_0.bin
for k in 1 2; do
FABGPU_PASS_TIMING=1 timeout 200 python tools/bench_block.py --block-file $B --steps 16 --threads 3 > gpurun_out/dw3.json 2> gpurun_out/dw3.err; sed -n 10,11p gpurun_out/dw3.err | cut -c1-160
python -c "
import json
d=json.loads(open('gpurun_out/dw3.json').read().strip().splitlines()[-1]); print(round(d['ms_per_block'],3), round(d['ms_min'],3), d['callers_in_flight'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/dw_prof3 -- python $GRAFT_REPO_ROOT/tools/bench_block.py --block-file $GRAFT_REPO_ROOT/$B --steps 20 > /dev/null 2>&1
