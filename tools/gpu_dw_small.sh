#!/bin/bash
# the device-route pass by block size (staging forced for the small ones), and its GPU tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_device_walk.py -m gpu -x -q 2>&1 | tail -3
for t in 100 300 1000 3000; do
    FABGPU_PASS_STAGE_MIN_BYTES=1 timeout 200 python tools/bench_block.py --tx $t --steps 16 > gpurun_out/dw_small_${t}.json 2>/dev/null
    python -c "
import json
d=json.loads(open('gpurun_out/dw_small_${t}.json').read().strip().splitlines()[-1]); print('$t tx: median %.3f ms min %.3f' % (d['ms_per_block'], d['ms_min']), d['routes']['device_walks'])"
done
FABGPU_PASS_TIMING=1 timeout 200 python tools/bench_block.py --block-file .bench_blocks/ecdsa_10000_0.bin --steps 16 --threads 3 > gpurun_out/dw3.json 2> gpurun_out/dw3.err; sed -n 10,12p gpurun_out/dw3.err | cut -c1-200
python -c "
import json
d=json.loads(open('gpurun_out/dw3.json').read().strip().splitlines()[-1]); print('10000 tx: median %.3f ms min %.3f' % (d['ms_per_block'], d['ms_min']), d['callers_in_flight'])"
