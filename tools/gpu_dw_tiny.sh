#!/bin/bash
# blocks of 5 .. 100 transactions on both routes (staging forced): where the device walk starts to win
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for t in 5 20 50 100; do
  for route in device host; do
    if [ $route = host ]; then export FABGPU_PASS_DEVICE_WALK=0; else unset FABGPU_PASS_DEVICE_WALK; fi
    FABGPU_PASS_STAGE_MIN_BYTES=1 timeout 200 python tools/bench_block.py --tx $t --steps 24 > gpurun_out/dw_tiny_${t}_$route.json 2>/dev/null
    python -c "
import json
d=json.loads(open('gpurun_out/dw_tiny_${t}_$route.json').read().strip().splitlines()[-1]); print('$t tx $route: median %.3f ms min %.3f' % (d['ms_per_block'], d['ms_min']), d['routes']['device_walks'])"
  done
done
