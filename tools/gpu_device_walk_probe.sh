#!/bin/bash
# The device-side block walk in one gpurun call: its GPU tests, then the 10 000-tx block pass with the walk on the device and on the
# host (FABGPU_PASS_DEVICE_WALK=0), flags only and with memo seeding, stage timing on stderr.  Blocks: tools/make_bench_blocks.py ecdsa 10000 0
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B=.bench_blocks/ecdsa_10000_0.bin
timeout 400 python -m pytest tests/test_device_walk.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/dw_tests.log
tail -5 gpurun_out/dw_tests.log
run() { n=$1; shift; env "$@" FABGPU_PASS_TIMING=1 timeout 200 python tools/bench_block.py --block-file $B --steps 12 $EXTRA > gpurun_out/dw_$n.json 2> gpurun_out/dw_$n.err; tail -3 gpurun_out/dw_$n.err | cut -c1-250; python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/dw_$n.json").read().strip().splitlines()[-1])
    print("$n", round(d["ms_per_block"], 3), "ms/block median,", round(d["ms_min"], 3), "min,", int(d["value"]), "tx/s", d.get("routes"))
except Exception as e:
    print("$n FAILED", e)
PY
}
run device A=1
run host FABGPU_PASS_DEVICE_WALK=0
EXTRA=--memo run device_memo A=1
EXTRA=--memo run host_memo FABGPU_PASS_DEVICE_WALK=0
EXTRA="--threads 3" run device_3callers A=1
