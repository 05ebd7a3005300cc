#!/bin/bash
# Which kernels of the device-side block walk get CUs of their own, and who waits for whom: eleven combinations in one gpurun call.
# The knob (FABGPU_WALK_SCHED: 1 hash checks after the gates, 2 hash checks on CUs of their own, 4 mid-states on CUs of their own,
# 8 mid-states on the main stream ahead of the gates) existed in fabgpu_api.hip::walk_block_pass for this measurement only; variant 4
# won (device phase 1.04 -> 0.90 ms) and is what the code now does unconditionally.  Kept as the record of the experiment.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=.bench_blocks/ecdsa_10000_0.bin
for s in 0 1 2 3 4 5 7 8 9 11 0; do
FABGPU_WALK_SCHED=$s FABGPU_PASS_TIMING=1 timeout 100 python tools/bench_block.py --block-file $B --steps 24 > gpurun_out/sched_$s.json 2> gpurun_out/sched_$s.err
python - <<PY
import json,re,statistics
d=json.loads(open("gpurun_out/sched_$s.json").read().strip().splitlines()[-1])
dev=[float(m.group(1)) for m in re.finditer(r"device ([0-9.]+)\)", open("gpurun_out/sched_$s.err").read())]
print("sched $s: ms/block median %.3f min %.3f | device phase median %.3f min %.3f" % (d["ms_per_block"], d["ms_min"], statistics.median(dev[4:]), min(dev[4:])))
PY
done
