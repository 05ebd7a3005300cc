#!/bin/bash
# HBM traffic of the device-side block walk's kernels (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes as the guide
# prescribes; kernel-trace only), and the pass by block size on both routes.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc_$c
timeout 200 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -- python $R/tools/bench_block.py --block-file $R/.bench_blocks/ecdsa_10000_0.bin --steps 6 > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $(find /tmp/pmc_$c -name "*.db") > $OUT/dw_pmc_$c.txt 2>&1
grep -E "mean=" $OUT/dw_pmc_$c.txt | grep -E "walk_|sha256_|gather|verify_keyed" | cut -c1-120
done
cd $R
for t in 3000 1000 300; do
  for route in device host; do
    if [ $route = host ]; then export FABGPU_PASS_DEVICE_WALK=0; else unset FABGPU_PASS_DEVICE_WALK; fi
    FABGPU_PASS_STAGE_MIN_BYTES=1 timeout 200 python tools/bench_block.py --tx $t --steps 12 > $OUT/dw_size_${t}_$route.json 2>/dev/null
    python -c "
import json
d=json.loads(open('$OUT/dw_size_${t}_$route.json').read().strip().splitlines()[-1]); print('$t tx $route: median %.3f ms min %.3f' % (d['ms_per_block'], d['ms_min']), d['routes'])"
  done
done
