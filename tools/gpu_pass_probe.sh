#!/bin/bash
# where the block pass's HOST time goes: staging thread on/off, walk threads, memo on/off (FABGPU_PASS_TIMING prints the stages)
R=$GRAFT_REPO_ROOT; cd $R
python tools/make_walk_block.py /tmp/blk.bin
echo "walker alone (hot): r01 / r02 / r02b, 16 threads then 1"; for b in walk_r01 walk_r02 walk_r02b; do fabric-mod_amd/lib/$b /tmp/blk.bin 16 | tail -1; fabric-mod_amd/lib/$b /tmp/blk.bin 1 | tail -1; done
run() { echo "== $1"; shift; env "$@" FABGPU_PASS_TIMING=1 python tools/bench_block.py --steps 6 2>&1 | grep -E "fabgpu pass|ms_per_block" | tail -4 | cut -c1-260; }
run "default" A=1
run "no staging thread" FABGPU_PASS_STAGE_MIN_BYTES=999999999
run "walk threads 8" FABGPU_PASS_WALK_THREADS=8
run "walk threads 4" FABGPU_PASS_WALK_THREADS=4
run "walk threads 1" FABGPU_PASS_WALK_THREADS=1
run "walk threads 4, no staging" FABGPU_PASS_WALK_THREADS=4 FABGPU_PASS_STAGE_MIN_BYTES=999999999
