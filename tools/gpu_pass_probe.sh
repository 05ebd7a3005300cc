#!/bin/bash
# where the block pass's HOST time goes, in one gpurun call: the walker alone (hot, by thread count), then the pass with
# FABGPU_PASS_TIMING for several thread settings, with and without the upload thread, with memo seeding, by block size
R=$GRAFT_REPO_ROOT; cd $R
python tools/make_walk_block.py /tmp/blk.bin
g++ -O3 -std=c++17 -Ifabric-mod_amd/csrc tools/walk_harness.cpp fabric-mod_amd/csrc/block_prepass.cpp -o /tmp/walk -lpthread 2>/dev/null
for t in 16 8 4 1; do echo "walker alone, threads=$t"; /tmp/walk /tmp/blk.bin $t | tail -2; done
run() { echo "== $1"; shift; env "$@" FABGPU_PASS_TIMING=1 python tools/bench_block.py --steps 8 $EXTRA 2>&1 | grep -E "fabgpu pass|ms_per_block" | tail -3 | cut -c1-230; }
run "10k default (8 walk workers)" A=1
run "10k walk threads 16" FABGPU_PASS_WALK_THREADS=16
run "10k walk threads 4" FABGPU_PASS_WALK_THREADS=4
run "10k walk threads 1" FABGPU_PASS_WALK_THREADS=1
run "10k no staging thread" FABGPU_PASS_STAGE_MIN_BYTES=999999999
run "10k gates 8" FABGPU_PASS_GATE_THREADS=8
EXTRA=--memo run "10k memo" A=1
EXTRA="--tx 3000" run "3k tx" A=1
EXTRA="--tx 1000" run "1k tx" A=1
EXTRA="--tx 100" run "100 tx" A=1
