#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python tools/make_walk_block.py /tmp/blk.bin
echo "walker alone: scouts / serial"; fabric-mod_amd/lib/walk_r02d /tmp/blk.bin 8 | tail -2; FABGPU_PASS_NO_SPECULATIVE_LISTING=1 fabric-mod_amd/lib/walk_r02d /tmp/blk.bin 8 | tail -2
run() { echo "== $1"; shift; env "$@" FABGPU_PASS_TIMING=1 python tools/bench_block.py --steps 8 $EXTRA 2>&1 | grep -E "fabgpu pass|ms_per_block" | tail -3 | cut -c1-200; }
run "10k scouts" A=1
run "10k serial" FABGPU_PASS_NO_SPECULATIVE_LISTING=1
run "10k scouts" A=1
run "10k serial" FABGPU_PASS_NO_SPECULATIVE_LISTING=1
EXTRA=--memo run "10k memo scouts" A=1
