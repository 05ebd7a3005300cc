#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { echo "== $1"; shift; env "$@" FABGPU_PASS_TIMING=1 python tools/bench_block.py --steps 8 $EXTRA 2>&1 | grep -E "fabgpu pass|ms_per_block" | tail -3 | cut -c1-230; }
run "default (walk 8, gates 16)" A=1
run "gates 8" FABGPU_PASS_GATE_THREADS=8
run "gates 4" FABGPU_PASS_GATE_THREADS=4
run "walk 6 gates 8" FABGPU_PASS_WALK_THREADS=6 FABGPU_PASS_GATE_THREADS=8
EXTRA=--memo run "memo, default" A=1
EXTRA=--memo run "memo, gates 8" FABGPU_PASS_GATE_THREADS=8
EXTRA="--tx 1000" run "1k tx" A=1
EXTRA="--tx 3000" run "3k tx" A=1
