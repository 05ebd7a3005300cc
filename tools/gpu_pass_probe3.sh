#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_block_prepass.py tests/test_ledger_goldens.py -m gpu -q 2>&1 | tail -2
run() { echo "== $1"; shift; env "$@" FABGPU_PASS_TIMING=1 python tools/bench_block.py --steps 8 $EXTRA 2>&1 | grep -E "fabgpu pass|ms_per_block" | tail -3 | cut -c1-230; }
run "10k default" A=1
run "10k default again" A=1
EXTRA=--memo run "10k memo" A=1
EXTRA="--tx 3000" run "3k tx" A=1
EXTRA="--tx 1000" run "1k tx" A=1
EXTRA="--tx 1000 --memo" run "1k tx memo" A=1
EXTRA="--tx 100" run "100 tx" A=1
