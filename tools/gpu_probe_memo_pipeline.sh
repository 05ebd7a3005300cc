#!/bin/bash
# two passes in flight WITH memo seeding (what the Go arrival hook runs): how much of their 2.5 ms per block is the memo's worker threads?
exec </dev/null
cd $GRAFT_REPO_ROOT
B=.bench_blocks/friendly_10000.bin
for gt in 1 2 4 8 16; do
  FABGPU_PASS_GATE_THREADS=$gt timeout 120 python tools/bench_block.py --block-file $B --threads 2 --memo --steps 8 --register-after 64 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('memo threads <= $gt: two callers, memo seeding: %.2f ms per block aggregate' % (d.get('ms_per_block')))"
done
timeout 120 python tools/bench_block.py --block-file $B --threads 2 --steps 8 --register-after 64 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('two callers, flags only: %.2f ms per block aggregate' % (d.get('ms_per_block')))"
