#!/bin/bash
# the pinned staging upload of the friendly 10 000-tx block: upload queues x copier threads (median whole-pass time, mean "arena stage")
exec </dev/null
cd $GRAFT_REPO_ROOT
B=.bench_blocks/friendly_10000.bin
for q in 2 4; do for t in 2 4 8 12; do for kb in 2048; do
  FABGPU_STAGE_QUEUES=$q FABGPU_STAGE_THREADS=$t FABGPU_STAGE_PIECE_KB=$kb FABGPU_PASS_TIMING=1 timeout 120 python tools/bench_block.py --block-file $B --steps 12 --register-after 64 2> /tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('queues $q threads $t piece $kb KB: pass %.2f ms (min %.2f)' % (d['ms_per_block'], d['ms_min']), end='  ')"
  grep "arena stage" /tmp/err.txt | tail -8 | awk '{s+=$7; n++} END {printf "stage mean %.2f ms\n", s/n}'
done; done; done
