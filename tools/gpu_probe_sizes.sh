#!/bin/bash
# the pass on friendly blocks of 100 / 1 000 / 10 000 tx, device route (stage timing) and host route
exec </dev/null
cd $GRAFT_REPO_ROOT
for n in 100 1000 10000; do
  if [ $n = 10000 ]; then BF="--block-file .bench_blocks/friendly_10000.bin"; else BF="--tx $n"; fi
  timeout 150 python tools/bench_block.py $BF --timing --steps 16 --register-after 8 2> /tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$n tx device route: pass %.3f ms (min %.3f)' % (d['ms_per_block'], d['ms_min']))"
  grep "fabgpu pass" /tmp/err.txt | tail -3
  timeout 150 python tools/bench_block.py $BF --host-walk --steps 16 --register-after 8 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$n tx host route:   pass %.3f ms (min %.3f)' % (d['ms_per_block'], d['ms_min']))"
done
