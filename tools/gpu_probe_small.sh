#!/bin/bash
# small friendly blocks on both routes (flags only, and with memo seeding): device walk against --host-walk
exec </dev/null
cd $GRAFT_REPO_ROOT
for n in 5 10 20 50 100 300; do
  for memo in "" "--memo"; do
    d=$(timeout 100 python tools/bench_block.py --tx $n --steps 24 --register-after 8 $memo 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f' % d['ms_per_block'], d.get('block_bytes', ''))")
    h=$(timeout 100 python tools/bench_block.py --host-walk --tx $n --steps 24 --register-after 8 $memo 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f' % d['ms_per_block'])")
    echo "$n tx $memo: device route $d   host route $h"
  done
done
