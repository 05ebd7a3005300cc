#!/bin/bash
# round 5: stream priorities for the mixed batch (configs[4] on one GPU), with and without the timing events of FABGPU_FLAG_TIME_KERNELS,
# and with a caller-side marker after every ECDSA launch
for v in "--no-time-kernels" "--no-time-kernels --prio=-1,0" "--no-time-kernels --prio=-1,0 --marker timing" "--no-time-kernels --prio=-1,0 --marker plain" "--no-time-kernels --marker timing" "--prio=-1,0"; do echo "== $v"; timeout 300 python3 tools/bench_cfg5_mixed.py --warmup 40 $v 2>&1 | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('idemix alone %.3f  ecdsa alone %.3f  mixed step %.3f ms' % (d['idemix_alone']['ms_per_step'], d['ecdsa_alone']['ms_per_step'], d['ms_per_step']))"; done
