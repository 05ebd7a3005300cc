#!/bin/bash
# round 5: what the second and third pass of a fresh provider still pay for - stage breakdown of the first passes of
# N fresh processes (default 4); passes slower than 2.6 ms after the first are printed with their stage clocks
exec </dev/null
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export LD_LIBRARY_PATH=$R/fabric-mod_amd/lib:${LD_LIBRARY_PATH:-}
for k in $(seq 1 ${N:-4}); do
  echo "== fresh process $k"
  GO_REPLAY_PASS_TIMING=1 FABGPU_PASS_TIMING=1 $R/fabric-mod_amd/lib/go_call_replay $R/.bench_blocks/friendly_10000.bin 8 16 1 2> /tmp/err_$k.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('provider_new_ms','lone_passes_ms','pipelined_pass_ms_median')})"
  grep -n "fabgpu pass2\|arena stage\|pinned memory" /tmp/err_$k.txt | awk 'BEGIN{p=0} /fabgpu pass2/{p++; split($0,a,"total "); split(a[2],b," "); if (p>1 && b[1]+0>2.6) print "   SLOW pass " p ": " $0; next} {last=$0}' 
  if [ -n "${VERBOSE:-}" ]; then grep -n "fabgpu pass2\|arena stage" /tmp/err_$k.txt | head -24; fi
done 2>&1 | tee $R/gpurun_out/r05_fresh_provider_probe2.txt
