#!/bin/bash
# round 5: what the second and third pass of a fresh provider still pay for - stage breakdown of the first passes, four fresh processes
exec </dev/null
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export LD_LIBRARY_PATH=$R/fabric-mod_amd/lib:${LD_LIBRARY_PATH:-}
for k in 1 2 3 4; do
  echo "== fresh process $k"
  GO_REPLAY_PASS_TIMING=1 FABGPU_PASS_TIMING=1 $R/fabric-mod_amd/lib/go_call_replay $R/.bench_blocks/friendly_10000.bin 8 16 1 2> /tmp/err_$k.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('provider_new_ms','lone_passes_ms','pipelined_pass_ms_median','first_block_of_a_fresh_process')})"
  grep -n "fabgpu pass2\|arena stage\|fabgpu " /tmp/err_$k.txt | head -24
done 2>&1 | tee $R/gpurun_out/r05_fresh_provider_probe.txt
