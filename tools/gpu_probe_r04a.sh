#!/bin/bash
# round-4 probe A: the memo pipeline with and without construction-time allocation, the Go replay, a timeline of the oversize-identity block
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for cp in 0 3; do echo "== PROBE_CP=$cp memo"; PROBE_CP=$cp ROUNDS=2 timeout 120 python tools/gpu_probe_memo_pipeline.py 10 1 2>&1 | cut -c1-260; done
echo "== PROBE_CP=3 flags"; PROBE_CP=3 timeout 120 python tools/gpu_probe_memo_pipeline.py 10 0 2>&1 | cut -c1-200
echo "== replay"; timeout 120 ./fabric-mod_amd/lib/go_call_replay .bench_blocks/friendly_10000.bin 12 16 1
