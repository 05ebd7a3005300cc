#!/bin/bash
# kernel timelines of one device-route pass: friendly blocks of 100 and 10 000 tx  ->  gpurun_out/timeline_<n>.txt
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in ${SIZES:-100 10000}; do
  if [ $n = 10000 ]; then BF="--block-file $R/.bench_blocks/friendly_10000.bin"; else BF="--tx $n"; fi
  rm -rf /tmp/prof_tl
  cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -- python $R/tools/bench_block.py $BF --steps 6 --register-after 8 > /dev/null 2>&1
  f=$(find /tmp/prof_tl -name "*.db" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python $R/profiles/timeline_rocprof.py "$f" > $R/gpurun_out/timeline_$n.txt 2>&1; cut -c1-130 $R/gpurun_out/timeline_$n.txt | head -32; else echo "no db for $n"; fi
done
