#!/bin/bash
# HBM-side traffic of the dominant kernel per 30 000-tuple launch, both homes of the pair kernel's per-signature table:
# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (TCC slots: MI355X_MICROARCH.md "rocprofv3 PMC slots") over
# `python bench.py --steps 10 --warmup 2 --no-extras`, summarised by profiles/summarize_rocprof.py into gpurun_out/r03_pmc_<table>_<counter>.txt
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in global lds; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${mode}_$c
    ( cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${mode}_$c -- python $R/bench.py --steps 10 --warmup 2 --no-extras --pair-table $mode > /dev/null 2>&1 )
    f=$(find /tmp/pmc_${mode}_$c -name "*.db" 2>/dev/null | head -1)
    if [ -n "$f" ]; then python $R/profiles/summarize_rocprof.py "$f" > $R/gpurun_out/r03_pmc_${mode}_$c.txt 2>&1; grep -A3 "kernel .*p256_verify_pair" $R/gpurun_out/r03_pmc_${mode}_$c.txt | grep -v "^==" | head -4 | cut -c1-160; grep "$c" $R/gpurun_out/r03_pmc_${mode}_$c.txt | grep p256 | head -2; else echo "no db for $mode $c"; fi
  done
done
