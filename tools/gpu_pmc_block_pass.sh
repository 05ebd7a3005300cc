#!/bin/bash
# HBM-side traffic of every kernel of a device-route pass over the friendly 10 000-tx block (48.6 MB), flags only and with memo seeding:
# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (--kernel-trace only), summarised by profiles/summarize_rocprof.py
# into gpurun_out/r03_pmc_block_pass_<mode>_<counter>.txt  (FETCH_SIZE is to be doubled on gfx950: MI355X_MICROARCH.md "HBM")
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in flags memo; do
  M=""; [ $mode = memo ] && M="--memo"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcb_${mode}_$c
    ( cd /tmp && timeout 150 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmcb_${mode}_$c -- python $R/tools/bench_block.py --block-file $R/.bench_blocks/friendly_10000.bin --steps 6 --register-after 8 $M > /dev/null 2>&1 )
    f=$(find /tmp/pmcb_${mode}_$c -name "*.db" 2>/dev/null | head -1)
    if [ -n "$f" ]; then python $R/profiles/summarize_rocprof.py "$f" > $R/gpurun_out/r03_pmc_block_pass_${mode}_$c.txt 2>&1; grep "$c" $R/gpurun_out/r03_pmc_block_pass_${mode}_$c.txt | cut -c1-120 | head -24; else echo "no db for $mode $c"; fi
  done
done
