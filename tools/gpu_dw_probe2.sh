cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/overlap_probe.hip -o /tmp/overlap_probe -lpthread 2>/dev/null
timeout 120 /tmp/overlap_probe > gpurun_out/overlap_probe.txt 2>&1; cat gpurun_out/overlap_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/dw_prof -- python $GRAFT_REPO_ROOT/tools/bench_block.py --block-file $GRAFT_REPO_ROOT/.bench_blocks/ecdsa_10000_0.bin --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/dw_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/dw_prof.err
cd $GRAFT_REPO_ROOT; find gpurun_out/dw_prof -name "*kernel_stats.csv" | head -1 | xargs cat | cut -d, -f1-8 | head -30
