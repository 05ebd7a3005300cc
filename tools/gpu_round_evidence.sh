#!/bin/bash
# End-of-round evidence in ONE gpurun call (TAG=r05 ...): the whole GPU test suite, smoke(), the driver's bench command (stdout's LAST
# line is the bench line; the full detail is bench_detail.json), the same command under rocprofv3 --kernel-trace --stats (summary for
# profiles/), the N = 2 code path of bench.py on gloo (both ranks on the one GPU: numbers meaningless, the path must run), the PMC passes
# of the kernels the rooflines are about, and two kernel timelines of the block pass.  Everything lands in gpurun_out/${TAG}_final_*.
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r05}
cd $R
O=$R/gpurun_out
mkdir -p $O
if [ -z "${SKIP_TESTS:-}" ]; then
  ( time timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 ) > $O/${TAG}_final_pytest.txt 2>&1
  tail -4 $O/${TAG}_final_pytest.txt
fi
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_final_smoke.txt 2>&1; tail -1 $O/${TAG}_final_smoke.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/${TAG}_final_bench.out 2> $O/${TAG}_final_bench.err
tail -n1 $O/${TAG}_final_bench.out > $O/${TAG}_final_bench_line.json; wc -c $O/${TAG}_final_bench_line.json; cp $R/bench_detail.json $O/${TAG}_final_bench_detail.json
tail -3 $O/${TAG}_final_bench.err | grep real
rm -rf /tmp/prof_bench
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_final_bench_under_rocprof.out 2> $O/${TAG}_final_bench_under_rocprof.err )
tail -n1 $O/${TAG}_final_bench_under_rocprof.out > $O/${TAG}_final_bench_under_rocprof_line.json
f=$(find /tmp/prof_bench -name "*.db" -printf "%s %p\n" 2>/dev/null | sort -n | tail -1 | cut -d" " -f2)   # (bench.py starts helper processes: theirs are the small ones)
if [ -n "$f" ]; then python $R/profiles/summarize_rocprof.py "$f" > $O/${TAG}_final_rocprof_stats.txt 2>&1; python $R/profiles/timed_region_rocprof.py "$f" $O/${TAG}_final_bench_under_rocprof_line.json 5 20 >> $O/${TAG}_final_rocprof_stats.txt 2>&1; head -14 $O/${TAG}_final_rocprof_stats.txt | cut -c1-160; tail -8 $O/${TAG}_final_rocprof_stats.txt; else echo "no rocprof db"; fi
FABGPU_BENCH_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/${TAG}_final_bench_n2_gloo.out 2> $O/${TAG}_final_bench_n2_gloo.err
echo "n2 rc=$?"; tail -n1 $O/${TAG}_final_bench_n2_gloo.out > $O/${TAG}_final_bench_n2_gloo_line.json; wc -c $O/${TAG}_final_bench_n2_gloo_line.json
if [ -z "${SKIP_PMC:-}" ]; then TAG=$TAG bash $R/tools/gpu_pmc_sq.sh > $O/${TAG}_final_pmc.log 2>&1; grep -c "mean=" $O/${TAG}_pmc_sq.txt; fi
SIZES="100 10000" timeout 300 bash tools/gpu_timeline.sh > $O/${TAG}_final_timelines.txt 2>&1; grep -c "walk_" $O/${TAG}_final_timelines.txt
