#!/bin/bash
# End-of-round evidence in ONE gpurun call: the whole GPU test suite, smoke(), the driver's bench line, the same command under
# rocprofv3 --kernel-trace --stats (summary for profiles/), the N = 2 code path of bench.py on gloo (both ranks on the one GPU:
# numbers meaningless, the path must run), and two kernel timelines of the block pass.  Everything lands in gpurun_out/r04_final_*.
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 ) > $O/r04_final_pytest.txt 2>&1
tail -4 $O/r04_final_pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r04_final_smoke.txt 2>&1; tail -1 $O/r04_final_smoke.txt
( time timeout 600 python bench.py ) > $O/r04_final_bench.json 2> $O/r04_final_bench.err; tail -3 $O/r04_final_bench.err | cut -c1-200; wc -c $O/r04_final_bench.json
rm -rf /tmp/prof_bench
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python $R/bench.py > $O/r04_final_bench_under_rocprof.json 2> $O/r04_final_bench_under_rocprof.err )
f=$(find /tmp/prof_bench -name "*.db" -printf "%s %p\n" 2>/dev/null | sort -n | tail -1 | cut -d" " -f2)   # (bench.py starts helper processes: theirs are the small ones)
if [ -n "$f" ]; then python $R/profiles/summarize_rocprof.py "$f" > $O/r04_final_rocprof_stats.txt 2>&1; head -12 $O/r04_final_rocprof_stats.txt | cut -c1-160; else echo "no rocprof db"; fi
FABGPU_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 > $O/r04_final_bench_n2_gloo.json 2> $O/r04_final_bench_n2_gloo.err
echo "n2 rc=$?"; cut -c1-300 $O/r04_final_bench_n2_gloo.json | tail -2
SIZES="100 10000" timeout 300 bash tools/gpu_timeline.sh > $O/r04_final_timelines.txt 2>&1; grep -c "walk_" $O/r04_final_timelines.txt
