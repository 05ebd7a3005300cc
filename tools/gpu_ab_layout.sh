#!/bin/bash
# A/B of the per-signature table layout (p256_pair29.h FABGPU_QTAB_SIG_MAJOR) inside ONE gpurun call, plus the rocprofv3 passes of the
# bench command: kernel trace + stats, and the FETCH_SIZE / WRITE_SIZE counters in separate passes (MI355X_MICROARCH.md).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in new old new old; do
  if [ $v = old ]; then export FABGPU_LIB_PATH=$R/fabric-mod_amd/lib/libfabgpu_oldlayout.so; else unset FABGPU_LIB_PATH; fi
  python $R/bench.py --steps 50 --warmup 5 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['dispersion'])"
done
for v in new old; do
  if [ $v = old ]; then export FABGPU_LIB_PATH=$R/fabric-mod_amd/lib/libfabgpu_oldlayout.so; else unset FABGPU_LIB_PATH; fi
  rm -rf /tmp/prof_$v /tmp/pmcF_$v /tmp/pmcW_$v
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -- python $R/bench.py --steps 20 --warmup 5 --no-extras > $OUT/r02_bench_under_rocprof_$v.json 2>/dev/null
  rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcF_$v -- python $R/bench.py --steps 10 --warmup 2 --no-extras > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d /tmp/pmcW_$v -- python $R/bench.py --steps 10 --warmup 2 --no-extras > /dev/null 2>&1
  python $R/profiles/summarize_rocprof.py $(find /tmp/prof_$v /tmp/pmcF_$v /tmp/pmcW_$v -name "*.db") > $OUT/r02_rocprof_layout_$v.txt 2>&1
done
unset FABGPU_LIB_PATH
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmcS -- python $R/bench.py --steps 10 --warmup 2 --no-extras > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d /tmp/pmcT -- python $R/bench.py --steps 10 --warmup 2 --no-extras > /dev/null 2>&1
python $R/profiles/summarize_rocprof.py $(find /tmp/pmcS /tmp/pmcT -name "*.db") > $OUT/r02_rocprof_pmc_sq.txt 2>&1
head -c 1500 $OUT/r02_rocprof_layout_new.txt; echo; head -c 1500 $OUT/r02_rocprof_layout_old.txt; echo; head -c 2500 $OUT/r02_rocprof_pmc_sq.txt
