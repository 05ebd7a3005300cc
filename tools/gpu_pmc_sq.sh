#!/bin/bash
# SQ counters of the kernels bench.py's rooflines are about (the round-2 table described a kernel that is no longer
# the default).  rocprofv3 --pmc in passes of <= 8 SQ counters (MI355X_MICROARCH.md "rocprofv3 PMC slots"), --kernel-trace only, over
# tools/gpu_pmc_kernels.py; summarised by profiles/summarize_rocprof.py into gpurun_out/${TAG}_pmc_sq.txt (copy it to profiles/).
exec </dev/null
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04}
OUT=$R/gpurun_out/${TAG}_pmc_sq.txt
: > $OUT
KERNELS=${KERNELS:-"verify verify_global fused nym"}
pass() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  ( cd /tmp && timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$name -- python $R/tools/gpu_pmc_kernels.py $KERNELS > /tmp/pmc_$name.log 2>&1 )
  f=$(find /tmp/pmc_$name -name "*.db" 2>/dev/null | head -1)
  echo "#### pass $name: $*" >> $OUT
  if [ -n "$f" ]; then python $R/profiles/summarize_rocprof.py "$f" 2>&1 | grep -v "at::native\|rocclr\|elementwise" >> $OUT; else echo "no db ($(tail -2 /tmp/pmc_$name.log | tr '\n' ' '))" >> $OUT; fi
}
pass issue SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
grep -c "mean=" $OUT
tail -5 $OUT | cut -c1-200
