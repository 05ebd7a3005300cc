cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=.bench_blocks/ecdsa_10000_0.bin
for s in 0 1 2 3 4 5 7 8 9 11 0; do
FABGPU_WALK_SCHED=$s FABGPU_PASS_TIMING=1 timeout 100 python tools/bench_block.py --block-file $B --steps 24 > gpurun_out/sched_$s.json 2> gpurun_out/sched_$s.err
python - <<PY
import json,re,statistics
d=json.loads(open("gpurun_out/sched_$s.json").read().strip().splitlines()[-1])
dev=[float(m.group(1)) for m in re.finditer(r"device ([0-9.]+)\)", open("gpurun_out/sched_$s.err").read())]
print("sched $s: ms/block median %.3f min %.3f | device phase median %.3f min %.3f" % (d["ms_per_block"], d["ms_min"], statistics.median(dev[4:]), min(dev[4:])))
PY
done
