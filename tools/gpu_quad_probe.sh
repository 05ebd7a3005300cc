#!/bin/bash
# idemix: four lanes per signature against two (FABGPU_FLAG_NO_QUAD), BASELINE config 5 and idemix-only batches -> profiles/r02_idemix_quad.jsonl
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
: > gpurun_out/quad_probe.jsonl
for args in "" "--no-quad" "--n 3000 --idemix-share 1.0" "--n 3000 --idemix-share 1.0 --no-quad" "--n 12000 --idemix-share 1.0" "--n 12000 --idemix-share 1.0 --no-quad"; do
  timeout 300 python tools/bench_cfg5_mixed.py $args 2>>gpurun_out/quad_probe.err | tail -1 >> gpurun_out/quad_probe.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/quad_probe.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:70], "| mixed %.3f ms = %.1f M/s | idemix alone %.3f ms (kernel %.3f) | ecdsa alone %.3f" % (
        d["ms_per_step"], d["value"] / 1e6, d["idemix_alone"]["ms_per_step"], d["idemix_alone"]["kernel_ms"], d["ecdsa_alone"]["ms_per_step"]))
PY
