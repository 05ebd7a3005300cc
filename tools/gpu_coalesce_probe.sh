#!/bin/bash
# one-signature Verify calls from many native threads through the coalescer -> profiles/r02_coalescer.jsonl
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
gcc -O2 -std=gnu99 -Iinclude tools/coalesce_harness.c -Lfabric-mod_amd/lib -lfabgpu -lpthread -o /tmp/coalesce || exit 1
export LD_LIBRARY_PATH=$PWD/fabric-mod_amd/lib:$LD_LIBRARY_PATH
: > gpurun_out/coalesce_native.jsonl
for t in 1 16 64 256 1024 4096; do
  c=$(( t < 64 ? 300 : (t < 1024 ? 200 : 60) ))
  timeout 120 /tmp/coalesce $t $c >> gpurun_out/coalesce_native.jsonl 2>> gpurun_out/coalesce_native.err
done
cat gpurun_out/coalesce_native.jsonl
