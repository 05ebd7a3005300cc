#!/bin/bash
# block pass with several callers on one provider (channels of a peer): parity tests + aggregate rate -> profiles/r02_block_pass_concurrent.jsonl
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_block_prepass.py tests/test_ledger_goldens.py -m gpu -x -q 2>&1 | tail -5
for t in 2 3; do
  timeout 300 python tools/bench_block.py --tx 10000 --steps 12 --threads $t 2>&1 | tail -1
  timeout 300 python tools/bench_block.py --tx 10000 --steps 12 --threads $t --memo 2>&1 | tail -1
done
timeout 300 python tools/bench_block.py --tx 1000 --steps 30 --threads 2 2>&1 | tail -1
} > gpurun_out/conc_probe.log 2>&1
