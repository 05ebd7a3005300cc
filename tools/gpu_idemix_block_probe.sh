#!/bin/bash
# block pass over a 10 000-tx block whose every 5th creator is an idemix identity (.bench_blocks/, made by tools/make_bench_blocks.py)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
timeout 300 python tools/bench_block.py --timing --block-file .bench_blocks/idemix_10000_5.bin --idemix --steps 8 2>&1 | grep -E "^fabgpu pass|^\{" | tail -4
timeout 300 python tools/bench_block.py --block-file .bench_blocks/idemix_10000_5.bin --idemix --steps 8 --memo 2>&1 | tail -1
} > gpurun_out/idemix_block.log 2>&1
