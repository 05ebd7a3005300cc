#!/usr/bin/env python3
"""a few device-route passes over the 10 000-transaction friendly block, for a rocprofv3 --kernel-trace timeline"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd")]
import fabgpu   # noqa: E402
blk = open(os.path.join(ROOT, ".bench_blocks", sys.argv[1] if len(sys.argv) > 1 else "friendly_10000.bin"), "rb").read()
csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=1, expect_block_bytes=len(blk) + (1 << 20), expect_tuples=40256)
for k in range(8):
    r = fabgpu.preverify_block2(csp, bytes(bytearray(blk)), block_seq=k, lean=True)
print(r["ms_stage"], r["n_keyed"])
csp.close()
