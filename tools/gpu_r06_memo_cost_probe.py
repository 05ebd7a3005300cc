#!/usr/bin/env python3
"""round 6: what memo seeding (verdict memo + digest memo) adds to a lone pass over the 10 000-transaction block, stage by stage
(fabgpu_block_pass.ms_stage: outline + identity table, wait for the upload, device phase, bookkeeping), flags only against
FABGPU_PASS_SEED_MEMO, with the digest memo on and off.  Prints medians over 12 passes each, fresh copy of the block per pass."""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import fabgpu   # noqa: E402

blk = open(os.path.join(ROOT, ".bench_blocks", "friendly_10000.bin"), "rb").read()
for hm in (0, -1):
    csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=2, expect_block_bytes=len(blk) + 4096, expect_tuples=40064, pass_hash_memo=hm)
    for k in range(4):
        fabgpu.preverify_block2(csp, blk, block_seq=k, lean=True)
    for memo in (False, True, False, True):
        wall, stages = [], []
        for k in range(12):
            b = bytes(bytearray(blk))
            c0 = time.perf_counter()
            r = fabgpu.preverify_block2(csp, b, block_seq=1000 + k, seed_memo=memo, lean=True)
            wall.append((time.perf_counter() - c0) * 1e3)
            stages.append(r["ms_stage"])
            if memo:
                fabgpu.memo_evict_block(csp, 1000 + k)
        med = [round(statistics.median(s[i] for s in stages), 3) for i in range(4)]
        print("digest memo %s, seed_memo %-5s: wall %.3f ms  stages [outline+idtab, upload wait, device, post] = %s" % ("on " if hm == 0 else "off", memo, statistics.median(wall), med))
    csp.close()
