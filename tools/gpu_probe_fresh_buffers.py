#!/usr/bin/env python3
"""Does it matter to the pass whether a block arrives in memory the runtime has seen before?  The same friendly block, re-sent from
the same buffer and from a fresh copy each time (what a peer's blocks are), with the pass's own stage timing (FABGPU_PASS_TIMING=1)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fabric-mod_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np  # noqa: E402

import fabgpu  # noqa: E402

blk = open(os.path.join(ROOT, ".bench_blocks", "friendly_10000.bin"), "rb").read()
csp = fabgpu.GPUCSP(device=0)
for _ in range(3):
    fabgpu.preverify_block2(csp, blk, lean=True)


def run(tag, blocks):
    per = []
    for b in blocks:
        t0 = time.perf_counter()
        r = fabgpu.preverify_block2(csp, b, lean=True)
        per.append((time.perf_counter() - t0) * 1e3)
        assert (r["tx_flags"] == 0).all()
    print(tag, " ".join("%.2f" % x for x in per), file=sys.stderr)


run("same buffer     ", [blk] * 8)
fresh = [bytes(bytearray(blk)) for _ in range(8)]
run("fresh copies    ", fresh)
run("the copies again", fresh)
arr = [np.frombuffer(blk, dtype=np.uint8).copy() for _ in range(4)]
run("numpy copies    ", [a.tobytes() for a in arr])
csp.close()
