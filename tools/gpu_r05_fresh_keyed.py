#!/usr/bin/env python3
"""round 5 probe: the first passes of a fresh provider over the friendly 10 000-tx block - per pass: wall ms, tuples through key tables,
certificates decoded on the device, identities learned, stage breakdown."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))
import fabgpu
blk = open(os.path.join(ROOT, ".bench_blocks", "friendly_10000.bin"), "rb").read()
csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=2, expect_block_bytes=len(blk) + 4096, expect_tuples=40064)
for k in range(6):
    b = bytes(bytearray(blk))
    t0 = time.perf_counter()
    r = fabgpu.preverify_block2(csp, b, block_seq=100 + k, seed_memo=True, lean=True)
    ms = (time.perf_counter() - t0) * 1e3
    print("pass %d: %.2f ms  n_keyed %d of %d  device-decoded %d  stages %s  learned %s" % (k + 1, ms, r["n_keyed"], r["n_tuples"], r["n_device_decoded"],
          ["%.2f" % x for x in r["ms_stage"]], getattr(fabgpu, "pass_routes")(csp)))
    fabgpu.memo_evict_block(csp, 100 + k)
csp.close()
