import sys, os, time, threading
sys.path[:0] = ["/root/repo/fabric-mod_amd", "/root/repo/tests", "/root/repo/oracle"]
import numpy as np
import fabgpu, blockgen
blk, _ = blockgen.endorser_block(1500, 21)
for hm in (0, -1):
    csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=2, expect_block_bytes=len(blk) + 4096, expect_tuples=6200, pass_hash_memo=hm)
    for k in range(6):
        fabgpu.preverify_block2(csp, blk, block_seq=k, lean=True)
    per = [[], []]
    def caller(t):
        for k in range(6):
            b = bytes(bytearray(blk))
            c0 = time.perf_counter()
            r = fabgpu.preverify_block2(csp, b, block_seq=100 * (t + 1) + k, seed_memo=True, lean=True)
            per[t].append(((time.perf_counter() - c0) * 1e3, [round(x, 2) for x in r["ms_stage"]]))
            fabgpu.memo_evict_block(csp, 100 * (t + 1) + k)
    th = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    print("hash_memo", hm)
    for t in range(2):
        print("  ", [(round(a, 2), s) for a, s in per[t]])
    csp.close()
