"""The first passes of a FRESH provider over the friendly 10 000-transaction block, with the library's own stage times (provider option
pass_timing): what a peer's first blocks on a channel cost - code objects, certificates, comb tables of the signers."""
import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd")]
import fabgpu
blk = open(os.path.join(ROOT, ".bench_blocks", "friendly_10000.bin"), "rb").read()
t0 = time.perf_counter()
csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=2, expect_block_bytes=len(blk) + (1 << 20), expect_tuples=40256, pass_timing=1)
print("provider %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
for k in range(4):
    fresh = bytes(bytearray(blk))                 # (a fresh buffer per pass, copied outside the timed call)
    t0 = time.perf_counter()
    r = fabgpu.preverify_block2(csp, fresh, block_seq=k, seed_memo=True, lean=True)
    print("pass %d: %.2f ms  stages %s keyed %d" % (k, (time.perf_counter() - t0) * 1e3, [round(x, 2) for x in r["ms_stage"]], r["n_keyed"]), flush=True)
    fabgpu.memo_evict_block(csp, k)
csp.close()
