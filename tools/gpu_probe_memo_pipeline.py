"""two callers, memo seeding + eviction per block (what the Go arrival hook runs): wall per block and every pass's own time"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd")]
import fabgpu   # noqa: E402
blk = open(os.path.join(ROOT, ".bench_blocks", "friendly_10000.bin"), "rb").read()
CP = int(os.environ.get("PROBE_CP", "0"))                 # GPUOpts.ConcurrentPasses: 0 = allocate when passes first overlap (round 3's behaviour)
csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=CP, expect_block_bytes=len(blk) + (1 << 20), expect_tuples=40256) if CP else fabgpu.GPUCSP(device=0)
for _ in range(3):
    fabgpu.preverify_block2(csp, blk, lean=True)
if len(sys.argv) > 3 and sys.argv[3] == "hostfirst":      # as bench.py's legs come: memo tables built on the HOST first (they end up in the free list)
    csp.set_option("pass_stage_min_bytes", 1 << 40)
    for k in range(10):
        fabgpu.preverify_block2(csp, bytes(bytearray(blk)), block_seq=50 + k, seed_memo=True, lean=True)
        fabgpu.memo_evict_block(csp, 50 + k)
    csp.set_option("pass_stage_min_bytes", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
memo = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
copies = [[bytes(bytearray(blk)) for _ in range(N)] for _ in range(2)]
lat = [[], []]


def caller(t):
    for k in range(N):
        c0 = time.perf_counter()
        r = fabgpu.preverify_block2(csp, copies[t][k], block_seq=1000 * (t + 1) + k, seed_memo=memo, lean=True)
        c1 = time.perf_counter()
        if memo:
            fabgpu.memo_evict_block(csp, 1000 * (t + 1) + k)
        lat[t].append(((c1 - c0) * 1e3, (time.perf_counter() - c1) * 1e3, r["ms_stage"]))


for rnd in range(int(os.environ.get("ROUNDS", "1"))):          # ROUNDS=2: new threads on a warm provider
    lat = [[], []]
    th = [threading.Thread(target=caller, args=(t,)) for t in range(2)]
    c0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - c0
    print("memo" if memo else "flags", "wall per block %.3f ms" % (wall / (2 * N) * 1e3))
    for t in range(2):
        print(" caller %d pass ms:" % t, " ".join("%.1f" % a[0] for a in lat[t]))
        print(" caller %d evict ms:" % t, " ".join("%.2f" % a[1] for a in lat[t]))
        print(" caller %d stages (outline | wait upload | device | post):" % t, " ".join("%.1f|%.1f|%.1f|%.1f" % tuple(a[2]) for a in lat[t][:10]))
csp.close()
