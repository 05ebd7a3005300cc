#!/usr/bin/env python3
"""round 5 probe: why do 30 000-tuple launches with an event pair around each take 0.66 ms and the same launches queued bare 0.71?
Wall clock per step of K back-to-back fabgpu_p256_verify_batch_dev calls on one stream: bare; with FABGPU_FLAG_TIME_KERNELS (the library
records two timing events per launch); with a torch timing event recorded by the caller after every step; with a non-timing event."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fabric-mod_amd"))
import numpy as np, torch, fabgpu
n = 30000
b = fabgpu.synth_batch(n, seed=20260921, invalid_permille=10)
d = {k: torch.from_numpy(b[k]).cuda() for k in ("qx", "qy", "e", "r", "s")}
words = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
def run(ctx, after=None, reps=5):
    def step():
        ctx.p256_verify_batch_dev(n, d["qx"].data_ptr(), d["qy"].data_ptr(), d["e"].data_ptr(), d["r"].data_ptr(), d["s"].data_ptr(), words.data_ptr(), 0, st.cuda_stream)
        if after: after()
    out = []
    for _ in range(reps):
        for _ in range(5): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / K * 1e3)
    return "%.4f ms/step (min %.4f)" % (statistics.median(out), min(out))
plain = fabgpu.Context(device=0, max_batch=n)
timed = fabgpu.Context(device=0, max_batch=n, flags=fabgpu.FLAG_TIME_KERNELS)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(64)]
evn = [torch.cuda.Event(enable_timing=False) for _ in range(64)]
k = [0]
def rec_t():
    evs[k[0] % 64].record(st); k[0] += 1
def rec_n():
    evn[k[0] % 64].record(st); k[0] += 1
for name, ctx, after in (("bare", plain, None), ("library timing events", timed, None), ("caller timing event per step", plain, rec_t), ("caller non-timing event per step", plain, rec_n), ("bare again", plain, None)):
    print("%-36s %s" % (name, run(ctx, after)))
assert (fabgpu.unpack_bits(words.cpu().numpy().view(np.uint64), n) == (b["kind"] == 0)).all()
# ---- clock ramp: per-launch durations right after the GPU sat idle (what bench.py's 5 warm-up steps + 20 timed steps see) ----
for idle in (0.0, 0.2, 2.0):
    torch.cuda.synchronize(); time.sleep(idle)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, bb in ev:
        a.record(st)
        plain.p256_verify_batch_dev(n, d["qx"].data_ptr(), d["qy"].data_ptr(), d["e"].data_ptr(), d["r"].data_ptr(), d["s"].data_ptr(), words.data_ptr(), 0, st.cuda_stream)
        bb.record(st)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(bb) for a, bb in ev]
    print("after %.1f s idle: launches 1-5 %s | 6-25 mean %.4f | 26-40 mean %.4f" % (idle, " ".join("%.3f" % x for x in ms[:5]), sum(ms[5:25]) / 20, sum(ms[25:]) / 15))
