#!/usr/bin/env python3
"""One-signature bccsp.Verify calls from T threads through the coalescer (fabgpu_csp_verify_coalesced) against one launch per call
(fabgpu_csp_verify): calls per second and launches used.  A C driver would push harder than Python threads can (the GIL is released
only inside the call); the point here is the shape - batch size grows with the number of callers in flight, the rate with it."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 8, 32, 128])
    ap.add_argument("--calls", type=int, default=4000, help="calls per measurement, spread over the threads")
    args = ap.parse_args()
    import numpy as np

    import bccsp_sw_oracle as po
    import fabgpu
    rng = np.random.default_rng(3)
    b = fabgpu.synth_batch(2048, seed=9, invalid_permille=100)
    keys = [fabgpu.ECDSAPublicKey(int.from_bytes(bytes(b["qx"][i]), "big"), int.from_bytes(bytes(b["qy"][i]), "big")) for i in range(2048)]
    sigs = [po.marshal_ecdsa_signature(int.from_bytes(bytes(b["r"][i]), "big"), int.from_bytes(bytes(b["s"][i]), "big")) for i in range(2048)]
    digs = [bytes(b["e"][i]) for i in range(2048)]
    csp = fabgpu.GPUCSP(device=0)
    want = [v for v, _ in csp.verify_batch(keys, sigs, digs)]
    rows = []
    t0 = time.perf_counter()
    for i in range(200):
        try:
            assert csp.verify(keys[i], sigs[i], digs[i]) == want[i]
        except fabgpu.BCCSPError:
            assert not want[i]
    per_call = (time.perf_counter() - t0) / 200
    rows.append({"mode": "fabgpu_csp_verify, one launch per call, 1 thread", "calls_per_s": 1 / per_call, "us_per_call": per_call * 1e6})
    for T in args.threads:
        bad = []
        s0 = csp.coalescer_stats()

        def worker(w):
            for i in range(w, args.calls, T):
                j = i % 2048
                try:
                    ok = csp.verify_coalesced(keys[j], sigs[j], digs[j])
                except fabgpu.BCCSPError:
                    ok = False
                if ok != want[j]:
                    bad.append(j)
        th = [threading.Thread(target=worker, args=(w,)) for w in range(T)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        s1 = csp.coalescer_stats()
        assert not bad
        rows.append({"mode": "fabgpu_csp_verify_coalesced", "threads": T, "calls_per_s": args.calls / dt, "launches": s1["launches"] - s0["launches"],
                     "calls": s1["calls"] - s0["calls"], "mean_batch": (s1["calls"] - s0["calls"]) / max(1, s1["launches"] - s0["launches"])})
    print(json.dumps({"metric": "one-signature Verify calls per second (many callers, one provider)", "rows": rows,
                      "note": "Python threads: the caller side is GIL-bound; parity with the batch entry point asserted on every call"}))
    csp.close()


if __name__ == "__main__":
    main()
