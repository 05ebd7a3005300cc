#!/usr/bin/env python3
"""The coalescer with a FAKE device (a launch = a sleep) - libfabgpu_hosttest.so, no GPU: what the host alone can sustain with T callers."""
import ctypes, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "fabric-mod_amd", "lib", "libfabgpu_hosttest.so"))
f = L.hosttest_coalescer
f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int] + [ctypes.POINTER(ctypes.c_uint64)] * 3
a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
for T, C in [(16, 300), (64, 200), (256, 200), (1024, 60), (4096, 60)]:
    t = time.time()
    rc = f(T, C, 700, 50, 32768, 0, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    dt = time.time() - t
    print(json.dumps({"threads": T, "calls": T * C, "fake_launch_us": 700, "rc": rc, "launches": a.value, "largest_batch": b.value, "calls_per_s": round(T * C / dt)}))
