#!/usr/bin/env python3
"""Randomised soak of the idemix pseudonym-signature call (round 5: three launches - commitments, fixed-base terms on a side stream with a
ready flag per wavefront and an inline fallback, challenges on eight lanes per message): for --seconds, draw batch sizes across the
four-lane range and its neighbours, replicate a base of oracle-signed signatures (valid, tampered, out-of-domain ...: make_batch) with
ragged message lengths, and compare every status byte and verdict bit with the oracle's - thousands of calls through ONE context, so that a
race between the side launch and the commitment launch (stale flags, a record read before it was written, a workspace reused too early)
would show as a wrong status sooner or later.  Test infrastructure; `python tests/soak_idemix.py --seconds 120`."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fabric-mod_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def soak(seconds, seed, flags=0):
    import numpy as np

    import fabgpu
    from idemix_common import be32, fixtures, make_batch
    fx = fixtures()
    ctx = fabgpu.Context(device=0, flags=flags)
    issuers = []
    for name in ("MSP1OU1", "MSP2OU1"):
        ipk = fx[name]["ipk"]
        ctx.idemix_issuer_register((be32(ipk.h_sk[0]), be32(ipk.h_sk[1])), (be32(ipk.h_rand[0]), be32(ipk.h_rand[1])), ipk.hash)
        issuers.append((ipk, fx[name]["signer"].sk))
    base_n = 240
    base = make_batch(issuers, base_n, seed)
    arena, off, iid, cols, expect = base.arrays()
    rng = np.random.default_rng(seed)
    stats = {"calls": 0, "signatures": 0, "invalid": 0, "sizes": {}}
    t_end = time.time() + seconds
    while time.time() < t_end:
        n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 2500), rng.integers(2500, 16385), rng.integers(16385, 40000)], p=[0.35, 0.45, 0.17, 0.03]))
        pick = rng.integers(0, base_n, size=n)
        lens = (off[1:] - off[:-1])[pick]
        off2 = np.zeros(n + 1, dtype=np.uint32)
        off2[1:] = np.cumsum(lens)
        arena2 = np.concatenate([arena[off[i]:off[i + 1]] for i in pick]) if n else np.zeros(0, np.uint8)
        reps = 1 if n > 2500 else int(rng.integers(1, 6))          # small calls back to back: the same workspace, flags re-zeroed each time
        for _ in range(reps):
            ok, st = ctx.idemix_nym_verify_batch(arena2, off2, *[c[pick] for c in cols], issuer_id=iid[pick])
            bad = np.nonzero(st != expect[pick])[0]
            assert bad.size == 0, "n=%d: rows %r got %r want %r" % (n, bad[:6].tolist(), st[bad[:6]].tolist(), expect[pick][bad[:6]].tolist())
            assert np.array_equal(ok, expect[pick] == 0), n
            stats["calls"] += 1
            stats["signatures"] += n
            stats["invalid"] += int((expect[pick] != 0).sum())
        k = "1-69" if n < 70 else ("70-2499" if n < 2500 else ("2500-16384" if n <= 16384 else "16385+"))
        stats["sizes"][k] = stats["sizes"].get(k, 0) + reps
    ctx.close()
    return stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    s = soak(a.seconds, a.seed)
    print(json.dumps({"soak": "idemix nym verify, three launches", "seconds": a.seconds, **s, "parity": "every status byte and verdict bit equal to the oracle's"}))


if __name__ == "__main__":
    main()
