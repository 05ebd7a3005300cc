#!/usr/bin/env python3
"""Digest-memo soak under churn: for --seconds, one provider, an ARRIVAL thread that pre-verifies blocks (memo seeding, a bounded pool of
kept block copies, random route: device-built table, host-seeded table, host walk) and evicts the oldest, and V VALIDATOR threads that ask
bccsp.Hash / bccsp.Verify questions about messages of blocks that are waiting, being evicted or long gone - plus mutated messages that
are in no block.  What must hold whatever interleaving the threads hit (fabgpu_csp_hash_lookup's contract, include/fabgpu_bccsp.h):
  * a hit is SHA-256 of exactly the bytes that were asked about (checked with hashlib: never a digest of other bytes),
  * a mutated message never hits,
  * the verdict memo answers with the status the pass reported, or misses.
Test infrastructure: tests/test_soak.py runs it for a few seconds under `pytest -m gpu`; `python tests/soak_hash_memo.py --seconds 120` is
the long run recorded under profiles/."""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
for p in ("fabric-mod_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def soak(seconds, seed, validators=6):
    import numpy as np

    import blockgen
    import fabgpu
    rng = np.random.default_rng(seed)
    # a handful of distinct blocks (signing is pure Python: made once), sizes on both sides of the 4 MiB threaded-upload threshold
    blocks = [blockgen.endorser_block(n_tx, 100 + k)[0] for k, n_tx in enumerate((40, 120, 300, 900))]
    csp = fabgpu.GPUCSP(device=0, devices=[0], concurrent_passes=2, hash_memo_blocks=3)
    stats = {"passes": 0, "evictions": 0, "hash_questions": 0, "hash_hits": 0, "hash_hits_checked": 0, "mutants_asked": 0, "verify_questions": 0,
             "verify_hits": 0, "blocks_without_copy": 0}
    lock = threading.Lock()
    live = {}          # block_seq -> list of (msg, qxy, sig, status) of tuples the device hashed and decided
    recent = []        # the same lists of blocks that were evicted a moment ago (their questions must miss, or hit ANOTHER live copy of the same bytes - correctly)
    errors = []
    stop = threading.Event()
    for b in blocks:
        fabgpu.preverify_block(csp, b)                                        # identities learned

    def arrivals():
        r = np.random.default_rng(seed + 1)
        seq = 1
        order = []
        try:
            while not stop.is_set():
                blk = blocks[int(r.integers(0, len(blocks)))]
                route = int(r.integers(0, 3))
                opt = [None, ("pass_device_memo", -1), ("pass_device_walk", -1)][route]
                prev = csp.set_option(opt[0], opt[1]) if opt else None
                out = fabgpu.preverify_block2(csp, blk, block_seq=seq, seed_memo=True)
                if opt:
                    csp.set_option(opt[0], prev)
                tuples = []
                for i in range(len(out["tuple_status"])):
                    if not out["tuple_hashed"][i] or out["tuple_status"][i] > 3:
                        continue
                    sp = [int(x) for x in out["tuple_spans"][i]]
                    msg = out["arena"][sp[2]:sp[2] + sp[3]] + out["arena"][sp[4]:sp[4] + sp[5]]
                    tuples.append((msg, bytes(out["tuple_qxy"][i]), out["arena"][sp[6]:sp[6] + sp[7]], int(out["tuple_status"][i])))
                with lock:
                    live[seq] = tuples
                    order.append(seq)
                    stats["passes"] += 1
                seq += 1
                while len(order) > int(r.integers(1, 5)):                        # up to four blocks wait for their validators: more than the pool of three copies
                    old = order.pop(0)
                    with lock:
                        gone = live.pop(old)
                        recent.append(gone)
                        del recent[:-2]
                        stats["evictions"] += 1
                    fabgpu.memo_evict_block(csp, old)
                time.sleep(float(r.uniform(0, 0.003)))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
            stop.set()

    def validator(k):
        r = np.random.default_rng(seed + 10 + k)
        try:
            while not stop.is_set():
                with lock:
                    pools = list(live.values()) + list(recent)
                if not pools:
                    time.sleep(0.001)
                    continue
                tuples = pools[int(r.integers(0, len(pools)))]
                if not tuples:
                    continue
                start = int(r.integers(0, len(tuples)))
                for msg, qxy, sig, status in tuples[start:start + int(r.integers(1, 9))]:   # a transaction's worth in a row: the per-thread hint's case
                    if r.random() < 0.25:                                        # a message that is in no block
                        m = bytearray(msg)
                        m[int(r.integers(0, len(m)))] ^= 1 << int(r.integers(0, 8))
                        if fabgpu.hash_lookup(csp, bytes(m)) is not None:
                            raise AssertionError("a mutated message was answered")
                        with lock:
                            stats["mutants_asked"] += 1
                        continue
                    d = fabgpu.hash_lookup(csp, msg)
                    want = hashlib.sha256(msg).digest()
                    if d is not None and d != want:
                        raise AssertionError("the digest memo answered with a digest of other bytes")
                    st = fabgpu.memo_lookup(csp, qxy[:32], qxy[32:], sig, want)
                    if st is not None and st != status:
                        raise AssertionError("the verdict memo answered %d for a tuple whose pass said %d" % (st, status))
                    with lock:
                        stats["hash_questions"] += 1
                        stats["hash_hits"] += d is not None
                        stats["hash_hits_checked"] += d is not None
                        stats["verify_questions"] += 1
                        stats["verify_hits"] += st is not None
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
            stop.set()

    th = [threading.Thread(target=arrivals)] + [threading.Thread(target=validator, args=(k,)) for k in range(validators)]
    for t in th:
        t.start()
    t_end = time.time() + seconds
    while time.time() < t_end and not stop.is_set():
        time.sleep(0.05)
    stop.set()
    for t in th:
        t.join(timeout=60)
    hm = fabgpu.hash_memo_stats(csp)
    stats["blocks_without_copy"] = hm["refused"]
    stats["library_hash_hits"], stats["library_hash_misses"] = hm["hits"], hm["misses"]
    csp.close()
    if errors:
        return {"soak": "MISMATCH", "errors": errors[:3], **stats}
    return {"soak": "ok", "seconds": seconds, "validators": validators, **{k: int(v) for k, v in stats.items()}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--validators", type=int, default=6)
    a = ap.parse_args()
    res = soak(a.seconds, a.seed, a.validators)
    print(json.dumps(res))
    sys.exit(0 if res["soak"] == "ok" else 1)
