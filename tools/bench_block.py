#!/usr/bin/env python3
"""Exploration bench for BASELINE.json's second metric, "validated tx/sec per block", through the block-level pre-verify pass
(fabgpu_csp_block_preverify): a marshalled common.Block of N endorser transactions x 3 endorsements (1024-byte
proposal-response payload, PEM x509 identities, 4 endorsers, 2 creators) goes in as bytes, per-transaction flags come out.
Timed per call: C++ block walk + identity cache + DER/low-S gates + staging + H2D of the block + mid-state and fused keyed
kernels + D2H.  Not the driver's bench.  The oracle is used only to sign the synthetic block."""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("fabric-mod_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tx", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import numpy as np

    import bccsp_sw_oracle as po
    import blockbuilder as bb
    import coracle
    import fabgpu
    ids = [i for i in json.load(open(os.path.join(ROOT, "tests", "golden", "block_identities.json")))["identities"] if i["curve"] == "prime256v1"]
    sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in ids]
    L = coracle.lib()
    rng = np.random.default_rng(1)

    def sign(k, msg):
        d = int(ids[k]["d"], 16).to_bytes(32, "big")
        e = hashlib.sha256(msg).digest()
        nonce = bytes(rng.integers(1, 255, size=32, dtype=np.uint8))
        nonce = b"\x00" + nonce[1:]
        r, s = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        assert L.oracle_p256_sign(d, e, nonce, 1, r, s) == 0
        return po.marshal_ecdsa_signature(int.from_bytes(r.raw, "big"), int.from_bytes(s.raw, "big"))
    envs = []
    for t in range(args.tx):
        picks = [int(j) for j in rng.choice(4, size=3, replace=False)]
        c = 4 + t % 2
        # TxID and proposal hash as the validators recompute them (the pass checks both: SURVEY 8(a) a12)
        payload, _ = bb.consistent_endorser_tx("mychannel", sid[c], bytes(rng.integers(0, 256, size=24, dtype=np.uint8)),
                                               bytes(rng.integers(0, 256, size=300, dtype=np.uint8)), bytes(rng.integers(0, 256, size=990, dtype=np.uint8)),
                                               lambda prp: [(sid[j], sign(j, prp + sid[j])) for j in picks])
        envs.append(bb.envelope(payload, sign(c, payload)))
    blk = bb.block(1, envs)
    csp = fabgpu.GPUCSP(device=0)
    out = fabgpu.preverify_block(csp, blk)
    assert (out["tx_flags"] == 0).all() and len(out["tuple_status"]) == 4 * args.tx
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fabgpu.preverify_block(csp, blk)
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"metric": "validated tx/sec per block (block-level pre-verify pass, marshalled block in, flags out)", "value": args.tx / dt,
                      "unit": "tx/s", "ms_per_block": dt * 1e3, "signatures_per_s": 4 * args.tx / dt,
                      "config": {"workload": "%d endorser tx x (1 creator + 3 endorsement signatures), %.1f MB block, 6 registered identities" % (
                          args.tx, len(blk) / 1e6)}, "checks": "creator signature, 3 endorsement signatures, TxID and proposal hash per transaction",
                      "parity": "every transaction flagged valid; corrupted blocks are covered by tests/test_block_prepass.py"}))
    csp.close()


if __name__ == "__main__":
    main()
