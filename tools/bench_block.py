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
    ap.add_argument("--memo", action="store_true", help="fabgpu_csp_block_preverify2 with FABGPU_PASS_SEED_MEMO (digests back + memo seeding) and eviction per block")
    ap.add_argument("--idle-ms", type=float, default=0.0, help="sleep this long between blocks (a peer sees a block every few hundred ms: the GPU clocks down)")
    ap.add_argument("--threads", type=int, default=1, help="callers validating blocks at once through the one provider (channels of a peer): aggregate rate")
    ap.add_argument("--block-file", help="a marshalled block made by tools/make_bench_blocks.py (all signatures valid) instead of building one here")
    ap.add_argument("--idemix", action="store_true", help="register the fixtures' IdemixMSP1 first (blocks of make_bench_blocks.py idemix ...)")
    ap.add_argument("--tables", type=int, default=256, help="device comb tables the identity cache may build (6 signers: fewer than 6 leaves newcomers on the fresh-key path)")
    ap.add_argument("--tables16", action="store_true", help="FABGPU_FLAG_KEY_TABLES_16BIT on every context: 16-bit comb tables for the registered keys (round 6)")
    ap.add_argument("--host-walk", action="store_true", help="provider option pass_device_walk off: the block is walked on the host (the round-2 route)")
    ap.add_argument("--timing", action="store_true", help="provider option pass_timing: per-stage times of every pass on stderr")
    ap.add_argument("--register-after", type=int, default=1, help="an identity earns its comb table after being named this often (the provider's default: 64)")
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
    for t in range(args.tx if not args.block_file else 0):
        picks = [int(j) for j in rng.choice(4, size=3, replace=False)]
        c = 4 + t % 2
        # TxID and proposal hash as the validators recompute them (the pass checks both: SURVEY 8(a) a12)
        payload, _ = bb.consistent_endorser_tx("mychannel", sid[c], bytes(rng.integers(0, 256, size=24, dtype=np.uint8)),
                                               bytes(rng.integers(0, 256, size=300, dtype=np.uint8)), bytes(rng.integers(0, 256, size=990, dtype=np.uint8)),
                                               lambda prp: [(sid[j], sign(j, prp + sid[j])) for j in picks])
        envs.append(bb.envelope(payload, sign(c, payload)))
    blk = bb.block(1, envs) if not args.block_file else open(args.block_file, "rb").read()
    switches = {}
    if args.host_walk:
        switches["pass_device_walk"] = -1
    if args.timing:
        switches["pass_timing"] = 1
    csp = fabgpu.GPUCSP(devices=[0], flags=fabgpu.FLAG_KEY_TABLES_16BIT if args.tables16 else 0, **switches)
    if args.idemix:
        raw_ipk = bytes.fromhex(json.load(open(os.path.join(ROOT, "tests", "golden", "idemix_fixtures.json")))["msps"]["MSP1OU1"]["ipk"])
        assert csp.idemix_msp_register("IdemixMSP1", raw_ipk) >= 0
    if args.block_file:
        args.tx = fabgpu.block_parse(blk)["n_tx"]
    csp._L.fabgpu_csp_identity_cache_limits(csp._h, 4096, args.tables, args.register_after)
    out = fabgpu.preverify_block(csp, blk)
    n_keyed = fabgpu.preverify_block2(csp, blk, lean=True)["n_keyed"]
    assert (out["tx_flags"] == 0).all() and len(out["tuple_status"]) == 4 * args.tx
    import statistics
    per = []
    hits = None
    for k in range(args.steps):
        if args.idle_ms:
            time.sleep(args.idle_ms * 1e-3)
        t0 = time.perf_counter()
        if args.memo:
            out = fabgpu.preverify_block2(csp, blk, block_seq=100 + k, seed_memo=True, lean=True)
            t1 = time.perf_counter()
            fabgpu.memo_evict_block(csp, 100 + k)
            per.append(t1 - t0)
        else:
            out = fabgpu.preverify_block(csp, blk)
            per.append(time.perf_counter() - t0)
    dt = statistics.median(per)
    in_flight = None
    if args.threads > 1:             # several channels: the walk of one block overlaps the gates + device call of another
        import threading

        def caller(t):
            for k in range(args.steps):
                if args.memo:
                    fabgpu.preverify_block2(csp, blk, block_seq=10000 * (t + 1) + k, seed_memo=True, lean=True)
                    fabgpu.memo_evict_block(csp, 10000 * (t + 1) + k)
                else:
                    fabgpu.preverify_block2(csp, blk, lean=True)
        th = [threading.Thread(target=caller, args=(t,)) for t in range(args.threads)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        in_flight = {"callers": args.threads, "blocks": args.threads * args.steps, "ms_per_block_aggregate": wall / (args.threads * args.steps) * 1e3,
                     "validated_tx_per_s": args.tx * args.threads * args.steps / wall, "note": "memo eviction inside the timed loop" if args.memo else "flags only"}
    if args.memo:     # replay of the validators' bccsp.Verify lookups against a seeded block: all hits, timed
        out = fabgpu.preverify_block2(csp, blk, block_seq=7, seed_memo=True)
        nt = len(out["tuple_status"])
        keys = [(bytes(out["tuple_qxy"][i][:32]), bytes(out["tuple_qxy"][i][32:]), out["arena"][int(out["tuple_spans"][i][6]):int(out["tuple_spans"][i][6]) + int(out["tuple_spans"][i][7])],
                 bytes(out["tuple_digest"][i])) for i in range(0, nt, 7)]
        t0 = time.perf_counter()
        hits = sum(1 for k in keys if fabgpu.memo_lookup(csp, *k) == 0)
        lookup_us = (time.perf_counter() - t0) / len(keys) * 1e6
        assert hits == len(keys) and out["memo_seeded"] == nt
        fabgpu.memo_evict_block(csp, 7)
    routes = fabgpu.pass_routes(csp)    # walked on the device / on the host (--host-walk keeps every block on the host walk)
    # the one SHA-256 MCS.VerifyBlock needs that a GPU cannot parallelise: BlockDataHash over the concatenated envelopes, on this host
    t0 = time.perf_counter()
    hashlib.sha256(b"".join(envs) if envs else blk).digest()
    data_hash_ms = (time.perf_counter() - t0) * 1e3
    print(json.dumps({"metric": "validated tx/sec per block (block-level pre-verify pass, marshalled block in, flags out)", "value": args.tx / dt,
                      "unit": "tx/s", "ms_per_block": dt * 1e3, "ms_min": min(per) * 1e3, "ms_max": max(per) * 1e3, "signatures_per_s": 4 * args.tx / dt,
                      "tuples_through_key_tables": n_keyed, "callers_in_flight": in_flight, "routes": routes,
                      "mode": ("preverify2 + memo seeding (eviction not timed)" if args.memo else "preverify (flags only)") + (", %.0f ms idle between blocks" % args.idle_ms if args.idle_ms else ", back to back"),
                      "memo_lookup_us_via_ctypes": (lookup_us if args.memo else None),
                      "host_block_data_hash_ms": data_hash_ms, "host_sha256_GB_per_s": len(blk) / data_hash_ms / 1e6,
                      "config": {"workload": "%d endorser tx x (1 creator + 3 endorsement signatures), %.1f MB block, 6 signers, %d with a device table%s" % (
                          args.tx, len(blk) / 1e6, min(6, args.tables), (", block file " + os.path.basename(args.block_file)) if args.block_file else "")}, "checks": "creator signature, 3 endorsement signatures, TxID and proposal hash per transaction",
                      "parity": "every transaction flagged valid; corrupted blocks are covered by tests/test_block_prepass.py"}))
    csp.close()


if __name__ == "__main__":
    main()
