#!/usr/bin/env python3
"""Pre-builds the blocks of the block-pass benches on the CPU (the signing is pure Python / C oracle and needs no GPU): written to
.bench_blocks/ at the repo root - git-ignored, but it travels to the GPU box with the snapshot - so that GPU-minutes are not spent
signing.      python tools/make_bench_blocks.py idemix 10000 5     -> .bench_blocks/idemix_10000_5.bin (every 5th creator idemix)
              python tools/make_bench_blocks.py ecdsa 10000 0      -> .bench_blocks/ecdsa_10000_0.bin  (x509 creators only)
              python tools/make_bench_blocks.py passlegs 10000 0   -> the blocks of bench.py's block_pass legs (tests/blockgen.py):
                  friendly_10000.bin (six signers), distinct_10000.bin (every creator a certificate nobody has met),
                  fresh1pct_10000.bin (10 x 1 % envelopes by never-seen creators, patched into the friendly block at bench time)"""
import ctypes
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "oracle")]
import numpy as np   # noqa: E402

import bccsp_sw_oracle as po   # noqa: E402
import blockbuilder as bb   # noqa: E402
import coracle   # noqa: E402
import idemix_oracle as io   # noqa: E402
from idemix_common import be32, fixtures   # noqa: E402


def mixed_block(kind, ntx, every, progress=False):
    """ntx endorser transactions (1 creator + 3 endorsements); kind 'idemix': every `every`-th creator is an idemix pseudonym of the
    fixtures' IdemixMSP1 with its nym signature"""
    ids = [i for i in json.load(open(os.path.join(ROOT, "tests", "golden", "block_identities.json")))["identities"] if i["curve"] == "prime256v1"]
    sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in ids]
    L = coracle.lib()
    rng = np.random.default_rng(1)
    prng = random.Random(7)
    fx = fixtures()
    ipk, sk = fx["MSP1OU1"]["ipk"], fx["MSP1OU1"]["signer"].sk

    def sign(k, msg):
        d = int(ids[k]["d"], 16).to_bytes(32, "big")
        e = hashlib.sha256(msg).digest()
        nonce = b"\x00" + bytes(rng.integers(1, 255, size=31, dtype=np.uint8))
        r, s = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        assert L.oracle_p256_sign(d, e, nonce, 1, r, s) == 0
        return po.marshal_ecdsa_signature(int.from_bytes(r.raw, "big"), int.from_bytes(s.raw, "big"))
    envs = []
    for t in range(ntx):
        picks = [int(j) for j in rng.choice(4, size=3, replace=False)]
        ends = lambda prp: [(sid[j], sign(j, prp + sid[j])) for j in picks]   # noqa: E731
        args = (bytes(rng.integers(0, 256, size=24, dtype=np.uint8)), bytes(rng.integers(0, 256, size=300, dtype=np.uint8)),
                bytes(rng.integers(0, 256, size=990, dtype=np.uint8)))
        if kind == "idemix" and t % every == 0:
            nym, r_nym = io.make_nym(sk, ipk, prng)
            cbytes = bb.serialized_idemix_identity("IdemixMSP1", be32(nym[0]), be32(nym[1]))
            payload, _ = bb.consistent_endorser_tx("mychannel", cbytes, *args, ends)
            envs.append(bb.envelope(payload, io.nym_signature_marshal(io.nym_sign(sk, nym, r_nym, ipk, payload, prng))))
        else:
            c = 4 + t % 2
            payload, _ = bb.consistent_endorser_tx("mychannel", sid[c], *args, ends)
            envs.append(bb.envelope(payload, sign(c, payload)))
        if progress and t % 1000 == 999:
            print(t + 1, file=sys.stderr)
    return bb.block(1, envs)


def main():
    kind, ntx, every = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    assert kind in ("idemix", "ecdsa", "passlegs")
    if kind == "passlegs":
        import blockgen
        os.makedirs(os.path.join(ROOT, ".bench_blocks"), exist_ok=True)
        per_block = max(1, ntx // 100)
        for name, build in (("friendly_%d.bin" % ntx, lambda: blockgen.endorser_block(ntx, 1)[0]),
                            ("distinct_%d.bin" % ntx, lambda: blockgen.endorser_block(ntx, 3, creators=blockgen.fresh_identities(ntx, 4))[0]),
                            ("fresh1pct_%d.bin" % ntx, lambda: blockgen.pack_envelopes(
                                blockgen.endorser_block(10 * per_block, 5, creators=blockgen.fresh_identities(10 * per_block, 6))[1]))):
            out = os.path.join(ROOT, ".bench_blocks", name)
            open(out, "wb").write(build())
            print(out, os.path.getsize(out))
        return
    blk = mixed_block(kind, ntx, every, progress=True)
    os.makedirs(os.path.join(ROOT, ".bench_blocks"), exist_ok=True)
    out = os.path.join(ROOT, ".bench_blocks", "%s_%d_%d.bin" % (kind, ntx, every))
    open(out, "wb").write(blk)
    print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
