#!/usr/bin/env python3
"""Writes a 10 000-transaction marshalled block with fake signatures (the walker does not look at them) to argv[1]: input of the
walker-only timing harness (fabric-mod_amd/lib/walk_*: ParseBlock in a loop, no device)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "fabric-mod_amd")]
import numpy as np   # noqa: E402

import blockbuilder as bb   # noqa: E402

ids = [i for i in json.load(open(os.path.join(ROOT, "tests", "golden", "block_identities.json")))["identities"] if i["curve"] == "prime256v1"]
sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in ids]
rng = np.random.default_rng(1)
fake = b"\x30\x44\x02\x20" + b"\x11" * 32 + b"\x02\x20" + b"\x22" * 32
envs = []
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10000):
    payload, _ = bb.consistent_endorser_tx("mychannel", sid[4 + t % 2], bytes(rng.integers(0, 256, size=24, dtype=np.uint8)),
                                           bytes(rng.integers(0, 256, size=300, dtype=np.uint8)), bytes(rng.integers(0, 256, size=990, dtype=np.uint8)),
                                           lambda prp: [(sid[j], fake) for j in (0, 1, 2)])
    envs.append(bb.envelope(payload, fake))
open(sys.argv[1], "wb").write(bb.block(1, envs))
