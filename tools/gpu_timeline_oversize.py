#!/usr/bin/env python3
"""A few device-route passes over the friendly 10 000-transaction block with ONE creator whose certificate the device decoder cannot
decide (its key beyond the 3 KiB window) - or, with `friendly`, over the block as it is - for a rocprofv3 --kernel-trace timeline."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np   # noqa: E402

import blockbuilder as bb   # noqa: E402
import blockgen   # noqa: E402
import fabgpu   # noqa: E402
from test_device_walk import _cert_with_long_issuer   # noqa: E402

blk = open(os.path.join(ROOT, ".bench_blocks", "friendly_10000.bin"), "rb").read()
if "friendly" not in sys.argv:
    _, envs = blockgen.split_envelopes(blk)
    fx = blockgen.fixture_signers()
    der = blockgen._pem_der(blockgen._IDS[4]["pem"])
    far = bb.serialized_identity("Org1MSP", blockgen._pem_wrap(_cert_with_long_issuer(der, 3300)))
    env_far = blockgen.endorser_tx(13, np.random.default_rng(79), (far, fx[4][1]), [fx[0], fx[1], fx[2]], blockgen.make_signer(80))
    blk = bb.block(1, envs[:13] + [env_far] + envs[14:])
csp = fabgpu.GPUCSP(devices=[0], concurrent_passes=1, expect_block_bytes=len(blk) + (1 << 20), expect_tuples=40256)
for k in range(10):
    r = fabgpu.preverify_block2(csp, bytes(bytearray(blk)), block_seq=k, lean=True)
print(r["ms_stage"], r["n_keyed"], int((np.asarray(r["tx_flags"]) != 0).sum()))
csp.close()
