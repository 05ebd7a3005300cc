import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [os.path.join(ROOT, "fabric-mod_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import blockbuilder as bb, blockgen, fabgpu
from test_device_walk import _cert_with_long_issuer
der = blockgen._pem_der(blockgen._IDS[4]["pem"])
csp = fabgpu.GPUCSP(devices=[0])
ids = {"normal": bb.serialized_identity("Org1MSP", blockgen._IDS[4]["pem"])}
for pad in (500, 1500, 2500, 3300, 6000):
    ids["pad%d" % pad] = bb.serialized_identity("Org1MSP", blockgen._pem_wrap(_cert_with_long_issuer(der, pad)))
for name, ident in ids.items():
    for n in (1, 256):
        fabgpu.idfix_probe(csp, [ident] * n)
        t0 = time.perf_counter()
        for _ in range(20):
            code, key = fabgpu.idfix_probe(csp, [ident] * n)
        dt = (time.perf_counter() - t0) / 20 * 1e6
        print("%-8s len %5d  n %3d  %.0f us per call  code %d" % (name, len(ident), n, dt, int(code[0])), flush=True)
