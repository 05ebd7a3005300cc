#!/usr/bin/env python3
"""Copies the reference's idemix KEY-MATERIAL fixtures (binary protobuf test data, not source code) into one JSON file so
that the oracle's pins travel to machines without /root/reference:  python3 tests/golden/gen_idemix_fixtures.py

Source: msp/testdata/idemix/<MSP>/{msp/IssuerPublicKey, ca/IssuerSecretKey, user/SignerConfig} of the reference tree
(used there by msp/idemixmsp_test.go).  Output: tests/golden/idemix_fixtures.json
"""
import json
import os

BASE = "/root/reference/msp/testdata/idemix"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "idemix_fixtures.json")


def main():
    out = {}
    for name in sorted(os.listdir(BASE)):
        d = os.path.join(BASE, name)
        ent = {}
        for key, rel in (("ipk", "msp/IssuerPublicKey"), ("isk", "ca/IssuerSecretKey"), ("signer_config", "user/SignerConfig")):
            p = os.path.join(d, rel)
            if os.path.exists(p):
                ent[key] = open(p, "rb").read().hex()
        if "ipk" in ent:
            out[name] = ent
    with open(OUT, "w") as f:
        json.dump({"source": "msp/testdata/idemix/*", "msps": out}, f, indent=1)
    print("wrote", OUT, {k: sorted(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
