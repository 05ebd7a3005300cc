#!/bin/bash
# Memory-safety fuzzing of the host-side parsers that see untrusted bytes in a peer - the block walker, the idemix identity /
# NymSignature walkers, the PEM decoder, the x509 SubjectPublicKeyInfo walker and the ECDSA DER gate - under AddressSanitizer and
# UndefinedBehaviorSanitizer (ROCm's clang: fp256.h uses clang builtins).  Mutation fuzzing from valid inputs: bit flips, random
# bytes, 0xFF / long-form length bytes, truncation; every mutant is copied to an exact-size heap block so that any read past its
# end is caught, and every span the walker reports is checked to lie inside the buffer.  Host only, no GPU.
#   tools/fuzz/run.sh        (about a minute; 20 000 block mutants - orderer block signatures included since round 2 - and 200 000 certificate
#                            mutants through the SPKI walker, the TBS / signature splitter of the x509 batch check and the DER gate: no finding)
# The block mutants also go through the code the DEVICE walk is compiled from (block_walk_core.h): the count / prefix sum / write
# procedure into exact-size arrays, the host's outline with its payload spans, the signature gate of the common DER shape - each
# compared with what ParseBlock says.
set -e
cd "$(dirname "$0")/../.."
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
SRC=fabric-mod_amd/csrc
FLAGS="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -I$SRC -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__"
python3 - <<'PY'
import json, os, sys
ROOT = os.getcwd()
for p in ("fabric-mod_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import blockbuilder as bb
ids = [i for i in json.load(open(ROOT + "/tests/golden/block_identities.json"))["identities"] if i["curve"] == "prime256v1"]
sid = [bb.serialized_identity("Org1MSP", i["pem"]) for i in ids]
rng = np.random.default_rng(1)
fake = b"\x30\x44\x02\x20" + b"\x11" * 32 + b"\x02\x20" + b"\x22" * 32
envs = []
for t in range(40):
    if t % 7 == 3:   # an idemix creator with a NymSignature
        c = bb.serialized_idemix_identity("IdemixMSP1", b"\x01" * 32, b"\x02" * 32)
        sig = bb.fbytes(1, b"\x03" * 32) + bb.fbytes(2, b"\x04" * 32) + bb.fbytes(3, b"\x05" * 32) + bb.fbytes(4, b"\x06" * 32)
    else:
        c, sig = sid[4 + t % 2], fake
    payload, _ = bb.consistent_endorser_tx("mychannel", c, bytes(rng.integers(0, 256, size=24, dtype=np.uint8)), bytes(rng.integers(0, 256, size=60, dtype=np.uint8)),
                                           bytes(rng.integers(0, 256, size=90, dtype=np.uint8)), lambda prp: [(sid[j], fake) for j in (0, 1, 2)])
    envs.append(bb.envelope(payload, sig))
# a block with real SIGNATURES metadata (two orderer signatures over the block): common.Metadata{1 value, 2 MetadataSignature{1 signature_header, 2 signature}}
hdr = bb.fvarint(1, 300) + bb.fbytes(2, b"\x11" * 32) + bb.fbytes(3, b"\x22" * 32)
msig = lambda k: bb.fbytes(2, bb.fbytes(1, bb.signature_header(sid[k], b"n" * 24)) + bb.fbytes(2, fake))
meta0 = bb.fbytes(1, b"\x0a\x02\x08\x05") + msig(0) + msig(1)
blk = bb.fbytes(1, hdr) + bb.fbytes(2, b"".join(bb.fbytes(1, e) for e in envs)) + bb.fbytes(3, bb.fbytes(1, meta0) + bb.fbytes(1, b"") + bb.fbytes(1, b"\x00" * 40))
open("/tmp/blk_small.bin", "wb").write(blk)
open("/tmp/cert.pem", "w").write(ids[0]["pem"])
open("/tmp/ipk.bin", "wb").write(bytes.fromhex(json.load(open(ROOT + "/tests/golden/idemix_fixtures.json"))["msps"]["MSP1OU1"]["ipk"]))
PY
$CXX $FLAGS tools/fuzz/fuzz_walk.cpp $SRC/block_prepass.cpp $SRC/idemix_host.cpp tools/fuzz/stubs.cpp -o /tmp/fuzz_walk -lpthread
$CXX $FLAGS tools/fuzz/fuzz_cert.cpp $SRC/block_prepass.cpp $SRC/bccsp_host.cpp $SRC/idemix_host.cpp tools/fuzz/stubs.cpp -o /tmp/fuzz_cert -lpthread
/tmp/fuzz_walk
/tmp/fuzz_cert
