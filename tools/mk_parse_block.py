"""Writes /tmp/blk10k.bin: a 10 000-transaction endorser block (reference-consistent TxIDs and proposal hashes, fake signatures) for
tools/parse_bench.cpp."""
import sys, os, json, time, hashlib, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('fabric-mod_amd','oracle','tests'): sys.path.insert(0, os.path.join(ROOT,p))
import numpy as np, blockbuilder as bb, fabgpu
ids=[i for i in json.load(open(ROOT+'/tests/golden/block_identities.json'))['identities'] if i['curve']=='prime256v1']
sid=[bb.serialized_identity('Org1MSP',i['pem']) for i in ids]
rng=np.random.default_rng(1)
envs=[]
fake=b'\x30\x44\x02\x20'+b'\x11'*32+b'\x02\x20'+b'\x22'*32
for t in range(10000):
    c=4+t%2
    payload,_=bb.consistent_endorser_tx('mychannel',sid[c],bytes(rng.integers(0,256,size=24,dtype=np.uint8)),bytes(rng.integers(0,256,size=300,dtype=np.uint8)),bytes(rng.integers(0,256,size=990,dtype=np.uint8)),lambda prp:[(sid[j],fake) for j in (0,1,2)])
    envs.append(bb.envelope(payload,fake))
blk=bb.block(1,envs)
open('/tmp/blk10k.bin','wb').write(blk)
print(len(blk))
