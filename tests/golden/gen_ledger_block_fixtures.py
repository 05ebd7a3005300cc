#!/usr/bin/env python3
"""Extracts the reference's own sample ledgers into golden block fixtures (test infrastructure).

    python3 tests/golden/gen_ledger_block_fixtures.py /root/reference  ->  tests/golden/ledger_blocks.json

Sources: core/ledger/kvledger/tests/testdata/{v11,v11 with commit hashes,v13_statecouchdb,v20}/sample_ledgers*/
ledgersData.zip — block files written by real Fabric 1.1 / 1.3 / 2.0 peers: every ENDORSER_TRANSACTION in them carries a
reference-produced creator signature, 1..3 endorsement signatures, a TxID and a proposal hash, and every block carries the
orderer's block signature in its metadata and the committing peer's TRANSACTIONS_FILTER.

Block file format (common/ledger/blkstorage): a file is a sequence of  varint(len) || serialized block
(blockfile_mgr.go:305, block_stream.go:99-115); a serialized block is (block_serialization.go:28-170)
    varint(number) || rawbytes(data_hash) || rawbytes(previous_hash)          addHeaderBytes   :82-93
    varint(n_tx)   || n_tx x rawbytes(envelope)                               addDataBytes...  :95-116
    varint(n_meta) || n_meta x rawbytes(metadata entry)                       addMetadataBytes :118-132
with rawbytes = varint(len) || bytes.  This script re-marshals each as a protobuf common.Block
(Block{1 header{1 number, 2 previous_hash, 3 data_hash}, 2 data{1 data...}, 3 metadata{1 metadata...}}) — the form the
gossip/deliver path hands to the validators and to fabgpu_csp_block_preverify — and records next to it the pieces a test
needs to check the pass WITHOUT trusting this repository's walker: the TRANSACTIONS_FILTER bytes (metadata index 2), the
block's data_hash and the number.  Nothing here interprets envelopes; that is the product's job and the tests' subject.
"""
import base64
import io
import json
import os
import sys
import zipfile

SOURCES = [
    ("v11", "core/ledger/kvledger/tests/testdata/v11/sample_ledgers/ledgersData.zip"),
    ("v11_commit_hashes", "core/ledger/kvledger/tests/testdata/v11/sample_ledgers_with_commit_hashes/ledgersData.zip"),
    ("v13_statecouchdb", "core/ledger/kvledger/tests/testdata/v13_statecouchdb/sample_ledgers/ledgersData.zip"),
    ("v20", "core/ledger/kvledger/tests/testdata/v20/sample_ledgers/ledgersData.zip"),
    # (v20_couchdb/sample_ledgers/ledgersData.zip holds state databases only - no chains/ directory, no block file)
]


def get_varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def get_raw(buf, pos):
    n, pos = get_varint(buf, pos)
    assert pos + n <= len(buf)
    return bytes(buf[pos:pos + n]), pos + n


def put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def fld_bytes(num, data):
    return put_varint((num << 3) | 2) + put_varint(len(data)) + data


def fld_varint(num, v):
    return put_varint(num << 3) + put_varint(v)


def split_block_file(buf):
    pos = 0
    while pos < len(buf):
        n, p2 = get_varint(buf, pos)
        if n == 0 or p2 + n > len(buf):      # a partially written tail block (block_stream.go:117-125) — none in the fixtures
            break
        yield bytes(buf[p2:p2 + n])
        pos = p2 + n


def deserialize(ser):
    pos = 0
    number, pos = get_varint(ser, pos)
    data_hash, pos = get_raw(ser, pos)
    prev_hash, pos = get_raw(ser, pos)
    n_tx, pos = get_varint(ser, pos)
    envs = []
    for _ in range(n_tx):
        e, pos = get_raw(ser, pos)
        envs.append(e)
    n_meta, pos = get_varint(ser, pos)
    meta = []
    for _ in range(n_meta):
        m, pos = get_raw(ser, pos)
        meta.append(m)
    assert pos == len(ser), "trailing bytes in a serialized block"
    return number, data_hash, prev_hash, envs, meta


def marshal_block(number, data_hash, prev_hash, envs, meta):
    # proto3: zero / empty scalars are omitted by golang/protobuf, repeated bytes always written
    header = (fld_varint(1, number) if number else b"") + (fld_bytes(2, prev_hash) if prev_hash else b"") + (fld_bytes(3, data_hash) if data_hash else b"")
    data = b"".join(fld_bytes(1, e) for e in envs)
    metadata = b"".join(fld_bytes(1, m) for m in meta)
    return fld_bytes(1, header) + fld_bytes(2, data) + fld_bytes(3, metadata)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ledger_blocks.json")
    blocks, seen = [], {}
    for tag, rel in SOURCES:
        zf = zipfile.ZipFile(os.path.join(ref, rel))
        for name in sorted(zf.namelist()):
            if "/blockfile_" not in name:
                continue
            chain = name.split("/")[-2]
            for ser in split_block_file(zf.read(name)):
                number, data_hash, prev_hash, envs, meta = deserialize(ser)
                raw = marshal_block(number, data_hash, prev_hash, envs, meta)
                rec = {"source": tag, "zip": rel, "file": name, "chain": chain, "number": number, "n_tx": len(envs),
                       "data_hash": data_hash.hex(), "previous_hash": prev_hash.hex(),
                       "tx_filter": (meta[2].hex() if len(meta) > 2 else None),
                       "n_metadata": len(meta)}
                # identical blocks recur (v11 == v11_commit_hashes up to metadata; ledger1 vs ledger2 differ): store bytes once
                key = raw
                if key in seen:
                    rec["same_as"] = seen[key]
                else:
                    seen[key] = len(blocks)
                    rec["block_b64"] = base64.b64encode(raw).decode()
                blocks.append(rec)
    with io.open(out_path, "w") as f:
        json.dump({"generator": "tests/golden/gen_ledger_block_fixtures.py", "format": "protobuf common.Block, base64",
                   "blocks": blocks}, f, indent=0, sort_keys=True)
        f.write("\n")
    n_tx = sum(b["n_tx"] for b in blocks)
    print("%d blocks (%d distinct), %d envelopes -> %s (%d bytes)" % (len(blocks), len(seen), n_tx, out_path, os.path.getsize(out_path)))


if __name__ == "__main__":
    main()
