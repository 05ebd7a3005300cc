"""Test helper: an independent, by-the-book DECODER of the Fabric messages on the signature path — what the reference's Go
validators see when they unmarshal a block.  It shares no code with the product's C++ walker (fabric-mod_amd/csrc/
block_prepass.cpp) and materialises BYTES, not spans, so that tests can compare the two and hand the result to the CPU oracle.

Semantics follow golang/protobuf's proto.Unmarshal: for a singular bytes field the LAST occurrence wins, repeated fields append.
(The C++ walker deliberately refuses messages in which a singular field repeats; no honest marshaller writes one.)

Extraction rules (reference file:line):
    creator       core/common/validation/msgvalidation.go:258-298   identity = SignatureHeader.creator, msg = Envelope.payload
    endorsements  core/common/validation/statebased/validator_keylevel.go:246-258   msg = prp || endorser
    TxID          protoutil/proputils.go:357-375   hex(SHA-256(nonce || creator)) == ChannelHeader.tx_id
    proposal hash protoutil/txutils.go:431-447, msgvalidation.go:233-241
    block sigs    internal/peer/gossip/mcs.go:166-193   msg = Metadata.value || signature_header || BlockHeaderBytes(header)
    header bytes  protoutil/blockutils.go:38-58   ASN.1 DER {INTEGER number, OCTET STRING previous_hash, OCTET STRING data_hash}
    data hash     protoutil/blockutils.go:65-68   SHA-256 of the concatenated envelopes
"""
import base64
import hashlib


def _varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def fields(buf):
    """[(number, wire type, value)] of one message; value = int (varint) or bytes (length-delimited / fixed)."""
    out, pos = [], 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > len(buf):
                raise ValueError("truncated field")
            v, pos = bytes(buf[pos:pos + n]), pos + n
        elif wt == 1:
            v, pos = bytes(buf[pos:pos + 8]), pos + 8
        elif wt == 5:
            v, pos = bytes(buf[pos:pos + 4]), pos + 4
        else:
            raise ValueError("wire type %d" % wt)
        out.append((num, wt, v))
    return out


def last(buf, num, default=None):
    v = default
    for n, wt, val in fields(buf):
        if n == num:
            v = val
    return v


def every(buf, num):
    return [val for n, wt, val in fields(buf) if n == num]


def asn1_len(n):
    if n < 128:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([0x80 | len(b)]) + b


def block_header_bytes(number, previous_hash, data_hash):
    """protoutil.BlockHeaderBytes (protoutil/blockutils.go:38-58)"""
    nb = number.to_bytes(max(1, (number.bit_length() + 8) // 8), "big")   # minimal two's complement of a non-negative integer
    body = b"\x02" + asn1_len(len(nb)) + nb + b"\x04" + asn1_len(len(previous_hash)) + previous_hash + b"\x04" + asn1_len(len(data_hash)) + data_hash
    return b"\x30" + asn1_len(len(body)) + body


def pem_cert_der(pem: bytes) -> bytes:
    txt = pem.decode("ascii", "replace")
    a = txt.index("-----BEGIN CERTIFICATE-----") + len("-----BEGIN CERTIFICATE-----")
    b = txt.index("-----END CERTIFICATE-----")
    return base64.b64decode("".join(txt[a:b].split()))


def decode_block(raw: bytes):
    """-> dict(number, previous_hash, data_hash, envelopes=[bytes], metadata=[bytes], txs=[...], block_sigs=[...], data_hash_ok)
    tx = dict(type, channel, tx_id, creator=(identity, msg, sig), txid_expect (hex str or None), actions=[dict(prp, proposal_hash_expect,
              proposal_hash_input, endorsements=[(identity, msg, sig)])])"""
    header = last(raw, 1, b"")
    data = last(raw, 2, b"")
    meta = last(raw, 3, b"")
    number = last(header, 1, 0)
    prev = last(header, 2, b"")
    dh = last(header, 3, b"")
    envs = every(data, 1)
    metas = every(meta, 1)
    txs = []
    for env in envs:
        payload = last(env, 1, b"")
        sig = last(env, 2, b"")
        hdr = last(payload, 1, b"")
        chdr = last(hdr, 1, b"")
        shdr = last(hdr, 2, b"")
        typ = last(chdr, 1, 0)
        creator = last(shdr, 1, b"")
        nonce = last(shdr, 2, b"")
        tx = dict(type=typ, channel=(last(chdr, 4, b"")).decode(), tx_id=(last(chdr, 5, b"")).decode("latin1"),
                  creator=(creator, payload, sig), txid_computed=hashlib.sha256(nonce + creator).hexdigest(), actions=[])
        if typ == 3:
            for act in every(last(payload, 2, b""), 1):
                ahdr = last(act, 1, b"")
                cap = last(act, 2, b"")
                ccpp = last(cap, 1, b"")
                cea = last(cap, 2, b"")
                prp = last(cea, 1, b"")
                ends = []
                for e in every(cea, 2):
                    endorser = last(e, 1, b"")
                    ends.append((endorser, prp + endorser, last(e, 2, b"")))
                tx["actions"].append(dict(prp=prp, proposal_hash_expect=last(prp, 1, b""),
                                          proposal_hash_computed=hashlib.sha256(chdr + ahdr + ccpp).digest(), endorsements=ends))
        txs.append(tx)
    block_sigs = []
    if metas:
        m0 = metas[0]                                   # BlockMetadataIndex_SIGNATURES
        value = last(m0, 1, b"")
        hb = block_header_bytes(number, prev, dh)
        for ms in every(m0, 2):
            sh = last(ms, 1, b"")
            block_sigs.append((last(sh, 1, b""), value + sh + hb, last(ms, 2, b"")))
    return dict(number=number, previous_hash=prev, data_hash=dh, envelopes=envs, metadata=metas, txs=txs, block_sigs=block_sigs,
                data_hash_ok=(hashlib.sha256(b"".join(envs)).digest() == dh))


def identity_pubkey(identity: bytes):
    """msp.SerializedIdentity{1 mspid, 2 id_bytes = PEM x509} -> (mspid, (qx, qy) ints or None) via the KAT generator's DER reader."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import gen_ref_cert_kats as g
    try:
        mspid = last(identity, 1, b"").decode("latin1")
        idb = last(identity, 2, b"")
    except Exception:
        return "", None
    try:
        pub = g.parse_cert(pem_cert_der(idb))["pub"]
    except Exception:
        return mspid, None
    return mspid, (None if pub is None else (int(pub[0], 16), int(pub[1], 16)))
