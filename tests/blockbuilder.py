"""Test helper: a minimal protobuf ENCODER for the Fabric messages the block pre-verify pass walks (field numbers of
fabric-protos-go: common/common.proto, peer/transaction.proto, peer/proposal_response.proto, msp/identities.proto), used to
build synthetic blocks.  Encoder and the product's C++ walker were written independently of each other's code; the outer
layers of the walker are additionally pinned by the reference's own block fixtures (test_block_prepass.py)."""


def varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def fbytes(num: int, data: bytes) -> bytes:
    return varint((num << 3) | 2) + varint(len(data)) + data


def fvarint(num: int, v: int) -> bytes:
    return varint(num << 3) + varint(v)


def serialized_identity(mspid: str, pem: str) -> bytes:
    return fbytes(1, mspid.encode()) + fbytes(2, pem.encode())


def serialized_idemix_identity(mspid: str, nym_x: bytes, nym_y: bytes, ou: bytes = b"\x0a\x03OU1", role: bytes = b"\x08\x01", proof: bytes = b"proof") -> bytes:
    """msp.SerializedIdentity{mspid, id_bytes = msp.SerializedIdemixIdentity{1 nym_x, 2 nym_y, 3 ou, 4 role, 5 proof}} (msp/idemixmsp.go:605-640)"""
    return fbytes(1, mspid.encode()) + fbytes(2, fbytes(1, nym_x) + fbytes(2, nym_y) + fbytes(3, ou) + fbytes(4, role) + fbytes(5, proof))


def channel_header(typ: int, channel: str, txid: str) -> bytes:
    return (fvarint(1, typ) if typ else b"") + fvarint(2, 0 + 1) + fbytes(4, channel.encode()) + fbytes(5, txid.encode()) + fvarint(6, 0 + 7)


def signature_header(creator: bytes, nonce: bytes) -> bytes:
    return fbytes(1, creator) + fbytes(2, nonce)


def endorser_tx_payload(typ, channel, txid, creator, nonce, actions) -> bytes:
    """actions: list of (ccpp bytes, prp bytes, [(endorser identity bytes, signature bytes), ...]) -> common.Payload bytes."""
    hdr = fbytes(1, channel_header(typ, channel, txid)) + fbytes(2, signature_header(creator, nonce))
    tx = b""
    for ccpp, prp, ends in actions:
        cea = fbytes(1, prp) + b"".join(fbytes(2, fbytes(1, e) + fbytes(2, s)) for e, s in ends)
        cap = fbytes(1, ccpp) + fbytes(2, cea)
        tx += fbytes(1, fbytes(1, signature_header(creator, nonce)) + fbytes(2, cap))
    return fbytes(1, hdr) + fbytes(2, tx)


def compute_txid(nonce: bytes, creator: bytes) -> str:
    """protoutil.ComputeTxID (protoutil/proputils.go:357-364)"""
    import hashlib
    return hashlib.sha256(nonce + creator).hexdigest()


def proposal_hash(typ, channel, txid, creator, nonce, ccpp: bytes) -> bytes:
    """protoutil.GetProposalHash2 (protoutil/txutils.go:431-447): channel header || the action's signature header || proposal payload"""
    import hashlib
    return hashlib.sha256(channel_header(typ, channel, txid) + signature_header(creator, nonce) + ccpp).digest()


def proposal_response_payload(phash: bytes, extension: bytes) -> bytes:
    """peer.ProposalResponsePayload{1 proposal_hash, 2 extension}"""
    return fbytes(1, phash) + fbytes(2, extension)


def consistent_endorser_tx(channel, creator, nonce, ccpp, extension, sign_endorsements, bad_txid=False, bad_phash=False):
    """An ENDORSER_TRANSACTION payload whose TxID and proposal hash are what the reference's validators recompute.
    sign_endorsements(prp) -> [(endorser identity bytes, signature bytes), ...].  Returns (payload bytes, prp bytes)."""
    txid = compute_txid(nonce, creator)
    if bad_txid:
        txid = txid[:-1] + ("0" if txid[-1] != "0" else "1")
    ph = proposal_hash(3, channel, txid, creator, nonce, ccpp)
    if bad_phash:
        ph = bytes([ph[0] ^ 1]) + ph[1:]
    prp = proposal_response_payload(ph, extension)
    return endorser_tx_payload(3, channel, txid, creator, nonce, [(ccpp, prp, sign_endorsements(prp))]), prp


def envelope(payload: bytes, signature: bytes) -> bytes:
    return fbytes(1, payload) + fbytes(2, signature)


def block(number: int, envelopes) -> bytes:
    header = fvarint(1, number) + fbytes(2, b"\x11" * 32) + fbytes(3, b"\x22" * 32)
    data = b"".join(fbytes(1, e) for e in envelopes)
    metadata = fbytes(1, b"") + fbytes(1, b"meta")
    return fbytes(1, header) + fbytes(2, data) + fbytes(3, metadata)
