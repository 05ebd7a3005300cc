"""CPU ORACLE loader (test infrastructure, NOT product code): ctypes view of oracle/libp256oracle.so
(p256_oracle.c) and oracle/libosslbaseline.so (ossl_baseline.c).  `ensure_built()` runs oracle/Makefile.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_ossl = None
u8p = ctypes.POINTER(ctypes.c_uint8)
u32p = ctypes.POINTER(ctypes.c_uint32)


def ensure_built():
    need = [os.path.join(HERE, n) for n in ("libp256oracle.so", "libosslbaseline.so")]
    src = [os.path.join(HERE, n) for n in ("p256_oracle.c", "ossl_baseline.c")]
    if all(os.path.exists(a) and os.path.getmtime(a) >= os.path.getmtime(b) for a, b in zip(need, src)):
        return
    subprocess.run(["make", "-C", HERE, "-s"], check=True)


def _p(a):
    return a.ctypes.data_as(u8p)


def lib():
    global _lib
    if _lib is None:
        ensure_built()
        _lib = ctypes.CDLL(os.path.join(HERE, "libp256oracle.so"))
        _lib.oracle_p256_verify_one.restype = ctypes.c_int
        _lib.oracle_der_unmarshal.restype = ctypes.c_int
        _lib.oracle_bccsp_verify.restype = ctypes.c_int
        _lib.oracle_p256_sign.restype = ctypes.c_int
    return _lib


def ossl():
    global _ossl
    if _ossl is None:
        ensure_built()
        _ossl = ctypes.CDLL(os.path.join(HERE, "libosslbaseline.so"))
    return _ossl


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def verify_batch(qx, qy, e, r, s):
    """n x 32 uint8 arrays (big-endian) -> status uint8[n]."""
    qx, qy, e, r, s = map(_c, (qx, qy, e, r, s))
    n = qx.shape[0]
    st = np.zeros(n, dtype=np.uint8)
    lib().oracle_p256_verify_batch(ctypes.c_size_t(n), _p(qx), _p(qy), _p(e), _p(r), _p(s), _p(st))
    return st


def ossl_verify_batch(qx, qy, e, r, s):
    qx, qy, e, r, s = map(_c, (qx, qy, e, r, s))
    n = qx.shape[0]
    st = np.zeros(n, dtype=np.uint8)
    ossl().ossl_p256_verify_batch(ctypes.c_size_t(n), _p(qx), _p(qy), _p(e), _p(r), _p(s), _p(st))
    return st


def verify_one(qx: bytes, qy: bytes, digest: bytes, r: bytes, s: bytes) -> int:
    return lib().oracle_p256_verify_one(qx, qy, digest, ctypes.c_size_t(len(digest)), r, s)


def der_unmarshal(sig: bytes):
    r = ctypes.create_string_buffer(32)
    s = ctypes.create_string_buffer(32)
    fl = ctypes.c_int(0)
    rc = lib().oracle_der_unmarshal(sig, ctypes.c_size_t(len(sig)), r, s, ctypes.byref(fl))
    return rc, r.raw, s.raw, fl.value


def bccsp_verify(qx: bytes, qy: bytes, sig: bytes, digest: bytes) -> int:
    return lib().oracle_bccsp_verify(qx, qy, sig, ctypes.c_size_t(len(sig)), digest, ctypes.c_size_t(len(digest)))


def sha256(msg: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib().oracle_sha256(msg, ctypes.c_size_t(len(msg)), out)
    return out.raw


def sha256_batch(arena, off):
    arena = _c(arena)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    n = off.shape[0] - 1
    out = np.zeros((n, 32), dtype=np.uint8)
    lib().oracle_sha256_batch(ctypes.c_size_t(n), _p(arena), off.ctypes.data_as(u32p), _p(out))
    return out


def sha256_verify_batch(arena, off, qx, qy, r, s, use_ossl=False):
    arena, qx, qy, r, s = map(_c, (arena, qx, qy, r, s))
    off = np.ascontiguousarray(off, dtype=np.uint32)
    n = off.shape[0] - 1
    st = np.zeros(n, dtype=np.uint8)
    fn = ossl().ossl_sha256_p256_verify_batch if use_ossl else lib().oracle_sha256_p256_verify_batch
    fn(ctypes.c_size_t(n), _p(arena), off.ctypes.data_as(u32p), _p(qx), _p(qy), _p(r), _p(s), _p(st))
    return st


N_INT = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551


def make_batch(n, seed, invalid_frac=0.0, digests=None):
    """Seeded synthetic tuples: fresh keypair per signature, low-S, optional invalid mix
    (equal parts flipped digest bit / wrong key / high-S mirror / r+1).  Returns dict of n x 32 arrays
    plus `kind` (0 valid, 1..4 the mutation applied).  `digests` (n x 32) overrides the random e."""
    rng = np.random.default_rng(seed)

    def scalars(m):
        a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
        a[:, 0] &= 0x7F  # < 2^255 < n
        a[:, 31] |= 1    # nonzero
        return a
    d, k = scalars(n), scalars(n)
    e = rng.integers(0, 256, size=(n, 32), dtype=np.uint8) if digests is None else _c(digests).copy()
    qx = np.zeros((n, 32), np.uint8)
    qy = np.zeros((n, 32), np.uint8)
    r = np.zeros((n, 32), np.uint8)
    s = np.zeros((n, 32), np.uint8)
    lib().oracle_p256_make_batch(ctypes.c_size_t(n), _p(d), _p(k), _p(e), _p(qx), _p(qy), _p(r), _p(s))
    kind = np.zeros(n, dtype=np.uint8)
    nbad = int(round(n * invalid_frac))
    if nbad:
        idx = rng.choice(n, size=nbad, replace=False)
        for j, i in enumerate(idx):
            m = 1 + j % 4
            kind[i] = m
            if m == 1:
                e[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
            elif m == 2:
                o = (i + 1) % n
                qx[i], qy[i] = qx[o].copy(), qy[o].copy()
            elif m == 3:
                sv = N_INT - int.from_bytes(s[i].tobytes(), "big")
                s[i] = np.frombuffer(sv.to_bytes(32, "big"), dtype=np.uint8)
            else:
                rv = (int.from_bytes(r[i].tobytes(), "big") + 1) % (1 << 256)
                r[i] = np.frombuffer(rv.to_bytes(32, "big"), dtype=np.uint8)
    return dict(qx=qx, qy=qy, e=e, r=r, s=s, kind=kind)


def make_pool_batch(n, seed, nkeys=16, invalid_frac=0.0, digests=None):
    """Like make_batch, but the signers are drawn from a pool of `nkeys` keypairs (the realistic Fabric shape: few distinct
    endorsers per block).  Adds key_index (n, into the pool) and pool_qx / pool_qy (nkeys x 32).
    Invalid mix: flipped digest bit / signature by another pool key / high-S mirror / r+1."""
    rng = np.random.default_rng(seed)

    def scalars(m):
        a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8)
        a[:, 0] &= 0x7F
        a[:, 31] |= 1
        return a
    dpool = scalars(nkeys)
    key_index = rng.integers(0, nkeys, size=n).astype(np.uint32)
    d, k = dpool[key_index], scalars(n)
    e = rng.integers(0, 256, size=(n, 32), dtype=np.uint8) if digests is None else _c(digests).copy()
    qx = np.zeros((n, 32), np.uint8); qy = np.zeros((n, 32), np.uint8)
    r = np.zeros((n, 32), np.uint8); s = np.zeros((n, 32), np.uint8)
    lib().oracle_p256_make_batch(ctypes.c_size_t(n), _p(np.ascontiguousarray(d)), _p(k), _p(e), _p(qx), _p(qy), _p(r), _p(s))
    pool_qx = np.zeros((nkeys, 32), np.uint8); pool_qy = np.zeros((nkeys, 32), np.uint8)
    for j in range(nkeys):
        i = int(np.nonzero(key_index == j)[0][0]) if (key_index == j).any() else None
        if i is not None:
            pool_qx[j], pool_qy[j] = qx[i], qy[i]
        else:
            lib().oracle_p256_pubkey(_p(np.ascontiguousarray(dpool[j])), _p(pool_qx[j:j + 1]), _p(pool_qy[j:j + 1]))
    kind = np.zeros(n, dtype=np.uint8)
    nbad = int(round(n * invalid_frac))
    if nbad:
        idx = rng.choice(n, size=nbad, replace=False)
        for j, i in enumerate(idx):
            m = 1 + j % 4
            kind[i] = m
            if m == 1:
                e[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
            elif m == 2:
                key_index[i] = (key_index[i] + 1) % nkeys
                qx[i], qy[i] = pool_qx[key_index[i]], pool_qy[key_index[i]]
            elif m == 3:
                sv = N_INT - int.from_bytes(s[i].tobytes(), "big")
                s[i] = np.frombuffer(sv.to_bytes(32, "big"), dtype=np.uint8)
            else:
                rv = (int.from_bytes(r[i].tobytes(), "big") + 1) % (1 << 256)
                r[i] = np.frombuffer(rv.to_bytes(32, "big"), dtype=np.uint8)
    return dict(qx=qx, qy=qy, e=e, r=r, s=s, kind=kind, key_index=key_index, pool_qx=pool_qx, pool_qy=pool_qy)
