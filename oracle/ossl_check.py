"""CPU ORACLE helper (test infrastructure, NOT product code).

ctypes binding to the system OpenSSL 3 libcrypto, used as an implementation that is independent
of both the restated oracle (bccsp_sw_oracle.py / p256_oracle.c) and the HIP kernels:
raw ECDSA P-256 verification of a (Qx,Qy,e,r,s) tuple via ECDSA_do_verify, and SHA-256 via
hashlib (which is OpenSSL-backed).  OpenSSL knows nothing of Fabric's low-S rule
(bccsp/sw/ecdsa.go:47-54); callers apply that gate themselves.
"""
import ctypes
import ctypes.util

_lib = None


def lib():
    global _lib
    if _lib is None:
        name = ctypes.util.find_library("crypto") or "libcrypto.so.3"
        L = ctypes.CDLL(name)
        vp = ctypes.c_void_p
        L.EC_KEY_new_by_curve_name.restype = vp
        L.EC_KEY_new_by_curve_name.argtypes = [ctypes.c_int]
        L.EC_KEY_free.argtypes = [vp]
        L.EC_KEY_set_public_key_affine_coordinates.argtypes = [vp, vp, vp]
        L.BN_bin2bn.restype = vp
        L.BN_bin2bn.argtypes = [ctypes.c_char_p, ctypes.c_int, vp]
        L.BN_free.argtypes = [vp]
        L.ECDSA_SIG_new.restype = vp
        L.ECDSA_SIG_free.argtypes = [vp]
        L.ECDSA_SIG_set0.argtypes = [vp, vp, vp]
        L.ECDSA_do_verify.argtypes = [ctypes.c_char_p, ctypes.c_int, vp, vp]
        _lib = L
    return _lib


NID_P256 = 415  # NID_X9_62_prime256v1


def verify_raw(qx: int, qy: int, digest: bytes, r: int, s: int):
    """True/False from OpenSSL; None if OpenSSL refuses the key (off-curve) or the inputs."""
    L = lib()
    if r <= 0 or s <= 0 or r >= 1 << 256 or s >= 1 << 256:
        return False
    key = L.EC_KEY_new_by_curve_name(NID_P256)
    bx = L.BN_bin2bn(qx.to_bytes(32, "big"), 32, None)
    by = L.BN_bin2bn(qy.to_bytes(32, "big"), 32, None)
    try:
        if L.EC_KEY_set_public_key_affine_coordinates(key, bx, by) != 1:
            return None
        sig = L.ECDSA_SIG_new()
        br = L.BN_bin2bn(r.to_bytes(32, "big"), 32, None)
        bs = L.BN_bin2bn(s.to_bytes(32, "big"), 32, None)
        L.ECDSA_SIG_set0(sig, br, bs)  # sig owns br, bs
        d = digest[:32] if len(digest) > 32 else digest
        rc = L.ECDSA_do_verify(d, len(d), sig, key)
        L.ECDSA_SIG_free(sig)
        return True if rc == 1 else False if rc == 0 else None
    finally:
        L.BN_free(bx)
        L.BN_free(by)
        L.EC_KEY_free(key)
