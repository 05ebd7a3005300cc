"""fabgpu - Python host binding of the MI355X block-validation signature verifier.

The reference is Go; this image has no Go toolchain, so the parity tests drive the C ABI
(include/fabgpu.h) and the C++ host mirror (include/fabgpu_bccsp.h) through ctypes with the
reference's vocabulary:

    bccsp/bccsp.go:90-134          BCCSP.Hash / Verify / KeyImport   -> GPUCSP.hash / verify / key_import
    bccsp/utils/ecdsa.go:43-92     UnmarshalECDSASignature / IsLowS  -> unmarshal_ecdsa_signature / is_low_s
    msp/identities.go:169-196      identity.Verify                   -> Identity.verify
    internal/pkg/txflags           one verdict per tx                -> validate_block_endorsements

There is no CPU implementation of SHA-256 or of the curve arithmetic behind this module: if
libfabgpu.so is missing or no gfx950 device is present, construction raises (it never falls back).
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FABGPU_LIB_PATH: A/B experiments load another build of the same library (tools/, never the tests or the driver)
_LIB_PATH = os.environ.get("FABGPU_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libfabgpu.so")

FABGPU_OK = 0
FABGPU_EINVAL = -1
FLAG_ONE_LANE_ONLY = 1   # fabgpu.h FABGPU_FLAG_ONE_LANE_ONLY
FLAG_TIME_KERNELS = 2    # fabgpu.h FABGPU_FLAG_TIME_KERNELS
FLAG_NYM_NO_SIDE_STREAM = 128  # fabgpu.h FABGPU_FLAG_NYM_NO_SIDE_STREAM (idemix four-lane form: fixed-base terms inside the commitment kernel)
FLAG_NYM_FUSED_HASH = 64  # fabgpu.h FABGPU_FLAG_NYM_FUSED_HASH (idemix four-lane form: the challenge hashed inside the same kernel, round 4's form)
FLAG_NO_QUAD = 4         # fabgpu.h FABGPU_FLAG_NO_QUAD (idemix: never the four-lanes-per-signature kernel)
FLAG_PAIR_TABLE_LDS = 8      # fabgpu.h: the verify-only pair kernel keeps its per-signature table in LDS
FLAG_PAIR_TABLE_GLOBAL = 16  # ... in the global workspace
FLAG_KEY_TABLES_16BIT = 256  # fabgpu.h: registered keys also get a 16-bit comb (80 MiB each, built behind the registration)
FLAG_NO_WIDE = 32            # fabgpu.h: registered keys never on the eight-lanes-per-signature two-phase kernels (launches <= 8 192 signatures)
ST_VALID, ST_BAD_MATH, ST_HIGH_S, ST_RANGE, ST_OFF_CURVE = 0, 1, 2, 3, 4

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t


class FabgpuError(RuntimeError):
    """Infrastructure failure (non-zero FABGPU_E*): the Go provider would fall back to bccsp/sw."""


class BCCSPError(Exception):
    """A non-nil Go `error` from the provider (same text as bccsp/sw)."""


class _IdBatch(ctypes.Structure):
    """fabgpu_identity_batch (include/fabgpu.h)."""
    _fields_ = [("n", ctypes.c_size_t), ("arena", ctypes.c_void_p), ("arena_bytes", ctypes.c_size_t), ("off", ctypes.c_void_p),
                ("n_prefixes", ctypes.c_uint32), ("pre_off", ctypes.c_void_p), ("pre_idx", ctypes.c_void_p), ("qx", ctypes.c_void_p),
                ("qy", ctypes.c_void_p), ("key_id", ctypes.c_void_p), ("r", ctypes.c_void_p), ("s", ctypes.c_void_p),
                ("verdict_bits", ctypes.c_void_p), ("status", ctypes.c_void_p), ("flags", ctypes.c_uint32),
                ("n_gather", ctypes.c_uint32), ("gather_spans", ctypes.c_void_p), ("gather_digests", ctypes.c_void_p),
                ("gather_off", ctypes.c_void_p), ("gather_scratch", ctypes.c_void_p), ("gather_scratch_bytes", ctypes.c_size_t),
                ("stage_token", ctypes.c_uint64), ("tail", ctypes.c_void_p), ("tail_base", ctypes.c_uint32), ("tail_len", ctypes.c_uint32),
                ("digests", ctypes.c_void_p),
                ("n_nym", ctypes.c_uint32), ("nym_off", ctypes.c_void_p), ("nym_issuer", ctypes.c_void_p), ("nym_fields", ctypes.c_void_p),
                ("nym_verdict_bits", ctypes.c_void_p), ("nym_status", ctypes.c_void_p)]


class _BlockPass(ctypes.Structure):
    """fabgpu_block_pass (include/fabgpu_bccsp.h)."""
    _fields_ = [("block", ctypes.c_void_p), ("len", ctypes.c_size_t), ("block_seq", ctypes.c_uint64), ("flags", ctypes.c_uint32),
                ("cap_tx", ctypes.c_uint32), ("cap_tuples", ctypes.c_uint32),
                ("n_tx", ctypes.c_uint32), ("n_tuples", ctypes.c_uint32), ("n_block_sigs", ctypes.c_uint32), ("memo_seeded", ctypes.c_uint32),
                ("tail_base", ctypes.c_uint32), ("tail_len", ctypes.c_uint32), ("block_sigs_understood", ctypes.c_uint8),
                ("tx_flags", ctypes.c_void_p), ("tx_type", ctypes.c_void_p), ("tuple_tx", ctypes.c_void_p), ("tuple_kind", ctypes.c_void_p),
                ("tuple_status", ctypes.c_void_p), ("tuple_spans", ctypes.c_void_p), ("tuple_digest", ctypes.c_void_p),
                ("tuple_hashed", ctypes.c_void_p), ("tuple_qxy", ctypes.c_void_p), ("tail", ctypes.c_void_p), ("tail_cap", ctypes.c_uint32),
                ("n_keyed", ctypes.c_uint32), ("n_device_decoded", ctypes.c_uint32), ("ms_stage", ctypes.c_float * 4), ("device_context", ctypes.c_int32)]


class _CspOpts(ctypes.Structure):
    """fabgpu_csp_opts (include/fabgpu_bccsp.h): the `GPU:` section of the BCCSP configuration."""
    _fields_ = [("size", ctypes.c_uint32), ("n_devices", ctypes.c_int32), ("devices", ctypes.POINTER(ctypes.c_int32)), ("ctx_flags", ctypes.c_uint32),
                ("concurrent_passes", ctypes.c_uint32), ("expect_block_bytes", ctypes.c_uint64), ("expect_tuples", ctypes.c_uint32),
                ("pass_device_walk", ctypes.c_int32), ("pass_stage_min_bytes", ctypes.c_int64), ("pass_device_memo", ctypes.c_int32),
                ("pass_host_counts", ctypes.c_int32), ("pass_timing", ctypes.c_int32), ("pass_hash_memo", ctypes.c_int32),
                ("hash_memo_blocks", ctypes.c_uint32)]


class _Cfg(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("max_batch", ctypes.c_uint32), ("max_arena", ctypes.c_uint32),
                ("flags", ctypes.c_uint32)]


# every symbol include/fabgpu.h and include/fabgpu_bccsp.h declare (tests check the export list)
ABI_SYMBOLS = [
    "fabgpu_init", "fabgpu_shutdown", "fabgpu_device_count", "fabgpu_strerror", "fabgpu_abi_version",
    "fabgpu_p256_verify_batch", "fabgpu_sha256_batch", "fabgpu_sha256_p256_verify_batch",
    "fabgpu_p256_verify_batch_dev", "fabgpu_sha256_batch_dev", "fabgpu_sha256_p256_verify_batch_dev",
    "fabgpu_p256_key_register", "fabgpu_p256_key_lookup", "fabgpu_p256_key_count", "fabgpu_p256_verify_batch_keyed", "fabgpu_p256_verify_batch_keyed_dev",
    "fabgpu_sha256_p256_verify_batch_keyed", "fabgpu_sha256_p256_verify_batch_keyed_dev",
    "fabgpu_identity_verify_batch", "fabgpu_identity_verify_batch_dev", "fabgpu_arena_stage",
    "fabgpu_idemix_issuer_register", "fabgpu_idemix_issuer_count", "fabgpu_idemix_nym_verify_batch", "fabgpu_idemix_nym_verify_batch_dev",
    "fabgpu_bn256_g1_on_curve",
    "fabgpu_ecdsa_unmarshal_signature", "fabgpu_ecdsa_is_low_s",
    "fabgpu_p256_pubkey_on_curve", "fabgpu_hash_to_int",
    "fabgpu_csp_new", "fabgpu_csp_free", "fabgpu_csp_ctx", "fabgpu_csp_key_import", "fabgpu_csp_hash", "fabgpu_csp_verify",
    "fabgpu_csp_verify_batch", "fabgpu_csp_identity_verify_batch", "fabgpu_csp_block_preverify", "fabgpu_block_parse", "fabgpu_x509_p256_pubkey",
    "fabgpu_csp_idemix_issuer_import", "fabgpu_csp_idemix_nym_verify_batch", "fabgpu_csp_idemix_msp_register", "fabgpu_csp_idemix_msp_register2", "fabgpu_block_hash_checks",
    "fabgpu_block_tuples", "fabgpu_csp_block_preverify2", "fabgpu_csp_block_pass_abandon", "fabgpu_idemix_issuer_key_is_canonical", "fabgpu_csp_memo_lookup", "fabgpu_csp_memo_lookup_nym", "fabgpu_csp_memo_has_block", "fabgpu_csp_memo_evict_block",
    "fabgpu_csp_verify_coalesced", "fabgpu_csp_identity_verify_coalesced", "fabgpu_csp_coalescer_configure", "fabgpu_csp_coalescer_stats",
    "fabgpu_csp_memo_stats", "fabgpu_csp_memo_set_capacity", "fabgpu_csp_identity_cache_limits", "fabgpu_csp_identity_cache_size",
    "fabgpu_multi_init", "fabgpu_multi_shutdown", "fabgpu_multi_device_count", "fabgpu_multi_p256_verify_batch",
    "fabgpu_multi_sha256_p256_verify_batch", "fabgpu_multi_plan", "fabgpu_multi_merged_bitmap_dev", "fabgpu_multi_collective",
    "fabgpu_csp_pass_routes",
    "fabgpu_identity_to_p256", "fabgpu_csp_pass_stats",
    "fabgpu_p256_key_register_many", "fabgpu_csp_new2", "fabgpu_csp_device_count", "fabgpu_csp_ctx_of", "fabgpu_csp_passes_per_device",
    "fabgpu_csp_route_block", "fabgpu_csp_set_option", "fabgpu_csp_get_option",
    "fabgpu_csp_hash_lookup", "fabgpu_csp_hash_memo_stats",
]

# what libfabgpu_testhooks.so exports (fabric-mod_amd/csrc/fabgpu_testhooks.h): probes, walker comparisons, the synthetic block generator, the
# kernel timer - test and bench infrastructure that is NOT part of the product's C ABI and not in libfabgpu.so
HOOK_SYMBOLS = [
    "fabgpu_synth_batch", "fabgpu_last_kernel_ms", "fabgpu_csp_block_walk_compare", "fabgpu_block_walk_twopass_compare", "fabgpu_gate_sig_fast",
    "fabgpu_gate_sig_any", "fabgpu_csp_idfix_probe", "fabgpu_csp_gate_probe", "fabgpu_identity_table_hash", "fabgpu_test_nym_side_after",
    "fabgpu_test_key_table", "fabgpu_test_key_table_host", "fabgpu_test_gtab_compare_with_host", "fabgpu_test_key_tables16",
]
_HOOKS_PATH = os.path.join(os.path.dirname(_LIB_PATH), "libfabgpu_testhooks.so")

_lib = None
_hooks = None


def hooks_path() -> str:
    return _HOOKS_PATH


def load_hooks():
    """libfabgpu_testhooks.so, next to (and linked against) the product library; tests, bench.py and tools only."""
    global _hooks
    if _hooks is not None:
        return _hooks
    load()                                               # the product library first: the hooks resolve against it
    if not os.path.exists(_HOOKS_PATH):
        raise FabgpuError("libfabgpu_testhooks.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % _HOOKS_PATH)
    H = ctypes.CDLL(_HOOKS_PATH)
    H.fabgpu_last_kernel_ms.argtypes = [_vp]
    H.fabgpu_last_kernel_ms.restype = ctypes.c_float
    H.fabgpu_test_nym_side_after.argtypes = [_vp, ctypes.c_int]
    H.fabgpu_test_nym_side_after.restype = None
    H.fabgpu_test_key_table.argtypes = [_vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_int32), _sz]
    H.fabgpu_test_key_table_host.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), _sz]
    H.fabgpu_test_gtab_compare_with_host.argtypes = [_vp]
    H.fabgpu_test_gtab_compare_with_host.restype = ctypes.c_longlong
    H.fabgpu_test_key_tables16.argtypes = [_vp, ctypes.c_uint32]
    H.fabgpu_test_key_tables16.restype = ctypes.c_longlong
    H.fabgpu_csp_block_walk_compare.argtypes = [_vp, _u8p, _sz, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, _sz]
    H.fabgpu_block_walk_twopass_compare.argtypes = [_u8p, _sz, ctypes.c_char_p, _sz]
    H.fabgpu_gate_sig_fast.argtypes = [ctypes.c_char_p, _sz, ctypes.c_char_p, ctypes.c_char_p]
    H.fabgpu_csp_gate_probe.argtypes = [_vp, ctypes.c_uint32, _u8p, _sz, _u32p, _u8p, _u8p, _u8p]
    H.fabgpu_gate_sig_any.argtypes = [ctypes.c_char_p, _sz, ctypes.c_char_p, ctypes.c_char_p]
    H.fabgpu_csp_idfix_probe.argtypes = [_vp, ctypes.c_uint32, _u8p, _sz, _u32p, _u8p, _u8p]
    H.fabgpu_identity_table_hash.argtypes = [ctypes.c_char_p, _sz]
    H.fabgpu_identity_table_hash.restype = ctypes.c_uint64
    H.fabgpu_synth_batch.argtypes = [_sz, ctypes.c_uint64, ctypes.c_uint32, _u8p, _u8p, _u8p, _u8p, _u8p, _u8p, _u8p, ctypes.c_int]
    _hooks = H
    return H


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libfabgpu.so (built in-tree by __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise FabgpuError("libfabgpu.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % _LIB_PATH)
    L = ctypes.CDLL(_LIB_PATH)
    L.fabgpu_strerror.restype = ctypes.c_char_p
    L.fabgpu_strerror.argtypes = [ctypes.c_int]
    L.fabgpu_init.argtypes = [ctypes.POINTER(_Cfg), ctypes.POINTER(_vp)]
    L.fabgpu_shutdown.argtypes = [_vp]
    L.fabgpu_shutdown.restype = None
    L.fabgpu_device_count.argtypes = [_vp]
    L.fabgpu_p256_verify_batch.argtypes = [_vp, _sz, _u8p, _u8p, _u8p, _u8p, _u8p, _u64p, _u8p]
    L.fabgpu_sha256_batch.argtypes = [_vp, _sz, _u8p, _u32p, _u8p]
    L.fabgpu_sha256_p256_verify_batch.argtypes = [_vp, _sz, _u8p, _u32p, _u8p, _u8p, _u8p, _u8p, _u64p, _u8p]
    L.fabgpu_p256_verify_batch_dev.argtypes = [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.fabgpu_sha256_batch_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _vp, _vp]
    L.fabgpu_sha256_p256_verify_batch_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.fabgpu_p256_key_register.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, _u32p]
    L.fabgpu_p256_key_lookup.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, _u32p]
    L.fabgpu_p256_key_count.argtypes = [_vp]
    L.fabgpu_p256_verify_batch_keyed.argtypes = [_vp, _sz, _u32p, _u8p, _u8p, _u8p, _u64p, _u8p]
    L.fabgpu_p256_verify_batch_keyed_dev.argtypes = [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.fabgpu_sha256_p256_verify_batch_keyed.argtypes = [_vp, _sz, _u8p, _u32p, _u32p, _u8p, _u8p, _u64p, _u8p]
    L.fabgpu_sha256_p256_verify_batch_keyed_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.fabgpu_arena_stage.argtypes = [_vp, _u8p, _sz, ctypes.POINTER(ctypes.c_uint64)]
    L.fabgpu_identity_verify_batch.argtypes = [_vp, ctypes.POINTER(_IdBatch)]
    L.fabgpu_identity_verify_batch_dev.argtypes = [_vp, ctypes.POINTER(_IdBatch), _vp, _vp]
    L.fabgpu_idemix_issuer_register.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _u32p]
    L.fabgpu_idemix_issuer_count.argtypes = [_vp]
    L.fabgpu_idemix_nym_verify_batch.argtypes = [_vp, _sz, _u8p, _u32p, _u32p, _u8p, _u8p, _u8p, _u8p, _u8p, _u8p, _u64p, _u8p]
    L.fabgpu_idemix_nym_verify_batch_dev.argtypes = [_vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    L.fabgpu_bn256_g1_on_curve.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.fabgpu_ecdsa_unmarshal_signature.argtypes = [ctypes.c_char_p, _sz, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.fabgpu_ecdsa_is_low_s.argtypes = [ctypes.c_char_p]
    L.fabgpu_p256_pubkey_on_curve.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.fabgpu_hash_to_int.argtypes = [ctypes.c_char_p, _sz, ctypes.c_char_p]
    L.fabgpu_hash_to_int.restype = None
    L.fabgpu_csp_new.argtypes = [ctypes.POINTER(_Cfg), ctypes.POINTER(_vp), ctypes.c_char_p, _sz]
    L.fabgpu_csp_new2.argtypes = [ctypes.POINTER(_CspOpts), ctypes.POINTER(_vp), ctypes.c_char_p, _sz]
    L.fabgpu_csp_device_count.argtypes = [_vp]
    L.fabgpu_csp_ctx_of.argtypes = [_vp, ctypes.c_int]
    L.fabgpu_csp_ctx_of.restype = _vp
    L.fabgpu_csp_passes_per_device.argtypes = [_vp, _u64p, ctypes.c_int]
    L.fabgpu_csp_route_block.argtypes = [_vp, ctypes.c_uint64]
    L.fabgpu_csp_set_option.argtypes = [_vp, ctypes.c_char_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int64)]
    L.fabgpu_csp_get_option.argtypes = [_vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]
    L.fabgpu_p256_key_register_many.argtypes = [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, _u32p]
    L.fabgpu_csp_free.argtypes = [_vp]
    L.fabgpu_csp_free.restype = None
    L.fabgpu_csp_ctx.argtypes = [_vp]
    L.fabgpu_csp_ctx.restype = _vp
    L.fabgpu_csp_key_import.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, _sz]
    L.fabgpu_csp_hash.argtypes = [_vp, ctypes.c_char_p, _sz, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _sz]
    L.fabgpu_csp_verify.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _sz, ctypes.c_char_p, _sz,
                                    ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, _sz]
    L.fabgpu_csp_verify_batch.argtypes = [_vp, _sz, _u8p, _u8p, _u8p, _u32p, _u8p, _u32p, _u8p, ctypes.c_char_p, _sz]
    L.fabgpu_csp_verify_coalesced.argtypes = L.fabgpu_csp_verify.argtypes
    L.fabgpu_csp_identity_verify_coalesced.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _sz, ctypes.c_char_p, _sz, ctypes.c_char_p, _sz]
    L.fabgpu_csp_coalescer_configure.argtypes = [_vp, ctypes.c_uint32, ctypes.c_uint32]
    L.fabgpu_csp_coalescer_stats.argtypes = [_vp, _u64p, _u64p, _u64p]
    L.fabgpu_csp_identity_verify_batch.argtypes = [_vp, _sz, _u8p, _u8p, _u8p, _u32p, _u8p, _u32p, ctypes.c_char_p, _sz]
    L.fabgpu_csp_block_preverify.argtypes = [_vp, _u8p, _sz, _u32p, _u8p, _u8p, ctypes.c_uint32, _u32p, _u32p, _u8p, _u8p, ctypes.c_uint32]
    L.fabgpu_block_parse.argtypes = [_u8p, _sz, _u32p, _u32p, _u32p, _u8p, ctypes.c_uint32, ctypes.c_char_p, _sz]
    L.fabgpu_csp_idemix_msp_register.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, _sz, ctypes.POINTER(ctypes.c_int64)]
    L.fabgpu_csp_idemix_msp_register2.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _sz, ctypes.POINTER(ctypes.c_int64)]
    L.fabgpu_csp_idemix_issuer_import.argtypes = [_vp, ctypes.c_char_p, _sz, ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, _sz]
    L.fabgpu_csp_idemix_nym_verify_batch.argtypes = [_vp, ctypes.c_int64, _sz, _u8p, _u32p, _u8p, _u32p, _u8p, _u32p, _u8p, _u8p, ctypes.c_char_p, _sz]
    L.fabgpu_block_hash_checks.argtypes = [_u8p, _sz, ctypes.c_uint32, _u32p, _u32p, _u8p, _u32p, _u32p]
    L.fabgpu_csp_block_preverify2.argtypes = [_vp, ctypes.POINTER(_BlockPass)]
    L.fabgpu_csp_memo_lookup.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _sz, ctypes.c_char_p, _sz, _u8p]
    L.fabgpu_csp_memo_has_block.argtypes = [_vp, ctypes.c_uint64, _u64p]
    L.fabgpu_csp_memo_lookup_nym.argtypes = [_vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, _sz, ctypes.c_char_p, _sz, _u8p]
    L.fabgpu_csp_memo_evict_block.argtypes = [_vp, ctypes.c_uint64, _u64p]
    L.fabgpu_csp_memo_stats.argtypes = [_vp, _u64p, _u64p, _u64p, _u64p]
    L.fabgpu_csp_memo_set_capacity.argtypes = [_vp, ctypes.c_uint64]
    L.fabgpu_csp_hash_lookup.argtypes = [_vp, ctypes.c_char_p, _sz, ctypes.c_char_p]
    L.fabgpu_csp_hash_memo_stats.argtypes = [_vp, _u64p, _u64p, _u64p, _u64p, _u64p]
    L.fabgpu_csp_identity_cache_limits.argtypes = [_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
    L.fabgpu_csp_identity_cache_size.argtypes = [_vp, _u64p]
    L.fabgpu_csp_pass_routes.argtypes = [_vp, _u64p, _u64p, ctypes.c_char_p, _sz]
    L.fabgpu_identity_to_p256.argtypes = [ctypes.c_char_p, _sz, ctypes.c_char_p]
    L.fabgpu_csp_pass_stats.argtypes = [_vp, _u64p]
    L.fabgpu_multi_init.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_uint32, ctypes.POINTER(_vp)]
    L.fabgpu_multi_shutdown.argtypes = [_vp]
    L.fabgpu_multi_shutdown.restype = None
    L.fabgpu_multi_device_count.argtypes = [_vp]
    L.fabgpu_multi_p256_verify_batch.argtypes = [_vp, _sz, _u8p, _u8p, _u8p, _u8p, _u8p, _u64p, _u8p]
    L.fabgpu_multi_sha256_p256_verify_batch.argtypes = [_vp, _sz, _u8p, _u32p, _u8p, _u8p, _u8p, _u8p, _u64p, _u8p]
    L.fabgpu_multi_plan.argtypes = [_sz, _u32p, ctypes.c_uint32, _u64p, _u64p, _u64p]
    L.fabgpu_multi_merged_bitmap_dev.argtypes = [_vp, ctypes.c_int]
    L.fabgpu_multi_merged_bitmap_dev.restype = _vp
    L.fabgpu_multi_collective.argtypes = [_vp, ctypes.c_char_p, ctypes.c_size_t]
    L.fabgpu_block_tuples.argtypes = [_u8p, _sz, ctypes.c_uint32, _u32p, _u32p, _u8p, _u32p, _u8p, ctypes.c_uint32, _u32p, _u32p]
    L.fabgpu_x509_p256_pubkey.argtypes = [ctypes.c_char_p, _sz, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    _lib = L
    return L


def strerror(code: int) -> str:
    return load().fabgpu_strerror(code).decode()


def _check(rc: int, what: str):
    if rc != FABGPU_OK:
        raise FabgpuError("%s failed: %s (%d)" % (what, strerror(rc), rc))


def _a8(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p8(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_u8p)


def unpack_bits(words: np.ndarray, n: int) -> np.ndarray:
    """verdict bitmap (u64 words, bit i%64 of word i/64) -> bool[n]."""
    b = np.unpackbits(np.ascontiguousarray(words, dtype="<u8").view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


# ------------------------------------------------------------------------------------------------
# host gates (pure CPU)
# ------------------------------------------------------------------------------------------------
def unmarshal_ecdsa_signature(raw: bytes) -> Tuple[int, bytes, bytes, int]:
    """utils.UnmarshalECDSASignature (bccsp/utils/ecdsa.go:43-67): (rc, r32, s32, flags)."""
    r = ctypes.create_string_buffer(32)
    s = ctypes.create_string_buffer(32)
    fl = ctypes.c_int(0)
    rc = load().fabgpu_ecdsa_unmarshal_signature(raw, len(raw), r, s, ctypes.byref(fl))
    return rc, r.raw, s.raw, fl.value


def is_low_s(s32: bytes) -> bool:
    return bool(load().fabgpu_ecdsa_is_low_s(s32))


def pubkey_on_curve(qx32: bytes, qy32: bytes) -> bool:
    return bool(load().fabgpu_p256_pubkey_on_curve(qx32, qy32))


def bn256_g1_on_curve(x32: bytes, y32: bytes) -> bool:
    return bool(load().fabgpu_bn256_g1_on_curve(bytes(x32), bytes(y32)))


def hash_to_int(digest: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    load().fabgpu_hash_to_int(digest, len(digest), out)
    return out.raw


def synth_batch(n: int, seed: int = 20260921, invalid_permille: int = 0, e_in: Optional[np.ndarray] = None, threads: int = 0):
    """Synthetic tuples; see fabric-mod_amd/csrc/fabgpu_testhooks.h fabgpu_synth_batch (test-hook library)."""
    qx, qy, e, r, s = (np.zeros((n, 32), np.uint8) for _ in range(5))
    kind = np.zeros(n, np.uint8)
    ein = None if e_in is None else _a8(e_in)
    _check(load_hooks().fabgpu_synth_batch(n, seed, invalid_permille, _p8(ein), _p8(qx), _p8(qy), _p8(e), _p8(r), _p8(s), _p8(kind), threads),
           "fabgpu_synth_batch")
    return dict(qx=qx, qy=qy, e=e, r=r, s=s, kind=kind)


# ------------------------------------------------------------------------------------------------
# raw C-ABI context (what the cgo provider binds)
# ------------------------------------------------------------------------------------------------
class Context:
    def __init__(self, device: int = -1, max_batch: int = 0, max_arena: int = 0, flags: int = 0):
        L = load()
        cfg = _Cfg(device, max_batch, max_arena, flags)
        h = _vp()
        _check(L.fabgpu_init(ctypes.byref(cfg), ctypes.byref(h)), "fabgpu_init")
        self._h = h
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.fabgpu_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def device_count(self) -> int:
        return self._L.fabgpu_device_count(self._h)

    def p256_verify_batch(self, qx, qy, e, r, s, want_status=True):
        qx, qy, e, r, s = map(_a8, (qx, qy, e, r, s))
        n = qx.shape[0] if qx.ndim == 2 else qx.size // 32
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        _check(self._L.fabgpu_p256_verify_batch(self._h, n, _p8(qx), _p8(qy), _p8(e), _p8(r), _p8(s),
                                                 bits.ctypes.data_as(_u64p), _p8(st)), "fabgpu_p256_verify_batch")
        return unpack_bits(bits, n), st

    def sha256_batch(self, arena, off):
        arena = _a8(arena)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = off.size - 1
        out = np.zeros((n, 32), dtype=np.uint8)
        _check(self._L.fabgpu_sha256_batch(self._h, n, _p8(arena), off.ctypes.data_as(_u32p), _p8(out)), "fabgpu_sha256_batch")
        return out

    def sha256_p256_verify_batch(self, arena, off, qx, qy, r, s, want_status=True):
        arena, qx, qy, r, s = map(_a8, (arena, qx, qy, r, s))
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = off.size - 1
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        _check(self._L.fabgpu_sha256_p256_verify_batch(self._h, n, _p8(arena), off.ctypes.data_as(_u32p), _p8(qx), _p8(qy), _p8(r),
                                                        _p8(s), bits.ctypes.data_as(_u64p), _p8(st)), "fabgpu_sha256_p256_verify_batch")
        return unpack_bits(bits, n), st

    # device-resident variants: arguments are integer device addresses (torch tensor .data_ptr())
    def p256_verify_batch_dev(self, n, qx, qy, e, r, s, verdict_bits, status, stream=0):
        _check(self._L.fabgpu_p256_verify_batch_dev(self._h, n, qx, qy, e, r, s, verdict_bits, status or None, stream or None),
               "fabgpu_p256_verify_batch_dev")

    def sha256_batch_dev(self, n, arena, arena_bytes, off, digests, stream=0):
        _check(self._L.fabgpu_sha256_batch_dev(self._h, n, arena, arena_bytes, off, digests, stream or None), "fabgpu_sha256_batch_dev")

    def sha256_p256_verify_batch_dev(self, n, arena, arena_bytes, off, qx, qy, r, s, verdict_bits, status, stream=0):
        _check(self._L.fabgpu_sha256_p256_verify_batch_dev(self._h, n, arena, arena_bytes, off, qx, qy, r, s, verdict_bits,
                                                            status or None, stream or None), "fabgpu_sha256_p256_verify_batch_dev")

    # registered public keys (bccsp.KeyImport): a comb table per key on the device, verification without doublings
    def key_register(self, qx32: bytes, qy32: bytes) -> int:
        kid = ctypes.c_uint32(0)
        _check(self._L.fabgpu_p256_key_register(self._h, bytes(qx32), bytes(qy32), ctypes.byref(kid)), "fabgpu_p256_key_register")
        return int(kid.value)

    def key_count(self) -> int:
        return self._L.fabgpu_p256_key_count(self._h)

    def p256_verify_batch_keyed(self, key_id, e, r, s, want_status=True):
        e, r, s = map(_a8, (e, r, s))
        key_id = np.ascontiguousarray(key_id, dtype=np.uint32)
        n = key_id.size
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        _check(self._L.fabgpu_p256_verify_batch_keyed(self._h, n, key_id.ctypes.data_as(_u32p), _p8(e), _p8(r), _p8(s),
                                                       bits.ctypes.data_as(_u64p), _p8(st)), "fabgpu_p256_verify_batch_keyed")
        return unpack_bits(bits, n), st

    def sha256_p256_verify_batch_keyed(self, arena, off, key_id, r, s, want_status=True):
        arena, r, s = map(_a8, (arena, r, s))
        off = np.ascontiguousarray(off, dtype=np.uint32)
        key_id = np.ascontiguousarray(key_id, dtype=np.uint32)
        n = off.size - 1
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        _check(self._L.fabgpu_sha256_p256_verify_batch_keyed(self._h, n, _p8(arena), off.ctypes.data_as(_u32p), key_id.ctypes.data_as(_u32p),
                                                              _p8(r), _p8(s), bits.ctypes.data_as(_u64p), _p8(st)),
               "fabgpu_sha256_p256_verify_batch_keyed")
        return unpack_bits(bits, n), st

    def sha256_p256_verify_batch_keyed_dev(self, n, arena, arena_bytes, off, key_id, r, s, verdict_bits, status, stream=0):
        _check(self._L.fabgpu_sha256_p256_verify_batch_keyed_dev(self._h, n, arena, arena_bytes, off, key_id, r, s, verdict_bits,
                                                                  status or None, stream or None), "fabgpu_sha256_p256_verify_batch_keyed_dev")

    # idemix pseudonym signatures (FP256BN): an issuer is registered once, batches name it by id
    def idemix_issuer_register(self, hsk_xy: Tuple[bytes, bytes], hrand_xy: Tuple[bytes, bytes], ipk_hash32: bytes) -> int:
        iid = ctypes.c_uint32(0)
        _check(self._L.fabgpu_idemix_issuer_register(self._h, bytes(hsk_xy[0]), bytes(hsk_xy[1]), bytes(hrand_xy[0]), bytes(hrand_xy[1]),
                                                      bytes(ipk_hash32), ctypes.byref(iid)), "fabgpu_idemix_issuer_register")
        return int(iid.value)

    def idemix_issuer_count(self) -> int:
        return self._L.fabgpu_idemix_issuer_count(self._h)

    def idemix_nym_verify_batch(self, arena, off, nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce, issuer_id=None, want_status=True):
        """NymSignature.Ver over a batch: message i = arena[off[i], off[i+1]); fields n x 32 big-endian bytes."""
        arena, nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce = map(_a8, (arena, nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce))
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = off.size - 1
        iid = None if issuer_id is None else np.ascontiguousarray(issuer_id, dtype=np.uint32)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        _check(self._L.fabgpu_idemix_nym_verify_batch(self._h, n, _p8(arena), off.ctypes.data_as(_u32p),
                                                       iid.ctypes.data_as(_u32p) if iid is not None else None, _p8(nym_x), _p8(nym_y),
                                                       _p8(proof_c), _p8(proof_s_sk), _p8(proof_s_r_nym), _p8(nonce),
                                                       bits.ctypes.data_as(_u64p), _p8(st)), "fabgpu_idemix_nym_verify_batch")
        return unpack_bits(bits, n), st

    def idemix_nym_verify_batch_dev(self, n, arena, arena_bytes, off, issuer_id, nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce,
                                    verdict_bits, status, stream=0):
        _check(self._L.fabgpu_idemix_nym_verify_batch_dev(self._h, n, arena, arena_bytes, off, issuer_id or None, nym_x, nym_y, proof_c,
                                                           proof_s_sk, proof_s_r_nym, nonce, verdict_bits, status or None, stream or None),
               "fabgpu_idemix_nym_verify_batch_dev")

    def arena_stage(self, arena) -> int:
        """fabgpu_arena_stage: upload ahead of the batch that refers to the bytes; returns the token."""
        arena = _a8(arena)
        tok = ctypes.c_uint64(0)
        _check(self._L.fabgpu_arena_stage(self._h, _p8(arena), arena.size, ctypes.byref(tok)), "fabgpu_arena_stage")
        return int(tok.value)

    def identity_verify_batch(self, arena, off, r, s, qx=None, qy=None, key_id=None, pre_off=None, pre_idx=None, want_status=True, spans=False,
                              gather_spans=None, stage_token=0, want_digests=False, tail=None, tail_base=0, nym=None):
        """fabgpu_identity_verify_batch: message i = [prefix pre_idx[i]] || arena[off[i], off[i+1]); keys by value or by id.
        gather_spans (m x 6 u32: three (start, end) pieces per gathered message): also returns their m x 32 digest bytes.
        want_digests: also returns (last) the n x 32 message digests as the fused kernel computed them.  tail / tail_base: bytes that
        are not in the arena but addressed at offsets >= tail_base (spans mode).  nym: pseudonym signatures over messages of the same
        arena (spans mode), verified on a second stream in the same submission; returns (verdicts, statuses) of those last."""
        arena, r, s = map(_a8, (arena, r, s))
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = off.size // 2 if spans else off.size - 1
        keep = [arena, off, r, s]
        b = _IdBatch()
        b.n = n
        b.flags = (1 if spans else 0) | (2 if stage_token else 0)
        b.stage_token = stage_token
        b.arena, b.off, b.r, b.s = arena.ctypes.data, off.ctypes.data, r.ctypes.data, s.ctypes.data
        if key_id is not None:
            key_id = np.ascontiguousarray(key_id, dtype=np.uint32); keep.append(key_id)
            b.key_id = key_id.ctypes.data
        else:
            qx, qy = _a8(qx), _a8(qy); keep += [qx, qy]
            b.qx, b.qy = qx.ctypes.data, qy.ctypes.data
        if pre_idx is not None:
            pre_off = np.ascontiguousarray(pre_off, dtype=np.uint32); pre_idx = np.ascontiguousarray(pre_idx, dtype=np.uint32)
            keep += [pre_off, pre_idx]
            b.n_prefixes, b.pre_off, b.pre_idx = (pre_off.size // 2 if spans else pre_off.size - 1), pre_off.ctypes.data, pre_idx.ctypes.data
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        b.verdict_bits = bits.ctypes.data
        b.status = st.ctypes.data if want_status else None
        mdig = None
        if want_digests:
            mdig = np.zeros((n, 32), dtype=np.uint8)
            keep.append(mdig)
            b.digests = mdig.ctypes.data
        if tail is not None:
            tail = _a8(tail)
            keep.append(tail)
            b.tail, b.tail_base, b.tail_len = tail.ctypes.data, tail_base, tail.size
        dig = None
        if gather_spans is not None:
            gather_spans = np.ascontiguousarray(gather_spans, dtype=np.uint32).reshape(-1, 6)
            dig = np.zeros((gather_spans.shape[0], 32), dtype=np.uint8)
            keep += [gather_spans, dig]
            b.n_gather, b.gather_spans, b.gather_digests = gather_spans.shape[0], gather_spans.ctypes.data, dig.ctypes.data
        nres = ()
        if nym is not None:   # (spans m x 2, issuer ids m, [nym_x, nym_y, proof_c, proof_s_sk, proof_s_r_nym, nonce] each m x 32): pseudonym signatures riding along
            nsp = np.ascontiguousarray(nym[0], dtype=np.uint32).reshape(-1, 2)
            niss = np.ascontiguousarray(nym[1], dtype=np.uint32)
            nf = np.ascontiguousarray(np.concatenate([_a8(c).reshape(-1, 32) for c in nym[2]], axis=0))
            m = nsp.shape[0]
            nbits, nst = np.zeros((m + 63) // 64, dtype=np.uint64), np.zeros(m, dtype=np.uint8)
            keep += [nsp, niss, nf, nbits, nst]
            b.n_nym, b.nym_off, b.nym_issuer, b.nym_fields = m, nsp.ctypes.data, niss.ctypes.data, nf.ctypes.data
            b.nym_verdict_bits, b.nym_status = nbits.ctypes.data, nst.ctypes.data
        _check(self._L.fabgpu_identity_verify_batch(self._h, ctypes.byref(b)), "fabgpu_identity_verify_batch")
        if nym is not None:
            nres = ((unpack_bits(nbits, m), nst),)
        res = (unpack_bits(bits, n), st) + ((dig,) if dig is not None else ()) + ((mdig,) if mdig is not None else ()) + nres
        return res

    def identity_verify_batch_dev(self, desc: "_IdBatch", mid_scratch, stream=0):
        _check(self._L.fabgpu_identity_verify_batch_dev(self._h, ctypes.byref(desc), mid_scratch or None, stream or None),
               "fabgpu_identity_verify_batch_dev")

    def p256_verify_batch_keyed_dev(self, n, key_id, e, r, s, verdict_bits, status, stream=0):
        _check(self._L.fabgpu_p256_verify_batch_keyed_dev(self._h, n, key_id, e, r, s, verdict_bits, status or None, stream or None),
               "fabgpu_p256_verify_batch_keyed_dev")

    KEY_TABLE_WORDS = 32 * 256 * 20

    def test_key_table(self, key_id: int) -> np.ndarray:
        """TEST HOOK: the comb table of a registered key as it lies on the device (int32[32 * 256 * 20])."""
        out = np.zeros(self.KEY_TABLE_WORDS, dtype=np.int32)
        _check(load_hooks().fabgpu_test_key_table(self._h, key_id, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), out.size), "fabgpu_test_key_table")
        return out

    @staticmethod
    def test_key_table_host(qx32: bytes, qy32: bytes) -> np.ndarray:
        """TEST HOOK: the same table as the host builder makes it."""
        out = np.zeros(Context.KEY_TABLE_WORDS, dtype=np.int32)
        _check(load_hooks().fabgpu_test_key_table_host(qx32, qy32, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), out.size), "fabgpu_test_key_table_host")
        return out

    def test_gtab_compare_with_host(self) -> int:
        """TEST HOOK: first differing word of the device-built generator comb against the host builder's table; -1 = identical."""
        return int(load_hooks().fabgpu_test_gtab_compare_with_host(self._h))

    def test_key_tables16(self, key_id: int) -> int:
        """TEST HOOK (FLAG_KEY_TABLES_16BIT): waits for the queued 16-bit key tables; their number, after key_id's was cross-checked against its
        8-bit table (-1: the key has none; -2: error; -(1000 + w): window w disagrees)."""
        return int(load_hooks().fabgpu_test_key_tables16(self._h, key_id))

    def test_nym_side_after(self, on: bool) -> None:
        """TEST HOOK (libfabgpu_testhooks.so): the idemix side launch behind the commitment launch while on."""
        load_hooks().fabgpu_test_nym_side_after(self._h, 1 if on else 0)

    def last_kernel_ms(self) -> float:
        return float(load_hooks().fabgpu_last_kernel_ms(self._h))


MULTI_HOST_MERGE = 1


def multi_plan(n: int, n_devices: int, off=None):
    """fabgpu_multi_plan: ([(lo, hi)] per device, words_per_rank) - by count, or by message bytes when off (n + 1 offsets) is given."""
    lo, hi = np.zeros(n_devices, np.uint64), np.zeros(n_devices, np.uint64)
    wpr = ctypes.c_uint64(0)
    o = None
    if off is not None:
        off = np.ascontiguousarray(off, dtype=np.uint32)
        o = off.ctypes.data_as(_u32p)
    _check(load().fabgpu_multi_plan(n, o, n_devices, lo.ctypes.data_as(_u64p), hi.ctypes.data_as(_u64p), ctypes.byref(wpr)), "fabgpu_multi_plan")
    return [(int(a), int(b)) for a, b in zip(lo, hi)], int(wpr.value)


class MultiContext:
    """fabgpu_multi_*: one batch cut over the GPUs of the node, RCCL all-gather of the verdict bitmaps (SURVEY 8(e), configs[2])."""

    def __init__(self, devices: Sequence[int], host_merge: bool = False, selfcheck_seconds: int = 0):
        """selfcheck_seconds: FABGPU_MULTI_SELFCHECK_SECONDS - the deadline of the init-time all-gather self-check (0: the library's 10 s)"""
        self._L = load()
        self._h = _vp()
        d = (ctypes.c_int32 * len(devices))(*devices)
        flags = (MULTI_HOST_MERGE if host_merge else 0) | ((int(selfcheck_seconds) & 0xFF) << 8)
        rc = self._L.fabgpu_multi_init(d, len(devices), flags, ctypes.byref(self._h))
        if rc != FABGPU_OK:
            raise FabgpuError("fabgpu_multi_init failed: %s (%d)" % (strerror(rc), rc))

    def close(self):
        if self._h:
            self._L.fabgpu_multi_shutdown(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_count(self) -> int:
        return self._L.fabgpu_multi_device_count(self._h)

    def collective(self):
        """(rccl_ranks, why): rccl_ranks = G when the shard bitmaps are merged by ncclAllGather (self-checked at init), 0 when the host
        merges them; why = the reason the library gives (fabgpu_multi_collective)"""
        buf = ctypes.create_string_buffer(256)
        r = self._L.fabgpu_multi_collective(self._h, buf, len(buf))
        if r < 0:
            raise FabgpuError("fabgpu_multi_collective: %s (%d)" % (strerror(r), r))
        return r, buf.value.decode("utf-8", "replace")

    def p256_verify_batch(self, qx, qy, e, r, s, want_status=True):
        qx, qy, e, r, s = map(_a8, (qx, qy, e, r, s))
        n = qx.size // 32
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        _check(self._L.fabgpu_multi_p256_verify_batch(self._h, n, _p8(qx), _p8(qy), _p8(e), _p8(r), _p8(s), bits.ctypes.data_as(_u64p), _p8(st)),
               "fabgpu_multi_p256_verify_batch")
        return unpack_bits(bits, n), st

    def sha256_p256_verify_batch(self, arena, off, qx, qy, r, s, want_status=True):
        arena, qx, qy, r, s = map(_a8, (arena, qx, qy, r, s))
        off = np.ascontiguousarray(off, dtype=np.uint32)
        n = off.size - 1
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        st = np.zeros(n, dtype=np.uint8) if want_status else None
        _check(self._L.fabgpu_multi_sha256_p256_verify_batch(self._h, n, _p8(arena), off.ctypes.data_as(_u32p), _p8(qx), _p8(qy), _p8(r), _p8(s),
                                                             bits.ctypes.data_as(_u64p), _p8(st)), "fabgpu_multi_sha256_p256_verify_batch")
        return unpack_bits(bits, n), st

    def merged_bitmap_dev(self, g: int) -> int:
        return int(self._L.fabgpu_multi_merged_bitmap_dev(self._h, g) or 0)


# ------------------------------------------------------------------------------------------------
# the reference's vocabulary
# ------------------------------------------------------------------------------------------------
class ECDSAPublicKey:
    """bccsp/sw/ecdsakey.go:72-117 (X, Y of an ecdsa.PublicKey on P-256)."""

    def __init__(self, x: int, y: int):
        self.x, self.y = x, y

    def xy_bytes(self) -> Tuple[bytes, bytes]:
        return self.x.to_bytes(32, "big"), self.y.to_bytes(32, "big")


class SHA256Opts:
    """bccsp/hashopts.go:20-70"""
    algorithm = "SHA256"


class SHA3_256Opts:
    algorithm = "SHA3_256"


class GPUCSP:
    """The accelerated verbs of bccsp.BCCSP (bccsp/bccsp.go:90-134); everything else the Go provider delegates to bccsp/sw."""

    def __init__(self, device: int = -1, devices: Optional[Sequence[int]] = None, flags: int = 0, concurrent_passes: int = 0,
                 expect_block_bytes: int = 0, expect_tuples: int = 0, **switches):
        """device: ONE context on that HIP ordinal (fabgpu_csp_new).  devices: one context per entry - an ordinal may repeat; an empty
        list means every visible device - behind ONE provider (fabgpu_csp_new2: what bccsp/factory builds from the `GPU:` section).
        switches: pass_device_walk / pass_stage_min_bytes / pass_device_memo / pass_host_counts / pass_timing / pass_hash_memo (0 default,
        > 0 on, < 0 off), hash_memo_blocks (host copies of blocks the digest memo keeps per device)."""
        L = load()
        h = _vp()
        err = ctypes.create_string_buffer(512)
        if devices is None and not (flags or concurrent_passes or switches):
            cfg = _Cfg(device, 0, 0, 0)
            rc = L.fabgpu_csp_new(ctypes.byref(cfg), ctypes.byref(h), err, 512)
        else:
            devs = [device] if devices is None else list(devices)
            arr = (ctypes.c_int32 * max(1, len(devs)))(*devs)
            o = _CspOpts()
            o.size, o.n_devices, o.devices = ctypes.sizeof(_CspOpts), len(devs), arr
            o.ctx_flags, o.concurrent_passes, o.expect_block_bytes, o.expect_tuples = flags, concurrent_passes, expect_block_bytes, expect_tuples
            for k, v in switches.items():
                if not hasattr(o, k):
                    raise TypeError("unknown provider switch %r" % k)
                setattr(o, k, v)
            rc = L.fabgpu_csp_new2(ctypes.byref(o), ctypes.byref(h), err, 512)
        if rc != FABGPU_OK:
            raise FabgpuError(err.value.decode() or strerror(rc))
        self._h, self._L = h, L

    def device_count(self) -> int:
        """Device contexts behind this provider."""
        return self._L.fabgpu_csp_device_count(self._h)

    def passes_per_device(self) -> List[int]:
        """Block passes each context of the pool has served."""
        v = (ctypes.c_uint64 * 64)()
        n = self._L.fabgpu_csp_passes_per_device(self._h, v, 64)
        if n < 0:
            raise FabgpuError("fabgpu_csp_passes_per_device: %s" % strerror(n))
        return [int(v[i]) for i in range(n)]

    def route_block(self, block_seq: int) -> int:
        return self._L.fabgpu_csp_route_block(self._h, block_seq)

    def set_option(self, name: str, value: int) -> int:
        """A switch of the living provider (0 default, > 0 on / threshold, < 0 off); returns the previous value."""
        prev = ctypes.c_int64(0)
        _check(self._L.fabgpu_csp_set_option(self._h, name.encode(), int(value), ctypes.byref(prev)), "fabgpu_csp_set_option(%s)" % name)
        return int(prev.value)

    def get_option(self, name: str) -> int:
        v = ctypes.c_int64(0)
        _check(self._L.fabgpu_csp_get_option(self._h, name.encode(), ctypes.byref(v)), "fabgpu_csp_get_option(%s)" % name)
        return int(v.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.fabgpu_csp_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def key_count(self, d: int = 0) -> int:
        """Number of public keys whose comb table is resident on device context d of the pool."""
        return self._L.fabgpu_p256_key_count(self._L.fabgpu_csp_ctx_of(self._h, d))

    def key_import(self, raw, opts=None) -> ECDSAPublicKey:
        """KeyImport(raw, &bccsp.ECDSAGoPublicKeyImportOpts{}) (bccsp/sw/keyimport.go:103-112): raw = (X, Y)."""
        if raw is None:
            raise BCCSPError("Invalid raw. It must not be nil.")
        k = ECDSAPublicKey(int(raw[0]), int(raw[1]))
        if 0 <= k.x < 1 << 256 and 0 <= k.y < 1 << 256:      # the provider registers the key's comb table on the device
            err = ctypes.create_string_buffer(256)
            _check(self._L.fabgpu_csp_key_import(self._h, k.x.to_bytes(32, "big"), k.y.to_bytes(32, "big"), None, err, 256), "fabgpu_csp_key_import")
        return k

    def hash(self, msg: Optional[bytes], opts) -> bytes:
        """CSP.Hash (bccsp/sw/impl.go:177-194)."""
        out = ctypes.create_string_buffer(32)
        err = ctypes.create_string_buffer(512)
        alg = None if opts is None else opts.algorithm.encode()
        msg = msg or b""
        _check(self._L.fabgpu_csp_hash(self._h, msg, len(msg), alg, out, err, 512), "fabgpu_csp_hash")
        if err.value:
            raise BCCSPError(err.value.decode())
        return out.raw

    def verify(self, k: Optional[ECDSAPublicKey], signature: bytes, digest: bytes, opts=None) -> bool:
        """CSP.Verify (bccsp/sw/impl.go:247-270): returns valid, raises BCCSPError where Go returns (false, err)."""
        valid = ctypes.c_int(0)
        flags = ctypes.c_int(0)
        err = ctypes.create_string_buffer(1024)
        qx, qy = (None, None) if k is None else k.xy_bytes()
        _check(self._L.fabgpu_csp_verify(self._h, qx, qy, signature, len(signature), digest, len(digest), ctypes.byref(valid),
                                         ctypes.byref(flags), err, 1024), "fabgpu_csp_verify")
        if err.value:
            raise BCCSPError(err.value.decode())
        return bool(valid.value)

    def verify_coalesced(self, k: Optional[ECDSAPublicKey], signature: bytes, digest: bytes) -> bool:
        """CSP.Verify for callers arriving many at a time on their own threads (fabgpu_csp_verify_coalesced): blocking, same answers as
        verify(); calls in flight together share a launch.  ctypes releases the GIL for the duration of the call."""
        valid = ctypes.c_int(0)
        flags = ctypes.c_int(0)
        err = ctypes.create_string_buffer(1024)
        qx, qy = (None, None) if k is None else k.xy_bytes()
        _check(self._L.fabgpu_csp_verify_coalesced(self._h, qx, qy, signature, len(signature), digest, len(digest), ctypes.byref(valid),
                                                   ctypes.byref(flags), err, 1024), "fabgpu_csp_verify_coalesced")
        if err.value:
            raise BCCSPError(err.value.decode())
        return bool(valid.value)

    def identity_verify_coalesced(self, k: Optional[ECDSAPublicKey], msg: bytes, signature: bytes) -> Optional[str]:
        """identity.Verify(msg, sig) through the coalescer: None (nil) or the error text."""
        err = ctypes.create_string_buffer(1024)
        qx, qy = (None, None) if k is None else k.xy_bytes()
        _check(self._L.fabgpu_csp_identity_verify_coalesced(self._h, qx, qy, msg, len(msg), signature, len(signature), err, 1024),
               "fabgpu_csp_identity_verify_coalesced")
        return err.value.decode() or None

    def coalescer_configure(self, window_us: int = 50, max_batch: int = 32768) -> None:
        _check(self._L.fabgpu_csp_coalescer_configure(self._h, window_us, max_batch), "fabgpu_csp_coalescer_configure")

    def coalescer_stats(self):
        v = [ctypes.c_uint64(0) for _ in range(3)]
        _check(self._L.fabgpu_csp_coalescer_stats(self._h, *[ctypes.byref(x) for x in v]), "fabgpu_csp_coalescer_stats")
        return dict(calls=v[0].value, launches=v[1].value, largest_batch=v[2].value)

    def verify_batch(self, keys: Sequence[ECDSAPublicKey], sigs: Sequence[bytes], digests: Sequence[bytes]):
        """n independent CSP.Verify calls in one launch: list of (valid, error-text-or-None)."""
        n = len(keys)
        qx = np.frombuffer(b"".join(k.x.to_bytes(32, "big") for k in keys), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
        qy = np.frombuffer(b"".join(k.y.to_bytes(32, "big") for k in keys), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
        sa, so = _ragged(sigs)
        da, do = _ragged(digests)
        valid = np.zeros(n, np.uint8)
        stride = 512
        errs = ctypes.create_string_buffer(max(1, n * stride))
        _check(self._L.fabgpu_csp_verify_batch(self._h, n, _p8(qx), _p8(qy), _p8(sa), so.ctypes.data_as(_u32p), _p8(da),
                                               do.ctypes.data_as(_u32p), _p8(valid), errs, stride), "fabgpu_csp_verify_batch")
        out = []
        for i in range(n):
            e = errs.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode()
            out.append((bool(valid[i]), e or None))
        return out

    def identity_verify_batch(self, keys: Sequence[ECDSAPublicKey], msgs: Sequence[bytes], sigs: Sequence[bytes]) -> List[Optional[str]]:
        """identity.Verify (msp/identities.go:169-196) for n triples: None (nil) or the error text."""
        n = len(keys)
        qx = np.frombuffer(b"".join(k.x.to_bytes(32, "big") for k in keys), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
        qy = np.frombuffer(b"".join(k.y.to_bytes(32, "big") for k in keys), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
        ma, mo = _ragged(msgs)
        sa, so = _ragged(sigs)
        stride = 512
        errs = ctypes.create_string_buffer(max(1, n * stride))
        _check(self._L.fabgpu_csp_identity_verify_batch(self._h, n, _p8(qx), _p8(qy), _p8(ma), mo.ctypes.data_as(_u32p), _p8(sa),
                                                        so.ctypes.data_as(_u32p), errs, stride), "fabgpu_csp_identity_verify_batch")
        return [(errs.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() or None) for i in range(n)]

    # ---- idemix creator signatures (bccsp/idemix/handlers) ----
    def idemix_issuer_import(self, ipk_raw: bytes) -> int:
        """IssuerPublicKeyImporter.KeyImport for the accelerated part: registers HSk / HRand / Hash; -1 = not accelerated."""
        iid = ctypes.c_int64(-1)
        err = ctypes.create_string_buffer(512)
        _check(self._L.fabgpu_csp_idemix_issuer_import(self._h, bytes(ipk_raw), len(ipk_raw), ctypes.byref(iid), err, 512),
               "fabgpu_csp_idemix_issuer_import")
        if err.value:
            raise BCCSPError(err.value.decode())
        return int(iid.value)

    def idemix_msp_register(self, mspid: str, ipk_raw: bytes, channel: Optional[str] = None) -> int:
        """An idemix MSP of the channel: preverify_block then verifies its creators' pseudonym signatures too.  With a channel name the
        channel's latest key for the MSP id replaces its earlier one (rotation); the MSP id is ambiguous only while two channels disagree.
        -1: not accelerated (a key the device does not take, or one whose Hash field is not the hash of the rest of the key)."""
        iid = ctypes.c_int64(-1)
        if channel is None:
            _check(self._L.fabgpu_csp_idemix_msp_register(self._h, mspid.encode(), bytes(ipk_raw), len(ipk_raw), ctypes.byref(iid)),
                   "fabgpu_csp_idemix_msp_register")
        else:
            _check(self._L.fabgpu_csp_idemix_msp_register2(self._h, channel.encode(), mspid.encode(), bytes(ipk_raw), len(ipk_raw), ctypes.byref(iid)),
                   "fabgpu_csp_idemix_msp_register2")
        return int(iid.value)

    def idemix_nym_verify_batch(self, issuer_id: int, nym_keys: Sequence[bytes], sigs: Sequence[bytes], msgs: Sequence[bytes]):
        """NymVerifier.Verify for n tuples: list of (valid, needs_sw, error text or None)."""
        n = len(sigs)
        ka, ko = _ragged(nym_keys)
        sa, so = _ragged(sigs)
        ma, mo = _ragged(msgs)
        stride = 256
        valid = np.zeros(max(1, n), np.uint8)
        flags = np.zeros(max(1, n), np.uint8)
        errs = ctypes.create_string_buffer(max(1, n * stride))
        _check(self._L.fabgpu_csp_idemix_nym_verify_batch(self._h, issuer_id, n, _p8(ka), ko.ctypes.data_as(_u32p), _p8(sa), so.ctypes.data_as(_u32p),
                                                          _p8(ma), mo.ctypes.data_as(_u32p), _p8(valid), _p8(flags), errs, stride),
               "fabgpu_csp_idemix_nym_verify_batch")
        return [(bool(valid[i]), bool(flags[i] & 1), errs.raw[i * stride:(i + 1) * stride].split(b"\0", 1)[0].decode() or None) for i in range(n)]


def _ragged(items: Sequence[bytes]):
    off = np.zeros(len(items) + 1, dtype=np.uint32)
    if items:
        off[1:] = np.cumsum([len(x) for x in items])
    arena = np.frombuffer(b"".join(items) + b"\0", dtype=np.uint8).copy()
    return arena, off


class Identity:
    """msp.identity restricted to Verify (msp/identities.go:169-196)."""

    def __init__(self, csp: GPUCSP, pk: ECDSAPublicKey, hash_family: str = "SHA2"):
        self.csp, self.pk, self.hash_family = csp, pk, hash_family

    def verify(self, msg: bytes, sig: bytes) -> None:
        """Returns None (nil) or raises BCCSPError with identity.Verify's error text."""
        if self.hash_family != "SHA2":
            raise BCCSPError("hash familiy not recognized [%s]" % self.hash_family) if self.hash_family != "SHA3" else \
                BCCSPError("failed computing digest: SHA3 is served by bccsp/sw, not by the GPU provider")
        err = self.csp.identity_verify_batch([self.pk], [msg], [sig])[0]
        if err:
            raise BCCSPError(err)


TX_ALL_SIGNATURES_VALID, TX_BAD_CREATOR_SIGNATURE, TX_BAD_ENDORSEMENT, TX_NOT_UNDERSTOOD, TX_NEEDS_SW = 0, 1, 2, 3, 4
TX_BAD_TXID, TX_BAD_PROPOSAL_HASH = 5, 6
TUPLE_ST_BAD_DER, TUPLE_ST_NEEDS_SW, TUPLE_ST_EMPTY_SIG = 5, 6, 7


def block_parse(block: bytes):
    """Structure of a marshalled common.Block as the pre-verify pass sees it (pure host): dict(n_tx, n_tuples, n_prefixes,
    tx_type, channel_id)."""
    L = load()
    buf = np.frombuffer(block, dtype=np.uint8)
    n_tx, n_tup, n_pre = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_uint32(0)
    cap = 1 << 20
    tx_type = np.zeros(cap, dtype=np.uint8)
    ch = ctypes.create_string_buffer(256)
    rc = L.fabgpu_block_parse(_p8(buf), buf.size, ctypes.byref(n_tx), ctypes.byref(n_tup), ctypes.byref(n_pre), _p8(tx_type), cap, ch, 256)
    if rc != FABGPU_OK:
        raise FabgpuError("fabgpu_block_parse failed: %s (%d)" % (strerror(rc), rc))
    return dict(n_tx=n_tx.value, n_tuples=n_tup.value, n_prefixes=n_pre.value, tx_type=tx_type[:n_tx.value].copy(), channel_id=ch.value.decode(errors="replace"))


def block_hash_checks(block: bytes):
    """The TxID / proposal-hash checks the pre-verify pass derives from a block: list of (tx, kind, [(start, end)] * 3, (start, end))."""
    buf = np.frombuffer(block, dtype=np.uint8)
    cap = 4096
    while True:
        n = ctypes.c_uint32(0)
        tx, kind = np.zeros(cap, np.uint32), np.zeros(cap, np.uint8)
        sp, ex = np.zeros(cap * 6, np.uint32), np.zeros(cap * 2, np.uint32)
        rc = load().fabgpu_block_hash_checks(_p8(buf), buf.size, cap, ctypes.byref(n), tx.ctypes.data_as(_u32p), _p8(kind), sp.ctypes.data_as(_u32p),
                                             ex.ctypes.data_as(_u32p))
        if rc == -5:
            cap = n.value
            continue
        _check(rc, "fabgpu_block_hash_checks")
        m = n.value
        return [(int(tx[j]), int(kind[j]), [(int(sp[6 * j + 2 * p]), int(sp[6 * j + 2 * p + 1])) for p in range(3)], (int(ex[2 * j]), int(ex[2 * j + 1])))
                for j in range(m)]


TUPLE_CREATOR, TUPLE_ENDORSEMENT, TUPLE_BLOCK_SIG = 0, 1, 2
BLOCK_LEVEL_TX = 0xFFFFFFFF


def block_tuples(block: bytes):
    """Every (identity, message, signature) tuple the pre-verify pass derives from a marshalled block (pure host).
    Returns (tuples, virtual_arena): tuples = list of dict(tx, kind, identity, prefix, suffix, sig) with (start, length) spans into
    virtual_arena = block || zero padding || tail (the orderer block-signature messages the walker builds; block_prepass.h)."""
    buf = np.frombuffer(block, dtype=np.uint8)
    cap, tcap = 4096, 1 << 16
    while True:
        n, tl, tb = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_uint32(0)
        tx, kind, sp, tail = np.zeros(cap, np.uint32), np.zeros(cap, np.uint8), np.zeros(cap * 8, np.uint32), np.zeros(tcap, np.uint8)
        rc = load().fabgpu_block_tuples(_p8(buf), buf.size, cap, ctypes.byref(n), tx.ctypes.data_as(_u32p), _p8(kind), sp.ctypes.data_as(_u32p),
                                        _p8(tail), tcap, ctypes.byref(tl), ctypes.byref(tb))
        if rc == -5:
            cap, tcap = max(cap, n.value), max(tcap, tl.value)
            continue
        _check(rc, "fabgpu_block_tuples")
        arena = bytes(block) + b"\0" * (tb.value - len(block)) + bytes(tail[:tl.value])
        names = ("identity", "prefix", "suffix", "sig")
        out = []
        for i in range(n.value):
            d = dict(tx=int(tx[i]), kind=int(kind[i]))
            for k, nm in enumerate(names):
                d[nm] = (int(sp[8 * i + 2 * k]), int(sp[8 * i + 2 * k + 1]))
            out.append(d)
        return out, arena


def x509_p256_pubkey(cert: bytes, pem: bool = True) -> Optional[Tuple[bytes, bytes]]:
    """(qx, qy) of a P-256 x509 certificate, or None."""
    qx, qy = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    rc = load().fabgpu_x509_p256_pubkey(cert, len(cert), 1 if pem else 0, qx, qy)
    return (qx.raw, qy.raw) if rc == 0 else None


def preverify_block(csp: "GPUCSP", block: bytes):
    """fabgpu_csp_block_preverify: every creator / endorsement signature of a marshalled block in one fused launch.
    Returns dict(tx_flags, tx_type, tuple_tx, tuple_kind, tuple_status)."""
    buf = np.frombuffer(block, dtype=np.uint8)
    # room for the answers: remembered from the previous block of this provider (a caller that knows its block passes
    # len(block.Data.Data) and the endorsement count; a too-small guess costs a second walk AND a wasted upload)
    cap_tx, cap_tu = getattr(csp, "_pass_caps", (1024, 4096))
    while True:
        n_tx, n_tu = ctypes.c_uint32(0), ctypes.c_uint32(0)
        tx_flags, tx_type = np.zeros(cap_tx, np.uint8), np.zeros(cap_tx, np.uint8)
        t_tx, t_kind, t_st = np.zeros(cap_tu, np.uint32), np.zeros(cap_tu, np.uint8), np.zeros(cap_tu, np.uint8)
        rc = csp._L.fabgpu_csp_block_preverify(csp._h, _p8(buf), buf.size, ctypes.byref(n_tx), _p8(tx_flags), _p8(tx_type), cap_tx, ctypes.byref(n_tu),
                                               t_tx.ctypes.data_as(_u32p), _p8(t_kind), _p8(t_st), cap_tu)
        if rc == -5:   # FABGPU_ETOOBIG: counts are set
            cap_tx, cap_tu = max(cap_tx, n_tx.value), max(cap_tu, n_tu.value)
            csp._pass_caps = (cap_tx, cap_tu)
            continue
        _check(rc, "fabgpu_csp_block_preverify")
        a, b = n_tx.value, n_tu.value
        return dict(tx_flags=tx_flags[:a].copy(), tx_type=tx_type[:a].copy(), tuple_tx=t_tx[:b].copy(), tuple_kind=t_kind[:b].copy(),
                    tuple_status=t_st[:b].copy())


def pass_routes(csp: "GPUCSP"):
    """How the block passes of this provider went: dict(device_walks, host_walks, last_decline)."""
    d, h = ctypes.c_uint64(0), ctypes.c_uint64(0)
    why = ctypes.create_string_buffer(256)
    _check(csp._L.fabgpu_csp_pass_routes(csp._h, ctypes.byref(d), ctypes.byref(h), why, 256), "fabgpu_csp_pass_routes")
    st = (ctypes.c_uint64 * 4)()
    _check(csp._L.fabgpu_csp_pass_stats(csp._h, st), "fabgpu_csp_pass_stats")
    return dict(device_walks=d.value, host_walks=h.value, last_decline=why.value.decode(errors="replace"), relaunches=st[0], device_decoded=st[1],
                learned=st[2], general_der=st[3])


def identity_cache_size(csp: "GPUCSP") -> int:
    n = ctypes.c_uint64(0)
    _check(csp._L.fabgpu_csp_identity_cache_size(csp._h, ctypes.byref(n)), "fabgpu_csp_identity_cache_size")
    return n.value


def block_walk_compare(csp: "GPUCSP", block: bytes):
    """TEST HOOK: the device walker against the host walker on one block -> (identical, declined, text)."""
    buf = np.frombuffer(block, dtype=np.uint8)
    declined = ctypes.c_int(0)
    diff = ctypes.create_string_buffer(256)
    rc = load_hooks().fabgpu_csp_block_walk_compare(csp._h, _p8(buf), buf.size, ctypes.byref(declined), diff, 256)
    if rc < 0:
        raise FabgpuError("fabgpu_csp_block_walk_compare failed: %s (%d)" % (strerror(rc), rc))
    return rc == 0, bool(declined.value), diff.value.decode(errors="replace")


def block_walk_twopass_compare(block: bytes):
    """TEST HOOK (pure host): the device walk's count / prefix-sum / write procedure on the host against ParseBlock ->
    (identical, text); None when both refuse the framing."""
    buf = np.frombuffer(block, dtype=np.uint8)
    diff = ctypes.create_string_buffer(256)
    rc = load_hooks().fabgpu_block_walk_twopass_compare(_p8(buf), buf.size, diff, 256)
    if rc == FABGPU_EINVAL:
        return None
    return rc == 0, diff.value.decode(errors="replace")


def gate_sig_fast(sig: bytes):
    """TEST HOOK (pure host): the device's signature gate -> (code, r32, s32); code 0 submit, 1 high-S, 2 empty, 3 declined."""
    r, s2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    code = load_hooks().fabgpu_gate_sig_fast(sig, len(sig), r, s2)
    return code, r.raw, s2.raw


def gate_sig_any(sig: bytes):
    """TEST HOOK (pure host): the gate the device route applies to every signature -> (code, r32, s32); code 0 submit, 1 high-S, 2 empty,
    4 does not unmarshal / r, s <= 0, 5 r of more than 256 bits."""
    r, s2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    code = load_hooks().fabgpu_gate_sig_any(sig, len(sig), r, s2)
    return code, r.raw, s2.raw


def identity_to_p256(ident: bytes):
    """Pure host: the P-256 key (64 bytes) the host route reads out of a SerializedIdentity, or None."""
    q = ctypes.create_string_buffer(64)
    return q.raw if load().fabgpu_identity_to_p256(ident, len(ident), q) == 0 else None


def idfix_probe(csp: "GPUCSP", idents: Sequence[bytes]):
    """TEST HOOK (device): the device route's identity decoder over many identities -> (codes, keys (n x 64))."""
    n = len(idents)
    arena = np.frombuffer(b"".join(idents) + b"\0" * 8, dtype=np.uint8)
    ends = np.cumsum([len(x) for x in idents], dtype=np.int64)
    spans = np.zeros((n, 2), dtype=np.uint32)
    spans[:, 1] = ends
    spans[1:, 0] = ends[:-1]
    code, key = np.zeros(n, np.uint8), np.zeros((n, 64), np.uint8)
    _check(load_hooks().fabgpu_csp_idfix_probe(csp._h, n, _p8(arena), arena.size, spans.ctypes.data_as(_u32p), _p8(code), _p8(key)), "fabgpu_csp_idfix_probe")
    return code, key


def gate_probe(csp: "GPUCSP", sigs: Sequence[bytes]):
    """TEST HOOK (device): the wavefront form of the signature gate over many signatures -> (codes, r (n x 32), s (n x 32))."""
    n = len(sigs)
    arena = np.frombuffer(b"".join(sigs) + b"\0" * 8, dtype=np.uint8)
    ends = np.cumsum([len(x) for x in sigs], dtype=np.int64)
    spans = np.zeros((n, 2), dtype=np.uint32)
    spans[:, 1] = ends
    spans[1:, 0] = ends[:-1]
    code, r, s2 = np.zeros(n, np.uint8), np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8)
    _check(load_hooks().fabgpu_csp_gate_probe(csp._h, n, _p8(arena), arena.size, spans.ctypes.data_as(_u32p), _p8(code), _p8(r), _p8(s2)), "fabgpu_csp_gate_probe")
    return code, r, s2


def identity_table_hash(b: bytes) -> int:
    return load_hooks().fabgpu_identity_table_hash(b, len(b))


PASS_SEED_MEMO, PASS_NO_BLOCK_SIGS = 1, 2
TUPLE_ST_SKIPPED = 8


def preverify_block2(csp: "GPUCSP", block: bytes, block_seq: int = 0, seed_memo: bool = False, block_sigs: bool = True, lean: bool = False):
    """fabgpu_csp_block_preverify2: the pass with verdicts tied to bytes.  Returns dict(tx_flags, tx_type, tuple_tx, tuple_kind,
    tuple_status, tuple_spans (n x 8), tuple_digest (n x 32), tuple_hashed, tuple_qxy (n x 64), n_block_sigs, block_sigs_understood,
    memo_seeded, arena = block || padding || tail).  lean: only what a memo-seeding caller needs (tx_flags and the counts) - no per-tuple
    arrays are requested from the library and no arena copy is made (the timing benches use this)."""
    buf = np.frombuffer(block, dtype=np.uint8)
    if lean:
        cap_tx, cap_tu = getattr(csp, "_pass_caps", (1024, 4096))
        while True:
            flags = np.zeros(cap_tx, np.uint8)
            ps = _BlockPass()
            ps.block, ps.len, ps.block_seq = buf.ctypes.data, buf.size, block_seq
            ps.flags = (PASS_SEED_MEMO if seed_memo else 0) | (0 if block_sigs else PASS_NO_BLOCK_SIGS)
            ps.cap_tx, ps.cap_tuples, ps.tx_flags = cap_tx, cap_tu, flags.ctypes.data
            rc = csp._L.fabgpu_csp_block_preverify2(csp._h, ctypes.byref(ps))
            if rc == -5:
                cap_tx, cap_tu = max(cap_tx, ps.n_tx), max(cap_tu, ps.n_tuples)
                csp._pass_caps = (cap_tx, cap_tu)
                continue
            _check(rc, "fabgpu_csp_block_preverify2")
            return dict(tx_flags=flags[:ps.n_tx], n_tuples=ps.n_tuples, n_block_sigs=ps.n_block_sigs, memo_seeded=ps.memo_seeded, n_keyed=ps.n_keyed,
                        n_device_decoded=ps.n_device_decoded, ms_stage=[float(x) for x in ps.ms_stage], device_context=int(ps.device_context))
    cap_tx, cap_tu = getattr(csp, "_pass_caps", (1024, 4096))
    tail_cap = 1 << 16
    while True:
        a = dict(tx_flags=np.zeros(cap_tx, np.uint8), tx_type=np.zeros(cap_tx, np.uint8), tuple_tx=np.zeros(cap_tu, np.uint32),
                 tuple_kind=np.zeros(cap_tu, np.uint8), tuple_status=np.zeros(cap_tu, np.uint8), tuple_spans=np.zeros((cap_tu, 8), np.uint32),
                 tuple_digest=np.zeros((cap_tu, 32), np.uint8), tuple_hashed=np.zeros(cap_tu, np.uint8), tuple_qxy=np.zeros((cap_tu, 64), np.uint8),
                 tail=np.zeros(tail_cap, np.uint8))
        ps = _BlockPass()
        ps.block, ps.len, ps.block_seq = buf.ctypes.data, buf.size, block_seq
        ps.flags = (PASS_SEED_MEMO if seed_memo else 0) | (0 if block_sigs else PASS_NO_BLOCK_SIGS)
        ps.cap_tx, ps.cap_tuples, ps.tail_cap = cap_tx, cap_tu, tail_cap
        for k, v in a.items():
            setattr(ps, k, v.ctypes.data)
        rc = csp._L.fabgpu_csp_block_preverify2(csp._h, ctypes.byref(ps))
        if rc == -5:
            cap_tx, cap_tu, tail_cap = max(cap_tx, ps.n_tx), max(cap_tu, ps.n_tuples), max(tail_cap, ps.tail_len)
            csp._pass_caps = (cap_tx, cap_tu)
            continue
        _check(rc, "fabgpu_csp_block_preverify2")
        nt, nu = ps.n_tx, ps.n_tuples
        out = {k: (v[:nt].copy() if k.startswith("tx_") else v[:nu].copy()) for k, v in a.items() if k != "tail"}
        out.update(n_block_sigs=ps.n_block_sigs, block_sigs_understood=bool(ps.block_sigs_understood), memo_seeded=ps.memo_seeded, n_keyed=ps.n_keyed,
                   n_device_decoded=ps.n_device_decoded, ms_stage=[float(x) for x in ps.ms_stage], device_context=int(ps.device_context), arena=bytes(block) + b"\0" * (ps.tail_base - len(block)) + bytes(a["tail"][:ps.tail_len]))
        return out


def memo_lookup(csp: "GPUCSP", qx32: bytes, qy32: bytes, sig: bytes, digest: bytes) -> Optional[int]:
    """The bccsp.Verify(k, sig, digest) question against the verdict memo: tuple status on a hit, None on a miss (ask bccsp/sw)."""
    st = ctypes.c_uint8(255)
    rc = csp._L.fabgpu_csp_memo_lookup(csp._h, qx32, qy32, sig, len(sig), digest, len(digest), ctypes.byref(st))
    return int(st.value) if rc == 0 else None


def hash_lookup(csp: "GPUCSP", msg: bytes) -> Optional[bytes]:
    """The bccsp.Hash(msg, &bccsp.SHA256Opts{}) question against the digest memo (msp/identities.go:173-181): the digest the device computed
    over exactly these bytes, or None (miss: hash on the CPU)."""
    out = ctypes.create_string_buffer(32)
    rc = csp._L.fabgpu_csp_hash_lookup(csp._h, msg, len(msg), out)
    return out.raw if rc == 0 else None


def hash_memo_stats(csp: "GPUCSP"):
    v = [ctypes.c_uint64(0) for _ in range(5)]
    _check(csp._L.fabgpu_csp_hash_memo_stats(csp._h, *[ctypes.byref(x) for x in v]), "fabgpu_csp_hash_memo_stats")
    return dict(hits=v[0].value, misses=v[1].value, blocks_held=v[2].value, bytes_held=v[3].value, refused=v[4].value)


def memo_has_block(csp: "GPUCSP", block_seq: int) -> int:
    """Entries the verdict memo still holds under block_seq."""
    n = ctypes.c_uint64(0)
    _check(csp._L.fabgpu_csp_memo_has_block(csp._h, block_seq, ctypes.byref(n)), "fabgpu_csp_memo_has_block")
    return n.value


def memo_lookup_nym(csp: "GPUCSP", issuer_hash32: bytes, nym_x32: bytes, nym_y32: bytes, sig: bytes, digest: bytes) -> Optional[int]:
    """The verdict memo's entry for an idemix pseudonym signature verified under the issuer key with that ipk.Hash: status, or None."""
    st = ctypes.c_uint8(0)
    rc = csp._L.fabgpu_csp_memo_lookup_nym(csp._h, issuer_hash32, nym_x32, nym_y32, sig, len(sig), digest, len(digest), ctypes.byref(st))
    return st.value if rc == 0 else None


def memo_evict_block(csp: "GPUCSP", block_seq: int) -> int:
    g = ctypes.c_uint64(0)
    _check(csp._L.fabgpu_csp_memo_evict_block(csp._h, block_seq, ctypes.byref(g)), "fabgpu_csp_memo_evict_block")
    return int(g.value)


def memo_stats(csp: "GPUCSP"):
    v = [ctypes.c_uint64(0) for _ in range(4)]
    _check(csp._L.fabgpu_csp_memo_stats(csp._h, *[ctypes.byref(x) for x in v]), "fabgpu_csp_memo_stats")
    return dict(entries=v[0].value, hits=v[1].value, misses=v[2].value, evicted=v[3].value)


def validate_block_endorsements(csp: GPUCSP, txs) -> np.ndarray:
    """Block-level pre-verify pass: txs = list of (prp, [(endorser_bytes, key, sig), ...]).
    Signed message of endorsement j is prp || endorser_j (validator_keylevel.go:247-249).  Returns one flag per tx:
    True iff every endorsement verifies (verify-all-then-evaluate, common/cauthdsl/policy.go:92-94)."""
    keys, msgs, sigs, owner = [], [], [], []
    for t, (prp, ends) in enumerate(txs):
        for endorser, key, sig in ends:
            keys.append(key)
            msgs.append(prp + endorser)
            sigs.append(sig)
            owner.append(t)
    errs = csp.identity_verify_batch(keys, msgs, sigs)
    ok = np.ones(len(txs), dtype=bool)
    for t, e in zip(owner, errs):
        if e is not None:
            ok[t] = False
    return ok
