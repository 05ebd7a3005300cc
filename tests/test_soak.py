"""Randomised GPU-vs-oracle soak (tests/soak_parity.py) as a short GPU test; FABGPU_SOAK_SECONDS lengthens it."""
import os

import pytest

import soak_parity


@pytest.mark.gpu
def test_randomised_parity_soak():
    r = soak_parity.soak(float(os.environ.get("FABGPU_SOAK_SECONDS", "15")), seed=int(os.environ.get("FABGPU_SOAK_SEED", "7")))
    assert r["soak"] == "ok", r
    assert r["batches"] >= 3 and r["tuples"] > 1000


@pytest.mark.gpu
def test_registered_keys_soak_while_their_16_bit_tables_are_being_built():
    """tests/soak_parity.py, registered-key batches only, on a context with FABGPU_FLAG_KEY_TABLES_16BIT: every batch registers new keys and
    verifies at once - wavefronts meet keys whose 16-bit table is finished, still building, or (past the 64th key) never coming."""
    r = soak_parity.soak(float(os.environ.get("FABGPU_SOAK_SECONDS", "8")), seed=int(os.environ.get("FABGPU_SOAK_SEED", "7")), tables16=True, keyed_only=True)
    assert r["soak"] == "ok", r
    assert r["keyed"] >= 3


@pytest.mark.gpu
def test_randomised_idemix_soak():
    """tests/soak_idemix.py for a few seconds (thousands of calls through one context: the side launch's flag protocol must never let a stale
    or half-written record through); profiles/r05_soak_idemix.json is the 150-second run (147 562 calls, 240 M signatures)"""
    import soak_idemix
    r = soak_idemix.soak(float(os.environ.get("FABGPU_SOAK_SECONDS", "10")), seed=int(os.environ.get("FABGPU_SOAK_SEED", "7")))
    assert r["calls"] >= 100 and r["signatures"] > 10000, r


@pytest.mark.gpu
def test_digest_memo_soak_under_churn():
    """tests/soak_hash_memo.py for a few seconds: passes, evictions and lookups from six threads at once over a pool of three kept block
    copies - a hit is always SHA-256 of the bytes that were asked about, a mutated message never hits, the verdict memo answers with the
    pass's status or misses; profiles/r06_soak_hash_memo.json is the long run"""
    import soak_hash_memo
    r = soak_hash_memo.soak(float(os.environ.get("FABGPU_SOAK_SECONDS", "8")), seed=int(os.environ.get("FABGPU_SOAK_SEED", "7")))
    assert r["soak"] == "ok", r
    assert r["hash_hits"] > 100 and r["mutants_asked"] > 50 and r["evictions"] > 3, r
