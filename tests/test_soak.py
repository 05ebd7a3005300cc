"""Randomised GPU-vs-oracle soak (tests/soak_parity.py) as a short GPU test; FABGPU_SOAK_SECONDS lengthens it."""
import os

import pytest

import soak_parity


@pytest.mark.gpu
def test_randomised_parity_soak():
    r = soak_parity.soak(float(os.environ.get("FABGPU_SOAK_SECONDS", "15")), seed=int(os.environ.get("FABGPU_SOAK_SEED", "7")))
    assert r["soak"] == "ok", r
    assert r["batches"] >= 3 and r["tuples"] > 1000
