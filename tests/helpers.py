"""Shared helpers for the parity tests."""

from __future__ import annotations

import math
from pathlib import Path

import numpy as np

from cattus_amd.weights import NetDesc, seeded_blob

GOLDEN = Path(__file__).resolve().parent / "golden"

# The reference's own "same net, different runtime" tolerance
# (training/tests/test_net_output.py:28-33): value rel 1e-5 / abs 1e-6, policy rtol 1e-3 / atol 1e-6.
REF_POLICY_RTOL, REF_POLICY_ATOL = 1e-3, 1e-6
REF_VALUE_REL, REF_VALUE_ABS = 1e-5, 1e-6


def load_golden(name: str):
    z = np.load(GOLDEN / f"{name}.npz")
    d = NetDesc(*[int(x) for x in z["desc"]])
    return d, int(z["seed"]), z


def golden_names():
    return sorted(p.stem for p in GOLDEN.glob("*.npz"))


def outputs_equal_ref_tol(policy, value, policy_ref, value_ref) -> bool:
    """is_outputs_equals of training/tests/test_net_output.py:28-33, per position."""
    ok = True
    for b in range(len(value_ref)):
        ok &= math.isclose(float(value[b]), float(value_ref[b]), rel_tol=REF_VALUE_REL, abs_tol=REF_VALUE_ABS)
        ok &= bool(np.isclose(policy[b], policy_ref[b], rtol=REF_POLICY_RTOL, atol=REF_POLICY_ATOL).all())
    return ok


def blob_for(name: str):
    d, seed, z = load_golden(name)
    return d, seeded_blob(d, seed), z
