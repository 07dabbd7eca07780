"""Pins the CPU oracle against the reference's own network (fixtures made by oracle/gen_golden.py,
which runs training/cattus_train/net_utils.py from the reference checkout)."""

import numpy as np
import pytest

from oracle import oracle

from helpers import blob_for, golden_names, outputs_equal_ref_tol

# Every fixture, the 20-block one included, meets the reference's own cross-runtime tolerance
# (training/tests/test_net_output.py:28-33) at the outputs; the float64 run of the reference module tells rounding
# noise from real error: the oracle is no farther from it than a few times the reference's own f32 run.
DEEP = {"chess_20x256"}  # intermediate activations of the deep tower: looser bound below (f32 noise grows with depth)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_net(name):
    d, blob, z = blob_for(name)
    net = oracle.OracleNet(blob)
    policy, value = net.forward(z["planes"])
    assert outputs_equal_ref_tol(policy, value, z["policy"], z["value"])
    err_oracle = np.abs(policy - z["policy_f64"]).max()
    err_ref = np.abs(z["policy"] - z["policy_f64"]).max()
    assert err_oracle <= 4 * err_ref + 1e-6


@pytest.mark.parametrize("name", golden_names())
def test_oracle_intermediate_activations_match_reference_module(name):
    """Layer-level pin: the stem ConvBlock output and the output of the last residual block of the
    reference module (forward hooks in oracle/gen_golden.py) against the oracle's own intermediates, so that a
    BatchNorm-folding or skip-connection error cannot hide behind the heads."""
    d, blob, z = blob_for(name)
    _, _, stem, tower = oracle.OracleNet(blob).forward_debug(z["planes"][0])
    np.testing.assert_allclose(stem, z["stem0"], rtol=1e-4, atol=1e-5)
    if name in DEEP:
        np.testing.assert_allclose(tower, z["tower0"], rtol=2e-3, atol=2e-4)
    else:
        np.testing.assert_allclose(tower, z["tower0"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_planes_to_tensor(name):
    d, blob, z = blob_for(name)
    planes = z["planes"]
    n = len(planes)
    t = oracle.planes_to_tensor(planes, d.board, n + 3)
    assert t.shape == (n + 3, d.planes, d.board, d.board)
    assert (t[:n] == z["input_tensor"].astype(np.float32)).all()
    assert (t[n:] == 0).all()


def test_planes_to_tensor_rejects_bad_sample_len():
    planes = np.zeros((2, 3, 1), dtype=np.uint64)
    with pytest.raises(ValueError):
        oracle.planes_to_tensor(planes, 3, 1)  # n > batch (net/mod.rs:122-127 assert)


def test_oracle_tanh_accuracy():
    xs = np.concatenate([np.linspace(-12, 12, 4001), np.linspace(-0.6, 0.6, 2001), [0.0, 1e-8, -1e-8, 30.0]])
    got = np.array([oracle.tanhf(float(np.float32(x))) for x in xs])
    ref = np.tanh(xs.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(got, ref, rtol=5e-7, atol=1e-9)


def test_oracle_is_deterministic_and_row_independent():
    d, blob, z = blob_for("chess_7x16")
    net = oracle.OracleNet(blob)
    p1, v1 = net.forward(z["planes"], threads=1)
    p2, v2 = net.forward(z["planes"][::-1].copy(), threads=3)
    assert (p1 == p2[::-1]).all() and (v1 == v2[::-1]).all()


def test_softmax_legal_matches_definition():
    rng = np.random.default_rng(0)
    logits = rng.normal(size=1880).astype(np.float32)
    idx = np.sort(rng.choice(1880, size=37, replace=False)).astype(np.uint32)
    p = oracle.softmax_legal(logits, idx)
    s = logits[idx].astype(np.float64)
    ref = np.exp(s - s.max())
    ref /= ref.sum()
    np.testing.assert_allclose(p, ref, rtol=1e-5)
    assert abs(p.sum() - 1) < 1e-5
