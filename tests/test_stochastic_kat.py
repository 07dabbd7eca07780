"""Known-answer tests of the search's two stochastic paths, through the C ABI hooks that run the very
sampling functions the search calls (cattus_amd/csrc/host/mcts.h: sample_dirichlet, sample_with_temperature).

The reference draws from an unseeded thread RNG (engine/src/mcts/mod.rs:415,435), so there is nothing to compare draw for
draw; what must hold is the DISTRIBUTION: root noise ~ Dir(alpha, ..., alpha) (mcts/mod.rs:419-446 with
util/dirichlet.rs:226-352), move choice at temperature T with probability proportional to p^(1/T) (mcts/mod.rs:403-415).
A biased sampler would silently change exploration; these tests pin moments and frequencies within sampling error."""

import numpy as np
import pytest
from scipy import stats

from cattus_amd import selfplay as sp

DRAWS = 100_000


@pytest.mark.parametrize("alpha", [0.03, 0.3, 1.0])
@pytest.mark.parametrize("k", [2, 30, 218])
def test_dirichlet_noise_moments(alpha, k):
    x = sp.dirichlet_draws(seed=1234 + k, alpha=alpha, k=k, draws=DRAWS)
    assert np.isfinite(x).all() and (x >= 0).all()
    np.testing.assert_allclose(x.sum(1), 1.0, atol=1e-12)
    a0 = alpha * k
    mean, var = 1.0 / k, (k - 1) / (k * k * (a0 + 1.0))  # Dir(alpha 1_k): E x_i = 1/k, Var x_i = (k-1) / (k^2 (k alpha + 1))
    cov = -1.0 / (k * k * (a0 + 1.0))                    # Cov(x_i, x_j) = -1 / (k^2 (k alpha + 1))
    # sample means: standard error sqrt(var / DRAWS); 5 sigma over k components
    se = np.sqrt(var / DRAWS)
    assert np.abs(x.mean(0) - mean).max() <= 5 * se, (np.abs(x.mean(0) - mean).max(), se)
    # sample variances: within 6 % of the analytic value, every component (the fourth moment of a sparse Dirichlet is large)
    np.testing.assert_allclose(x.var(0), var, rtol=0.06 if alpha >= 0.3 else 0.12)
    np.testing.assert_allclose(x.var(0).mean(), var, rtol=0.02)
    c01 = np.cov(x[:, 0], x[:, 1])[0, 1]
    assert abs(c01 - cov) <= 6 * np.sqrt(var * var / DRAWS) + 0.02 * abs(cov), (c01, cov)


@pytest.mark.parametrize("alpha,k", [(0.3, 2), (1.0, 2), (0.03, 30), (1.0, 30)])
def test_dirichlet_marginal_is_beta(alpha, k):
    """x_1 of Dir(alpha 1_k) ~ Beta(alpha, (k-1) alpha): Kolmogorov-Smirnov against scipy's CDF."""
    x = sp.dirichlet_draws(seed=99, alpha=alpha, k=k, draws=20_000)[:, 0]
    x = np.clip(x, 1e-300, 1.0)
    stat, p = stats.kstest(x, stats.beta(alpha, (k - 1) * alpha).cdf)
    assert p > 1e-3, (stat, p)


def test_dirichlet_streams_are_seeded():
    a = sp.dirichlet_draws(7, 0.3, 5, 10)
    assert (a == sp.dirichlet_draws(7, 0.3, 5, 10)).all()
    assert (a != sp.dirichlet_draws(8, 0.3, 5, 10)).any()


@pytest.mark.parametrize("temperature", [1.0, 0.5, 2.0, 0.1])
def test_temperature_sampling_frequencies(temperature):
    rng = np.random.default_rng(3)
    p = rng.dirichlet(np.full(12, 0.7)).astype(np.float32)
    p[3] = 0.0  # an unvisited move is never chosen
    p /= p.sum()
    counts = sp.temperature_choice_counts(seed=5, probs=p, temperature=temperature, draws=200_000)
    want = p.astype(np.float64) ** (1.0 / temperature)
    want /= want.sum()
    assert counts.sum() == 200_000 and counts[3] == 0
    live = want * 200_000 >= 5
    chi2 = (((counts[live] - want[live] * 200_000) ** 2) / (want[live] * 200_000)).sum()
    assert chi2 <= stats.chi2.ppf(1 - 1e-4, live.sum() - 1), (chi2, counts, want)
    # events with an expectation below 5 draws: never more than a handful
    assert counts[~live].sum() <= 40


def test_temperature_one_reproduces_the_visit_distribution_and_is_seeded():
    p = np.array([0.5, 0.25, 0.125, 0.125], dtype=np.float32)
    c = sp.temperature_choice_counts(11, p, 1.0, 400_000)
    np.testing.assert_allclose(c / 400_000, p, atol=4e-3)
    assert (c == sp.temperature_choice_counts(11, p, 1.0, 400_000)).all()
