"""cattus_amd.agreement (search-level comparison of two evaluators) on CPU stand-in networks."""

import numpy as np

from cattus_amd import agreement as ag
from cattus_amd import selfplay as sp
from tests.test_selfplay import _hashed_logits


def test_openings_are_distinct_legal_lines():
    lines = ag.random_openings("chess", 12, 3, seed=5)
    assert len({tuple(l) for l in lines}) == 12
    for l in lines:
        assert sp.play_moves("chess", l) == ("ongoing", 3)
    assert lines == ag.random_openings("chess", 12, 3, seed=5)  # seeded


def test_identical_networks_agree_completely_and_a_perturbed_one_does_not():
    def exact(planes):
        return _hashed_logits(planes, 25)

    def noisy(planes):
        pol, val = _hashed_logits(planes, 25)
        h = (planes.astype(np.uint64).sum(axis=1) % np.uint64(97)).astype(np.float32)
        return pol + 0.5 * np.sin(h[:, None] + np.arange(25, dtype=np.float32)[None, :]), val

    res, ta, tb, lines = ag.search_agreement("hex5", sp.Net.stub("hex5"), sp.Net.stub("hex5"), games=6, plies=5, sim_num=40, seed=3)
    assert res["plies"] == 30 and res["move_agreement"] == 1.0 and res["visit_l1_max"] == 0.0
    assert ta == tb and all(len(l) == 2 + 5 for l in lines)
    res2, ta2, tb2, _ = ag.search_agreement("hex5", sp.Net.python(exact), sp.Net.python(noisy), games=6, plies=5, sim_num=40, seed=3)
    assert res2["plies"] == 30 and 0.0 < res2["visit_l1_mean"] <= 2.0
    assert res2["move_agreement"] < 1.0 or res2["visit_l1_max"] > 0.0
    # B searched A's positions: same root moves at every ply, and A's trace is its own free game
    for a, b in zip(ta2, tb2):
        assert [sorted(m for m, _ in v) for _, v in a] == [sorted(m for m, _ in v) for _, v in b]
