"""Search known-answer tests: the C++ host search (cattus_amd/csrc/host/mcts.h, through the C ABI)
against the independent pure-Python restatement in oracle/mcts_oracle.py, with the deterministic stub
network on both sides.  The reference has no test that pins visit counts, so
these are authored from its text: PUCT arithmetic in f32, petgraph edge order, max_by tie rule,
tree reuse with sibling-order reversal, the evaluation cache and the flip to Player1."""

import numpy as np
import pytest

from cattus_amd import selfplay as sp
from oracle import mcts_oracle as mo


def _trace_cpp(game, sim_num, explore=1.41421, **kw):
    cfg = sp.make_config(sim_num=sim_num, explore_factor=explore, temperature_policy=[(9999, 0.0)], cache_size=100000, **kw)
    return sp.trace_game(game, cfg, sp.Net.stub(game))


@pytest.mark.parametrize(
    "game,cls,sims",
    [
        ("tictactoe", mo.Ttt, 2),
        ("tictactoe", mo.Ttt, 3),
        ("tictactoe", mo.Ttt, 50),
        ("tictactoe", mo.Ttt, 600),
        ("hex4", mo.make_hex(4), 40),
        ("hex4", mo.make_hex(4), 100),
        ("hex5", mo.make_hex(5), 60),
    ],
)
def test_visit_counts_match_python_restatement(game, cls, sims):
    want, _ = mo.trace_game(cls, sims, 1.41421)
    got = _trace_cpp(game, sims)
    assert len(got) == len(want)
    for ply, ((gm, gv), (wm, wv)) in enumerate(zip(got, want)):
        assert gv == wv, f"ply {ply}: visit counts differ"
        assert gm == wm, f"ply {ply}: chosen move differs"


def test_fresh_tree_root_visits_sum_to_sims_minus_one():
    # the first simulation on a fresh tree evaluates the root itself (mcts/mod.rs:156-196)
    got = _trace_cpp("tictactoe", 25)
    assert sum(v for _, v in got[0][1]) == 24
    # later plies reuse the subtree, so the root already carries visits
    assert sum(v for _, v in got[2][1]) >= 24


def test_tie_goes_to_first_legal_move_with_uniform_net():
    # uniform priors and value 0: every child ties; max_by over newest-first edges keeps the earliest
    # inserted child, i.e. the first legal move (mcts/mod.rs:218-226)
    def uniform(planes):
        n = len(planes)
        return np.zeros((n, 9), dtype=np.float32), np.zeros(n, dtype=np.float32)

    cfg = sp.make_config(sim_num=2, temperature_policy=[(9999, 0.0)])
    got = sp.trace_game("tictactoe", cfg, sp.Net.python(uniform), max_plies=1)
    chosen, visits = got[0]
    assert visits[-1] == (0, 1) and all(v == 0 for _, v in visits[:-1])
    assert chosen == 0
    # result order is edges() order = newest first = descending move index on a fresh tree
    assert [m for m, _ in visits] == list(range(8, -1, -1))


def test_explore_factor_zero_and_large():
    for c in (0.0, 10.0):
        want, _ = mo.trace_game(mo.Ttt, 30, c)
        assert _trace_cpp("tictactoe", 30, explore=c) == want


def test_cache_does_not_change_results():
    a = sp.trace_game("hex4", sp.make_config(sim_num=50, cache_size=1), sp.Net.stub("hex4"))
    b = sp.trace_game("hex4", sp.make_config(sim_num=50, cache_size=100000), sp.Net.stub("hex4"))
    assert a == b
