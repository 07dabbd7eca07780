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


# ---- chess repetition: game level (chess/core.rs:438-450) and inside the search (mcts/mod.rs:133-154)

_NN = None


def _nn(name):
    global _NN
    if _NN is None:
        _NN = {nm: i for i, nm in enumerate(sp.chess_nn_moves())}
    return _NN[name]


SHUFFLE = ["g1f3", "g8f6", "f3g1", "f6g8"]  # back to the starting position after four plies


def test_chess_threefold_repetition_is_a_draw_at_game_level():
    # the starting position occurs at plies 0, 4, 8: the third occurrence ends the game as a draw
    moves = [_nn(m) for m in SHUFFLE * 2]
    assert sp.play_moves("chess", moves[:7]) == ("ongoing", 7)
    assert sp.play_moves("chess", moves) == (0, 8)
    # it is the repetition, not the position: the same position reached for the second time is ongoing
    assert sp.play_moves("chess", moves[:4]) == ("ongoing", 4)
    # further moves are not played once the game is over (play_single_turn asserts ongoing)
    assert sp.play_moves("chess", moves + [_nn("e2e4")]) == (0, 8)
    # equality ignores the fifty-move counter (chess/core.rs:288-305): the repeated positions carry
    # different counters and still count; a different castling right does not count
    # Rook shuffle h1-h2-h1 / h8-h7-h8 after h4 h5.  The placement after ply 2 recurs at plies 6 and 10, but with
    # both kingside castling rights gone, so ply 2 does not count towards it (it would end the game at ply 10);
    # the position after ply 4 (rooks on h2 / h7, rights already gone) recurs at 8 and 12 with equal rights
    # and different fifty-move counters: that one ends the game.
    rook = ["h2h4", "h7h5"] + ["h1h2", "h8h7", "h2h1", "h7h8"] * 3
    assert sp.play_moves("chess", [_nn(m) for m in rook]) == (0, 12)
    assert sp.play_moves("chess", [_nn(m) for m in rook[:11]]) == ("ongoing", 11)
    # the driver and the trace end such a game too
    cfg = sp.make_config(sim_num=4, cache_size=10000)
    got = sp.trace_game("chess", cfg, sp.Net.stub("chess"), forced=moves, search_from=0)
    assert len(got) == 8


def test_chess_search_scores_a_repeating_line_as_draw_without_expanding_it():
    """After g1f3 g8f6 f3g1 f6g8 g1f3 g8f6 f3g1 the starting position has occurred twice; the child f6g8
    of the root would be its third occurrence.  The search must treat that child as a draw (value 0, never
    expanded), and a grandchild that repeats a history position likewise.  Cross-checked against the
    independent Python search on the same stub network, visit count by visit count, over several plies
    (tree reuse carries unexpanded repetition leaves along)."""
    forced = [_nn(m) for m in (SHUFFLE * 2)[:7]]
    cfg = sp.make_config(sim_num=120, cache_size=100000)
    got = sp.trace_game("chess", cfg, sp.Net.stub("chess"), max_plies=3, forced=forced, search_from=7)
    want, _ = mo.trace_game(mo.make_chess(), 120, 1.41421, max_plies=3, forced=forced, search_from=7)
    assert got == want
    # the repeating move was visited (so the repetition branch ran) ...
    visits = dict(got[0][1])
    assert visits[_nn("f6g8")] > 0
    # ... and with a network that loves that move and hates every position (value -1 for the side to move is
    # irrelevant: the repeating child is scored 0 by rule), all simulations through it stay at depth 1
    def biased(planes):
        n = len(planes)
        pol = np.zeros((n, 1880), dtype=np.float32)
        pol[:, _nn("f6g8")] = 8.0  # black moves at the root: the network sees the flipped position, where this
        pol[:, _nn("f3g1")] = 8.0  # move reads f3g1; both set, so the prior is high whichever side evaluates
        return pol, np.full(n, 0.9, dtype=np.float32)

    cfg2 = sp.make_config(sim_num=50, cache_size=100000)
    got2 = sp.trace_game("chess", cfg2, sp.Net.python(biased), max_plies=1, forced=forced, search_from=7)
    v2 = dict(got2[0][1])
    # 49 simulations below the root; the favoured repeating move takes most of them and none of them expanded
    # it (a draw score of 0 beats the -0.9 every evaluated child returns for black)
    assert v2[_nn("f6g8")] >= 40


def test_searches_next_to_a_repetition_and_the_game_end_match_the_python_search():
    """Searches at plies 6 and 7 of the knight shuffle (the history holds two positions twice, so several
    root children and grandchildren are third occurrences), then the forced eighth move completes the
    threefold repetition and the game loop stops: same visit counts and the same stop as the Python
    restatement of search + game loop."""
    forced = [_nn(m) for m in SHUFFLE * 2]
    cfg = sp.make_config(sim_num=30, cache_size=100000)
    got = sp.trace_game("chess", cfg, sp.Net.stub("chess"), max_plies=4, forced=forced, search_from=6)
    want, _ = mo.trace_game(mo.make_chess(), 30, 1.41421, max_plies=4, forced=forced, search_from=6)
    assert got == want and len(got) == 2  # plies 6 and 7 searched, the game ends with ply 8's position
