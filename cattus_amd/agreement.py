"""Search-level agreement between two evaluators of the same network (e.g. the bf16 and the f32 tower).

Per-leaf tolerances say little about what the search does with the outputs.  This module runs the
reference's search (sequential PUCT, temperature 0, noise off: engine/src/mcts/mod.rs:156-196,387-417)
on the SAME sequence of positions with two evaluators and compares what comes out of it: the move each
search would play and the root visit distribution (the training target, self_play.rs:203-209).

Games: every game starts with a few random legal opening moves (seeded; otherwise all greedy games
would be the same game), then evaluator A plays `plies` searched moves.  Evaluator B is "teacher-forced":
it searches the very positions A's game went through (its tree is carried over from ply to ply exactly
as a player's would be), and the move A played is played whatever B chose.  One host thread per game,
blocking in the network callback -- the reference's threading model; with ``Net.hip_batched`` the
evaluator's leaf server batches the threads' leaves (Batcher::apply, engine/src/util/batch.rs:49-177).
"""

from __future__ import annotations

import threading

import numpy as np

from . import selfplay as sp


def random_openings(game: str, games: int, plies: int, seed: int) -> list[list[int]]:
    """`games` different lines of `plies` random legal moves from the initial position, as policy indices."""
    rng = np.random.default_rng(seed)
    out, seen = [], set()
    while len(out) < games:
        pos, line = sp.Position(game), []
        for _ in range(plies):
            legal = pos.legal_moves()
            k = int(rng.integers(len(legal)))
            line.append(legal[k][1])
            pos = pos.moved(k)
        if pos.status() == "ongoing" and tuple(line) not in seen:
            seen.add(tuple(line))
            out.append(line)
    return out


def run_traces(game: str, cfg, net: sp.Net, forced: list[list[int]], search_from: int, plies: int):
    """One thread per game: trace_game(forced[g], search_from, plies) on `net`; returns the traces in order."""
    out = [None] * len(forced)
    errs = []

    def work(g):
        try:
            out[g] = sp.trace_game(game, cfg, net, max_plies=plies, forced=forced[g], search_from=search_from)
        except Exception as e:  # pragma: no cover - reported below
            errs.append(e)

    ths = [threading.Thread(target=work, args=(g,)) for g in range(len(forced))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]
    return out


def compare_traces(a, b) -> dict:
    """a, b: lists (games) of traces [(chosen, [(move, visits), ...]), ...] over the same positions."""
    same, total, l1s, top_share = 0, 0, [], []
    for ta, tb in zip(a, b):
        for (ca, va), (cb, vb) in zip(ta, tb):
            da, db = dict(va), dict(vb)
            if set(da) != set(db):
                raise ValueError("the two searches did not see the same root position")
            na, nb = float(sum(da.values())), float(sum(db.values()))
            l1s.append(sum(abs(da[m] / na - db[m] / nb) for m in da))
            # how much of B's visits went to the move A chose (1.0 = B is as sure of A's move as it can be)
            top_share.append(db[ca] / nb / max(max(db.values()) / nb, 1e-30))
            same += ca == cb
            total += 1
    l1 = np.array(l1s)
    return {
        "plies": total,
        "move_agreement": same / max(total, 1),
        "visit_l1_mean": float(l1.mean()) if total else 0.0,
        "visit_l1_p95": float(np.quantile(l1, 0.95)) if total else 0.0,
        "visit_l1_max": float(l1.max()) if total else 0.0,
        "b_visits_on_a_move_vs_b_best_mean": float(np.mean(top_share)) if total else 0.0,
    }


def search_agreement(game: str, net_a: sp.Net, net_b: sp.Net, games: int, plies: int, sim_num: int, opening_plies: int = 2,
                     seed: int = 1, cache_size: int = 1000000, explore_factor: float = 1.41421):
    """-> (comparison dict, traces of A, traces of B, the forced lines B searched)."""
    cfg = sp.make_config(sim_num=sim_num, explore_factor=explore_factor, temperature_policy=[(9999, 0.0)], cache_size=cache_size)
    opens = random_openings(game, games, opening_plies, seed)
    ta = run_traces(game, cfg, net_a, opens, opening_plies, plies)
    lines = [op + [chosen for chosen, _ in t] for op, t in zip(opens, ta)]
    tb = run_traces(game, cfg, net_b, lines, opening_plies, plies)
    res = compare_traces(ta, tb)
    res.update(games=games, sims_per_move=sim_num, opening_plies=opening_plies)
    return res, ta, tb, lines
