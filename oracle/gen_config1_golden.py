#!/usr/bin/env python3
"""tests/golden/e2e/config1_hex4.npz: BASELINE config 1 as SURVEY.md section 8d restates it -- hex4, ConvNetV1 7x16
(VHC = PHC = 16, training/config/hex4_cfg.yaml:9-12) with this repository's seeded weights (seed 0; the bundled model/hex4 net
is an LFS stub of a dead format), sim_num 100, explore_factor 1.41421, temperature [[9999, 0.0]], noise off, cache 1000,
batch_size 1, threads 1, games_num 2 -- played by the C++ search on the CPU oracle network, and cross-checked against
the independent Python search (oracle/mcts_oracle.py) before anything is written.

Stored: per searched ply of game 0 the chosen move and the root visit counts (the visit distribution), and the
.traindata records of both games byte for byte, with their (game, ply, directory) keys.

    python oracle/gen_config1_golden.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from cattus_amd import selfplay as sp  # noqa: E402
from cattus_amd.weights import NetDesc, hex_game, seeded_blob  # noqa: E402
from oracle import mcts_oracle as mo  # noqa: E402
from oracle import oracle  # noqa: E402

OUT = ROOT / "tests" / "golden" / "e2e" / "config1_hex4.npz"


def config1():
    d = NetDesc(**hex_game(4), blocks=7, filters=16, vhc=16, phc=16)
    blob = seeded_blob(d, 0)
    cfg = dict(sim_num=100, explore_factor=1.41421, temperature_policy=[(9999, 0.0)], prior_noise_alpha=0.0, prior_noise_epsilon=0.0,
               cache_size=1000, batch_size=1, threads=1)
    return d, blob, cfg


def flatten_trace(trace) -> np.ndarray:
    out = [len(trace)]
    for chosen, visits in trace:
        out += [chosen, len(visits)]
        for nn, n in visits:
            out += [nn, n]
    return np.array(out, dtype=np.uint32)


def play(net_obj):
    """(trace of game 0, record bytes [n, 113], record meta [n, 3]) with the given sp.Net"""
    d, blob, cfg = config1()
    trace = sp.trace_game("hex4", sp.make_config(**cfg), net_obj)
    res = sp.run_self_play("hex4", sp.make_config(**cfg), net_obj, None, 2)
    order = np.lexsort((res["record_meta"][:, 1], res["record_meta"][:, 0]))
    return trace, res["record_bytes"][order], res["record_meta"][order], res


def main():
    d, blob, cfg = config1()
    net = oracle.OracleNet(blob)
    trace, rec, meta, res = play(sp.Net.python(lambda planes: net.forward(planes.reshape(len(planes), 3, 2), threads=1)))

    def py_net(words, moves):
        p, v = net.forward(np.array(words, dtype=np.uint64).reshape(1, 3, 2), threads=1)
        return p[0], v[0]

    want, _ = mo.trace_game(mo.make_hex(4), cfg["sim_num"], cfg["explore_factor"], net=py_net)
    assert trace == want, "C++ search and Python search disagree: nothing written"
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, trace=flatten_trace(trace), record_bytes=rec, record_meta=meta,
                        result=np.array([res["player1_wins"], res["player2_wins"], res["draws"]], dtype=np.uint32))
    print(f"{OUT}: {len(trace)} plies in game 0, {len(rec)} records of {rec.shape[1]} bytes, result {res['player1_wins']}/{res['player2_wins']}/{res['draws']}")


if __name__ == "__main__":
    main()
