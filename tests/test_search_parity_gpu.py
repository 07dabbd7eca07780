"""Search-level parity on the headline network (chess 20x256) at the reference's operating point.

north_star: "results match the reference CPU net within stated fp tolerance (move-visit distributions
bit-identical under fixed seed + greedy argmax)".  Three statements, on the network and weights bench.py
times (BASELINE config 3: 800 simulations per move):

 1. f32 evaluator == CPU oracle network, search level: identical traces (chosen move and every root visit
    count) -- at a size the oracle finishes in seconds -- and, for the full-size run below, leaf for leaf
    on a sample of the positions the games went through.
 2. f32 evaluator, 800 sims/move, 16 games x 16 searched plies: the reference-precision search.
 3. the split-precision evaluator (f16x2, the product default) -- in its direct form AND in the Winograd form the headline
    of bench.py is timed on (an evaluator of max_batch 256) -- and the bf16 evaluator searching the SAME positions
    (teacher-forced, cattus_amd/agreement.py): chosen-move agreement and L1 distance of the root visit distributions
    are reported and bounded (floors stated below and in DESIGN.md section 4).

Reference: engine/src/mcts/mod.rs:156-196 (search), :387-417 (greedy choice), training/tests/test_net_output.py:28-33
(the reference's own bar is per-leaf; there is none at search level)."""

import json
import os
from pathlib import Path

import numpy as np
import pytest

from cattus_amd import agreement as ag
from cattus_amd import selfplay as sp
from cattus_amd.evaluator import HipEvaluator
from cattus_amd.weights import CHESS, NetDesc, seeded_blob
from oracle import oracle

pytestmark = pytest.mark.gpu

# floors for statement 3; measured values go to gpurun_out/search_agreement.json
# bf16, measured (256 searched plies): agreement 0.941, L1 mean 0.0037, p95 0.015, max 0.045 -> about twice that
BF16_MOVE_AGREEMENT_MIN = 0.90
BF16_VISIT_L1_MEAN_MAX = 0.0075
BF16_VISIT_L1_MAX = 0.09
# f16x2 (per-leaf error ~1e-6): a near-tie in a PUCT comparison can still fall the other way; over these 256 searches
# at most 2 chosen moves may differ (>= 99.2 %), see profiles/r03_search_agreement.json for 2,048 and 3,888 searches
F16X2_MOVE_AGREEMENT_MIN = 0.992
F16X2_VISIT_L1_MEAN_MAX = 4e-4
WINOGRAD_KERNELS = ("conv3x3_wino_kernel", "conv3x3_wino4_kernel", "tower_wino4_kernel", "conv3x3_wino8_kernel")  # the Winograd form's kernels (same bits)


def _positions_of(lines, upto):
    out = []
    for line in lines:
        pos = sp.Position("chess")
        for k, m in enumerate(line[:upto]):
            legal = [nn for _, nn in pos.legal_moves()]
            pos = pos.moved(legal.index(m))
            if k >= 1:
                p = pos if pos.turn() == 0 else pos.flipped()  # what the evaluator sees: Player1 to move
                out.append(p.planes())
    return np.stack(out)


def test_chess_20x256_search_f32_equals_oracle_and_bf16_agreement_is_bounded():
    d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
    blob = seeded_blob(d, 2)  # the weights bench.py uses
    games, plies, sims = 16, 16, 800

    with HipEvaluator(blob, batch_size=games, plane_words=1, dtype="f32", flush_us=100) as ev32:
        # ---- 1. f32 == oracle network at search level (4 games, 2 searched plies, 40 sims: ~300 oracle leaves)
        onet = oracle.OracleNet(blob)
        fn, ctx, keep = onet.callback(plane_words=1, threads=1)
        small, t_oracle, t_hip, _ = ag.search_agreement("chess", sp.Net.raw(fn, ctx, keep), sp.Net.hip_batched(ev32), games=4, plies=2,
                                                        sim_num=40, seed=11)
        assert t_oracle == t_hip, "f32 HIP search differs from the search on the CPU oracle network"
        assert small["move_agreement"] == 1.0 and small["visit_l1_max"] == 0.0 and small["plies"] == 8

        # ---- 2. the reference-precision search at full size
        cfg = sp.make_config(sim_num=sims, temperature_policy=[(9999, 0.0)], cache_size=1000000)
        opens = ag.random_openings("chess", games, 2, seed=7)
        ta = ag.run_traces("chess", cfg, sp.Net.hip_batched(ev32), opens, 2, plies)
        lines = [op + [chosen for chosen, _ in t] for op, t in zip(opens, ta)]
        assert sum(len(t) for t in ta) >= games * plies - 8  # a game may end inside the window
        for t in ta:
            for _, visits in t:
                assert sum(n for _, n in visits) >= sims - 1  # fresh tree: sims-1; a reused subtree starts with more
        # leaves of these very games, f32 evaluator vs oracle, bit for bit
        sample = _positions_of(lines, upto=2 + plies)[:: max(1, (games * plies) // 48)][:48]
        p_hip, v_hip = ev32.eval(sample[:games])
        p_or, v_or = onet.forward(sample[:games])
        assert (p_hip == p_or).all() and (v_hip == v_or).all()

    # ---- 3a. the split-precision tower on the same positions
    with HipEvaluator(blob, batch_size=games, plane_words=1, dtype="f16x2", flush_us=100) as evs:
        assert evs.tower_kernel() == "conv3x3_splitw_kernel"
        ts = ag.run_traces("chess", cfg, sp.Net.hip_batched(evs), lines, 2, plies)
        assert ts == ag.run_traces("chess", cfg, sp.Net.hip_batched(evs), lines, 2, plies)  # reproducible
        ps, vs = evs.eval(sample[:games])
    res_s = ag.compare_traces(ta, ts)
    res_s.update(dtype="f16x2", tower_kernel="conv3x3_splitw_kernel", leaf_max_abs_dlogit=float(np.abs(ps - p_or).max()), leaf_max_abs_dvalue=float(np.abs(vs - v_or).max()))
    print("f16x2 vs f32 search agreement:", json.dumps(res_s))
    assert res_s["move_agreement"] >= F16X2_MOVE_AGREEMENT_MIN, res_s
    assert res_s["visit_l1_mean"] <= F16X2_VISIT_L1_MEAN_MAX, res_s

    # ---- 3a'. the same tower in the form the HEADLINE runs: an evaluator created as bench.py creates its own (max_batch 256, tower_form
    # AUTO -> Winograd F(2x2,3x3), conv3x3_wino_kernel) searching the same positions through the same leaf server.  Same floors as the
    # direct form; measured on 2,048 searches (profiles/r04_search_agreement_winograd.jsonl): every move and every visit count equal.
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2", flush_us=100) as evw:
        assert evw.tower_kernel() in WINOGRAD_KERNELS, evw.tower_kernel()
        tw = ag.run_traces("chess", cfg, sp.Net.hip_batched(evw), lines, 2, plies)
        assert tw == ag.run_traces("chess", cfg, sp.Net.hip_batched(evw), lines, 2, plies)  # reproducible, whatever batch a leaf came in
        pw, vw = evw.eval(sample[:games])
        kernel_w = evw.tower_kernel()
    res_w = ag.compare_traces(ta, tw)
    res_w.update(dtype="f16x2", tower_kernel=kernel_w, leaf_max_abs_dlogit=float(np.abs(pw - p_or).max()), leaf_max_abs_dvalue=float(np.abs(vw - v_or).max()))
    print("f16x2 (Winograd form) vs f32 search agreement:", json.dumps(res_w))
    assert res_w["plies"] >= games * plies - 8
    assert res_w["move_agreement"] >= F16X2_MOVE_AGREEMENT_MIN, res_w
    assert res_w["visit_l1_mean"] <= F16X2_VISIT_L1_MEAN_MAX, res_w

    # ---- 3b. bf16 on the same positions
    with HipEvaluator(blob, batch_size=games, plane_words=1, dtype="bf16", flush_us=100) as ev16:
        t16 = ag.run_traces("chess", cfg, sp.Net.hip_batched(ev16), lines, 2, plies)
        again = ag.run_traces("chess", cfg, sp.Net.hip_batched(ev16), lines, 2, plies)
        assert t16 == again  # bf16 results are reproducible too
        p16, v16 = ev16.eval(sample[:games])
    res = ag.compare_traces(ta, t16)
    res.update(games=games, sims_per_move=sims, searched_plies_per_game=plies, opening_plies=2,
               leaf_max_abs_dlogit=float(np.abs(p16 - p_or).max()), leaf_max_abs_dvalue=float(np.abs(v16 - v_or).max()),
               leaf_argmax_agreement=float((p16.argmax(1) == p_or.argmax(1)).mean()))
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parent.parent)) / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "search_agreement.json").write_text(json.dumps({"bf16": res, "f16x2": res_s, "f16x2_winograd": res_w}, indent=1))
    print("bf16 vs f32 search agreement:", json.dumps(res))
    assert res["plies"] >= games * plies - 8
    assert res["move_agreement"] >= BF16_MOVE_AGREEMENT_MIN, res
    assert res["visit_l1_mean"] <= BF16_VISIT_L1_MEAN_MAX, res
    assert res["visit_l1_max"] <= BF16_VISIT_L1_MAX, res
    # where the two searches chose differently, bf16's own favourite had barely more visits than f32's move
    assert res["b_visits_on_a_move_vs_b_best_mean"] >= 0.97, res
