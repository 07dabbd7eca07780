#!/usr/bin/env python3
"""A reduced-cost tower (f16x2 or bf16) against the exact-f32 one at search level, on a larger sample than the GPU test's
(tests/test_search_parity_gpu.py: 16 games):
    python scripts/agreement_study.py [GAMES=128] [PLIES=16] [SIMS=800] [chess20x256|hex7_6x64] [DTYPES=f16x2,bf16]
(a dtype "f16x2w" is the f16x2 tower forced into its Winograd form, which an evaluator of this study's small batches would not pick)
f32 plays GAMES games of chess 20x256 (or hex7 6x64, BASELINE config 2's net) (random 2-ply openings, greedy, noise off)
for PLIES searched plies; each dtype under test searches the same positions (teacher-forced, trees carried over).
Prints one JSON object per dtype."""
import json
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: F401,E402  (its HIP runtime first, as in tests/conftest.py)

from cattus_amd import agreement as ag  # noqa: E402
from cattus_amd import selfplay as sp  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

games = int(sys.argv[1]) if len(sys.argv) > 1 else 128
plies = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 800
which = sys.argv[4] if len(sys.argv) > 4 else "chess20x256"
dtypes = (sys.argv[5] if len(sys.argv) > 5 else "f16x2,bf16").split(",")
if which == "hex7_6x64":
    game, words = "hex7", 2
    d = NetDesc(planes=3, board=7, moves=49, blocks=6, filters=64, vhc=16, phc=16)
    blob = seeded_blob(d, 1)
else:
    game, words = "chess", 1
    d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
    blob = seeded_blob(d, 2)
cfg = sp.make_config(sim_num=sims, temperature_policy=[(9999, 0.0)], cache_size=1000000)
opens = ag.random_openings(game, games, 2, seed=7)
t0 = time.time()
with HipEvaluator(blob, batch_size=games, plane_words=words, dtype="f32", flush_us=100) as ev32:
    ta = ag.run_traces(game, cfg, sp.Net.hip_batched(ev32), opens, 2, plies)
t1f = time.time()
lines = [op + [chosen for chosen, _ in t] for op, t in zip(opens, ta)]
import os  # noqa: E402

for dtype in dtypes:
    t1 = time.time()
    os.environ.pop("CATTUS_WINOGRAD", None)
    if dtype == "f16x2w":
        os.environ["CATTUS_WINOGRAD"] = "1"
    with HipEvaluator(blob, batch_size=games, plane_words=words, dtype=dtype.rstrip("w"), flush_us=100) as evx:
        tower = evx.tower_kernel()
        tb = ag.run_traces(game, cfg, sp.Net.hip_batched(evx), lines, 2, plies)
    t2 = time.time()
    res = ag.compare_traces(ta, tb)
    res.update(net=which, dtype=dtype, tower_kernel=tower, games=games, sims_per_move=sims, searched_plies_per_game=plies, opening_plies=2,
               f32_seconds=round(t1f - t0, 1), seconds=round(t2 - t1, 1))
    print(json.dumps(res), flush=True)
