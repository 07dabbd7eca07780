#!/usr/bin/env python3
"""End-to-end self-play throughput on one GPU: chess 20x256, batch 256.

    python scripts/e2e_selfplay.py THREADS SLOTS SIMS GAMES [diverse] [devsoftmax] [serial] [lif=K] [batch=B] [plies=P] [dtype=f16x2|bf16|f32] [wholeplies=N]

The STEADY window is the run up to the moment fewer than 3/4 of the slots still have a game to play: with GAMES >= 2 x SLOTS the
slots are refilled from the queue (the reference: games_queue.fetch_add, self_play.rs:184) and the window is the steady state of a
long self-play round.  `games_per_hour_steady` = plies played per second in that window x 3600 / plies of a whole game (`wholeplies`,
default 277.5: the mean of the 512 whole 800-sim games of profiles/r04_e2e_whole_games_800sims.json) -- measured plies over measured
game length, no estimate of the search.

``diverse`` uses the reference's self-play settings (temperature 1.0 for the first 30 moves,
Dirichlet noise 0.03/0.25: training/config/chess_dev.yaml:52-68,83-88) so that games differ and the
evaluation cache sees a realistic hit rate; without it every game is the same deterministic game.
"""
import json
import sys
import threading
import time

sys.path.insert(0, ".")
from cattus_amd import selfplay as sp  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

threads, slots, sims, games = (int(x) for x in sys.argv[1:5])
diverse = "diverse" in sys.argv[5:]
eval_threads = 1 if "serial" in sys.argv[5:] else 2  # batches in flight
lif = max([int(a[4:]) for a in sys.argv[5:] if a.startswith("lif=")] or [1])  # leaves in flight per tree (virtual loss)
batch = max([int(a[6:]) for a in sys.argv[5:] if a.startswith("batch=")] or [256])  # leaves per batch (search side)
plies = max([int(a[6:]) for a in sys.argv[5:] if a.startswith("plies=")] or [0])  # adjudicate after this many plies (0: play out)
devsoftmax = "devsoftmax" in sys.argv[5:]  # legal-move softmax on the GPU (cattus_hip_eval_legal)
dtype = ([a[6:] for a in sys.argv[5:] if a.startswith("dtype=")] or ["f16x2"])[-1]
wholeplies = float(([a[11:] for a in sys.argv[5:] if a.startswith("wholeplies=")] or ["277.5"])[-1])
d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
blob = seeded_blob(d, 2)
with HipEvaluator(blob, batch_size=256, plane_words=1, dtype=dtype) as ev:
    stop = threading.Event()

    def heartbeat():  # a long run keeps saying it is alive (evaluator statistics once a minute)
        t0 = time.time()
        while not stop.wait(60):
            st = ev.stats()
            print(f"[{time.time() - t0:5.0f} s] {st['positions']} leaves in {st['batches']} batches", file=sys.stderr, flush=True)

    threading.Thread(target=heartbeat, daemon=True).start()
    kw = dict(temperature_policy=[(30, 1.0), (9999, 0.0)], prior_noise_alpha=0.03, prior_noise_epsilon=0.25) if diverse else {}
    cfg = sp.make_config(sim_num=sims, batch_size=batch, max_game_plies=plies, threads=threads, concurrent_games=slots, cache_size=1000000, eval_threads=eval_threads, leaves_in_flight=lif, **kw)
    t = time.time()
    res = sp.run_self_play("chess", cfg, sp.Net.hip(ev, device_softmax=devsoftmax), None, games, keep_records=False)
    dt = time.time() - t
    st = ev.stats()
    kernel = ev.tower_kernel()
    stop.set()
steady_plies_per_s = res["steady_plies"] / max(res["steady_seconds"], 1e-9)
print(json.dumps(dict(
    tower_kernel=kernel, steady_plies=res["steady_plies"], steady_plies_per_s=round(steady_plies_per_s, 2),
    steady_batch_fill=round(res["steady_node_evals"] / max(1, res["steady_batches"]), 1), whole_game_plies=wholeplies,
    games_per_hour_steady=round(steady_plies_per_s * 3600 / wholeplies),
    dtype=dtype, threads=threads, slots=slots, sims=sims, games=games, diverse=diverse, device_softmax=devsoftmax, eval_threads=eval_threads, leaves_in_flight=lif, batch=batch, max_game_plies=plies, seconds=round(dt, 3),
    node_evals=res["node_evals"], evals_per_s=round(res["node_evals"] / dt),
    steady_evals_per_s=round(res["steady_node_evals"] / res["steady_seconds"]), steady_seconds=round(res["steady_seconds"], 2), batches=res["activation_count"],
    batch_fill=round(res["node_evals"] / max(1, res["activation_count"]), 1), positions=res["positions"],
    plies_per_game=round(res["positions"] / games, 1), games_per_hour=round(games * 3600 / dt),
    cache_hits=res["cache_hits"], gpu_busy_frac=round(st["run_seconds_total"] / dt, 3),
    results=[res["player1_wins"], res["player2_wins"], res["draws"]])))
