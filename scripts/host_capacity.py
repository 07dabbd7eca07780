import sys, time, os
sys.path.insert(0,'.')
from cattus_amd import selfplay as sp
import bench
def run(sims, threads, games, plies, net, tag):
    cfg = sp.make_config(sim_num=sims, batch_size=256, threads=threads, concurrent_games=games, cache_size=1000000, max_game_plies=plies, seed=1, **bench.SELFPLAY_SETTINGS)
    t=time.time()
    res = sp.run_self_play("chess", cfg, net, None, games, keep_records=False)
    dt=time.time()-t
    print(f"{tag} sims {sims} threads {threads} games {games}: {res['node_evals']/dt:.0f} evals/s, {res['positions']*sims/dt:.0f} sims/s, fill {res['node_evals']/res['activation_count']:.0f}, {dt:.1f}s", flush=True)
print("cpus", sp.available_cpus(), flush=True)
for th in (1, 4, 8, 15):
    run(800, th, 512, 3 if th < 8 else 6, sp.Net.stub("chess"), "stub")
from cattus_amd.evaluator import HipEvaluator
d, blob, planes = bench.make_workload("chess20x256")
with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="bf16") as ev:
    for th, games in ((15, 512), (15, 1024), (8, 512), (12, 512)):
        run(800, th, games, 12, sp.Net.hip(ev), "hip")
