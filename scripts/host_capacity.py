import sys, time, os
sys.path.insert(0,'.')
from cattus_amd import selfplay as sp
import bench
def throttled():
    try:
        st = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(st.get("nr_throttled", 0)), int(st.get("throttled_usec", 0)) / 1e6, int(st.get("usage_usec", 0)) / 1e6
    except OSError:
        return 0, 0.0, 0.0


def run(sims, threads, games, plies, net, tag):
    th0 = throttled()
    cfg = sp.make_config(sim_num=sims, batch_size=256, threads=threads, concurrent_games=games, cache_size=1000000, max_game_plies=plies, seed=1, **bench.SELFPLAY_SETTINGS)
    t=time.time()
    res = sp.run_self_play("chess", cfg, net, None, games, keep_records=False)
    dt=time.time()-t
    th1 = throttled()
    print(f"{tag} sims {sims} threads {threads} games {games}: {res['node_evals']/dt:.0f} evals/s, {res['positions']*sims/dt:.0f} sims/s, fill {res['node_evals']/res['activation_count']:.0f}, {dt:.1f}s; "
          f"cgroup: throttled {th1[0]-th0[0]} periods {th1[1]-th0[1]:.1f}s, cpu used {(th1[2]-th0[2])/dt:.1f} cores", flush=True)
print("cpus", sp.available_cpus(), flush=True)
for th in (1, 4, 8, 15):
    run(800, th, 512, 3 if th < 8 else 6, sp.Net.stub("chess"), "stub")
print("CATTUS_HIP_WAIT =", os.environ.get("CATTUS_HIP_WAIT", "(block)"), flush=True)
from cattus_amd.evaluator import HipEvaluator
d, blob, planes = bench.make_workload("chess20x256")
with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="bf16") as ev:
    for th, games in ((15, 1024), (12, 1024), (8, 1024), (6, 1024), (12, 512)):
        run(800, th, games, 12, sp.Net.hip(ev), "hip")
