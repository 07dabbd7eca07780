import sys, time, os
sys.path.insert(0,'.')
from cattus_amd import selfplay as sp
import bench
from cattus_amd.evaluator import HipEvaluator
d, blob, planes = bench.make_workload("chess20x256")
with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="bf16") as ev:
    for th, et, soft in ((12, 2, False), (12, 3, False), (12, 4, False), (12, 2, True), (12, 3, True)):
        cfg = sp.make_config(sim_num=800, batch_size=256, threads=th, concurrent_games=1024, cache_size=1000000, max_game_plies=12, seed=1, eval_threads=et, **bench.SELFPLAY_SETTINGS)
        t=time.time()
        res = sp.run_self_play("chess", cfg, sp.Net.hip(ev, device_softmax=soft), None, 1024, keep_records=False)
        dt=time.time()-t
        print(f"lib {os.environ.get('CATTUS_HIP_LIB','default')[-12:]} threads {th} eval_threads {et} devsoftmax {soft}: {res['node_evals']/dt:.0f} evals/s steady {res['steady_node_evals']/res['steady_seconds']:.0f} fill {res['node_evals']/res['activation_count']:.0f}", flush=True)
