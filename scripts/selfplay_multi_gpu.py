#!/usr/bin/env python3
"""Self-play sharded across the GPUs of one node (BASELINE config 4).

One process per GPU (launch with torch.distributed.run); rank r plays the global game indices
r, r+W, r+2W, ... on its own evaluator, then the fixed-size .traindata records are pooled with an
all-gather and the win counters with an all-reduce (RCCL over xGMI with --backend nccl).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        scripts/selfplay_multi_gpu.py --game chess --blocks 20 --filters 256 --games-num 1024 \
        --sim-num 800 --batch-size 256 --concurrent-games 64 --threads 12 --out summary.json

``--net stub`` runs the same plumbing on CPU with the deterministic stand-in network (gloo).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cattus_amd import dist as cdist  # noqa: E402
from cattus_amd import selfplay as sp  # noqa: E402
from cattus_amd.weights import CHESS, TTT, NetDesc, hex_game, seeded_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="chess")
    ap.add_argument("--net", choices=["hip", "stub"], default="hip")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--filters", type=int, default=256)
    ap.add_argument("--dtype", default="f16x2", choices=["f16x2", "bf16", "f32"])
    ap.add_argument("--games-num", type=int, default=128)
    ap.add_argument("--sim-num", type=int, default=800)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--concurrent-games", type=int, default=64)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--diverse", action="store_true", help="temperature 1.0 for 30 moves + Dirichlet noise (chess_dev.yaml)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--leaves-in-flight", type=int, default=1, help="> 1: virtual-loss leaf parallelism per tree (not the reference's search)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    backend = args.backend or ("nccl" if args.net == "hip" else "gloo")
    # a disjoint CPU share per rank, on its GPU's NUMA node where sysfs tells (before any worker thread exists)
    from cattus_amd import affinity

    share = affinity.pin_rank(local_rank, local_world, affinity.torch_pci_bus_ids(local_world) if backend == "nccl" and local_world > 1 else None)
    args.threads = max(1, min(args.threads, len(share) - 3)) if local_world > 1 else args.threads
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dev = torch.device("cuda", local_rank)
    else:
        dist.init_process_group("gloo")
        dev = None

    info = sp.game_info(args.game)
    first, stride, local_games = cdist.shard_games(args.games_num, rank, world)
    kw = dict(temperature_policy=[(30, 1.0), (9999, 0.0)], prior_noise_alpha=0.03, prior_noise_epsilon=0.25) if args.diverse else {}
    cfg = sp.make_config(sim_num=args.sim_num, batch_size=args.batch_size, threads=args.threads, concurrent_games=args.concurrent_games,
                         cache_size=1000000, first_game=first, game_stride=stride, seed=args.seed,
                         leaves_in_flight=args.leaves_in_flight, **kw)  # random streams are per global game index: same seed on every rank
    ev = None
    if args.net == "hip":
        from cattus_amd.evaluator import HipEvaluator

        base = CHESS if args.game == "chess" else TTT if args.game in ("ttt", "tictactoe") else hex_game(info["board"])
        d = NetDesc(**base, blocks=args.blocks, filters=args.filters, vhc=8, phc=8)
        ev = HipEvaluator(seeded_blob(d, 2), batch_size=args.batch_size, plane_words=info["plane_words"], dtype=args.dtype, device=local_rank)
        net = sp.Net.hip(ev)
    else:
        net = sp.Net.stub(args.game)

    dist.barrier()
    t0 = time.perf_counter()
    res = sp.run_self_play(args.game, cfg, net, None, local_games)
    t_play = time.perf_counter() - t0
    recs, meta = cdist.pool_records(res["record_bytes"], res["record_meta"], device=dev)
    tot = cdist.reduce_counters(res, device=dev)
    t = torch.tensor([t_play, time.perf_counter() - t0], dtype=torch.float64, device=dev or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        play_s, total_s = t.tolist()
        out = {
            "game": args.game, "n_gpus": world, "games": args.games_num, "sim_num": args.sim_num, "net": args.net,
            "player1_wins": tot["player1_wins"], "player2_wins": tot["player2_wins"], "draws": tot["draws"],
            "records_pooled": int(len(recs)), "record_bytes": int(recs.shape[1]) if len(recs) else info["record_bytes"],
            "node_evals": tot["node_evals"], "seconds_play": play_s, "seconds_total": total_s,
            "node_evals_per_sec": tot["node_evals"] / play_s, "games_per_hour": args.games_num * 3600 / total_s,
            "pool_seconds": total_s - play_s,
            "ranks": dist.get_world_size(), "collective_backend": dist.get_backend(), "threads_per_rank": args.threads,
        }
        assert len(recs) == tot["positions"]
        print(json.dumps(out), flush=True)
        if args.out:
            Path(args.out).write_text(json.dumps(out))
    if ev is not None:
        ev.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
