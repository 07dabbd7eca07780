#!/usr/bin/env python3
"""Self-play sharded across the GPUs of one node (BASELINE config 4), with rank-failure containment.

    python scripts/selfplay_multi_gpu.py --gpus 8 --game chess --blocks 20 --filters 256 --games-num 1024 \
        --sim-num 800 --batch-size 256 --concurrent-games 64 --threads 12 --work-dir round0 --out summary.json

Without a launcher (no WORLD_SIZE in the environment) this process is the SUPERVISOR (cattus_amd/supervisor.py): it never
touches a GPU, starts one fresh rank process per GPU, and when a rank dies re-queues exactly that rank's unfinished global
game indices on a fresh child process (the reference leaves a dead worker undetected: the TODO at
training/self-play/src/self_play.rs:128).  Rank r plays the global game indices r, r+W, r+2W, ... on its own evaluator and
writes every finished game's records to <work-dir>/out1|out2 and a progress line BEFORE any collective; then the fixed-size
.traindata records are gathered on rank 0 and the win counters all-reduced (RCCL over xGMI with --backend nccl).

Under a launcher (`python -m torch.distributed.run --nproc-per-node N scripts/selfplay_multi_gpu.py --gpus N ...`) every
process is a rank; WORLD_SIZE must equal --gpus, or the script exits non-zero rather than report a job of another size.

``--net stub`` runs the same plumbing on CPU with the deterministic stand-in network (gloo).
"""
import argparse
import datetime
import json
import os
import sys
import tempfile
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; without a launcher this process supervises them")
    ap.add_argument("--game", default="chess")
    ap.add_argument("--net", choices=["hip", "stub"], default="hip")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--filters", type=int, default=256)
    ap.add_argument("--dtype", default="f16x2", choices=["f16x2", "bf16", "f32", "f16"])
    ap.add_argument("--games-num", type=int, default=128)
    ap.add_argument("--sim-num", type=int, default=800)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--concurrent-games", type=int, default=64)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--diverse", action="store_true", help="temperature 1.0 for 30 moves + Dirichlet noise (chess_dev.yaml)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--leaves-in-flight", type=int, default=1, help="> 1: virtual-loss leaf parallelism per tree (not the reference's search)")
    ap.add_argument("--max-game-plies", type=int, default=0, help="> 0: adjudicate a draw after that many plies (bounded samples)")
    ap.add_argument("--work-dir", default=None, help="out1/, out2/ (.traindata files), progress/ and the ranks' counters; default: a temporary directory")
    ap.add_argument("--pg-timeout", type=float, default=600.0, help="deadline in seconds of every collective")
    ap.add_argument("--max-requeues", type=int, default=2)
    ap.add_argument("--rank-timeout", type=float, default=None,
                    help="supervisor: seconds after which ranks still running are killed and their unfinished games re-queued (a rank "
                         "wedged in a GPU call never exits by itself); default: none")
    ap.add_argument("--clean", action="store_true", help="supervisor: remove what an earlier round left in --work-dir (default: refuse to run there)")
    ap.add_argument("--game-list-file", default=None, help="(set by the supervisor) play exactly the global game indices listed in this file")
    ap.add_argument("--out", default=None)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ supervisor


def supervisor_main(args, argv):
    import numpy as np

    from cattus_amd import selfplay as sp
    from cattus_amd import supervisor

    world = args.gpus
    if world is None or world < 1:
        raise SystemExit("selfplay_multi_gpu.py: --gpus N is required when no launcher set WORLD_SIZE")
    if args.games_num % (2 * world) != 0:
        raise SystemExit(f"--games-num {args.games_num} must be a multiple of 2 * --gpus = {2 * world}")
    own_tmp = None
    if args.work_dir is None:
        own_tmp = tempfile.TemporaryDirectory(prefix="cattus_round_")
        args.work_dir = own_tmp.name
        argv = argv + ["--work-dir", args.work_dir]
    work = Path(args.work_dir)
    for d in ("out1", "out2"):
        (work / d).mkdir(parents=True, exist_ok=True)

    def rank_cmd(rank, list_file):
        cmd = [sys.executable, str(Path(__file__).resolve()), *argv]
        if list_file:
            cmd += ["--game-list-file", list_file]
        return cmd

    t0 = time.perf_counter()
    summary = supervisor.supervise(rank_cmd, world, args.games_num, work, max_requeues=args.max_requeues, rank_timeout=args.rank_timeout,
                                   stale="wipe" if args.clean else "refuse")
    info = sp.game_info(args.game)
    pooled = work / "pooled.npz"
    if summary["pooled_via"] == "collective" and pooled.exists():
        z = np.load(pooled)
        recs, meta = z["recs"], z["meta"]
    else:
        summary["pooled_via"] = "files"
        recs, meta = supervisor.pool_from_dirs(work / "out1", work / "out2", info["record_bytes"])
    if len(recs) != summary["positions"]:
        raise SystemExit(f"pooled {len(recs)} records, the progress files count {summary['positions']} positions")
    np.savez(work / "round.npz", recs=recs, meta=meta)
    summary.update(game=args.game, n_gpus=world, sim_num=args.sim_num, net=args.net, records_pooled=int(len(recs)),
                   record_bytes=int(info["record_bytes"]), seconds_total=time.perf_counter() - t0,
                   games_per_hour=args.games_num * 3600 / (time.perf_counter() - t0))
    print(json.dumps(summary), flush=True)
    if args.out:
        Path(args.out).write_text(json.dumps(summary))
    if own_tmp is not None:
        own_tmp.cleanup()
    return 0


# ------------------------------------------------------------------------------------------ one rank


def rank_main(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    from cattus_amd import dist as cdist
    from cattus_amd import selfplay as sp
    from cattus_amd import supervisor
    from cattus_amd.weights import CHESS, TTT, NetDesc, hex_game, seeded_blob

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    requeue = args.game_list_file is not None
    if not requeue and args.gpus is not None and args.gpus != world:
        raise SystemExit(f"selfplay_multi_gpu.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run a job of another size")
    device = int(os.environ.get("CATTUS_LOCAL_DEVICE", local_rank))  # a re-queue child runs on the dead rank's GPU
    tag = os.environ.get("CATTUS_REQUEUE_TAG", f"rank{rank}")
    backend = args.backend or ("nccl" if args.net == "hip" else "gloo")
    from cattus_amd import affinity

    share = affinity.pin_rank(local_rank, local_world, affinity.torch_pci_bus_ids(local_world) if backend == "nccl" and local_world > 1 else None)
    args.threads = max(1, min(args.threads, len(share) - 3)) if local_world > 1 else args.threads
    dev = None
    if backend == "nccl":
        torch.cuda.set_device(device)
        dev = torch.device("cuda", device)
    use_pg = world > 1 and not requeue
    if use_pg:
        # every collective carries a deadline: a peer that died costs the survivors an error, never a hang
        kw = dict(timeout=datetime.timedelta(seconds=args.pg_timeout))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group("gloo", **kw)

    info = sp.game_info(args.game)
    work = Path(args.work_dir) if args.work_dir else None
    out1 = out2 = progress = None
    if work is not None:
        out1, out2 = work / "out1", work / "out2"
        (work / "progress").mkdir(parents=True, exist_ok=True)
        progress = work / "progress" / f"{tag}.txt"
    if requeue:
        games = [int(x) for x in Path(args.game_list_file).read_text().split()]
        shard = dict(game_list=games)
        local_games = len(games)
    else:
        first, stride, local_games = cdist.shard_games(args.games_num, rank, world)
        shard = dict(first_game=first, game_stride=stride)
    kw = dict(temperature_policy=[(30, 1.0), (9999, 0.0)], prior_noise_alpha=0.03, prior_noise_epsilon=0.25) if args.diverse else {}
    cfg = sp.make_config(sim_num=args.sim_num, batch_size=args.batch_size, threads=args.threads, concurrent_games=args.concurrent_games,
                         cache_size=1000000, seed=args.seed, leaves_in_flight=args.leaves_in_flight, max_game_plies=args.max_game_plies,
                         progress_path=progress, **shard, **kw)  # random streams are per global game index: same seed on every rank
    ev = None
    if args.net == "hip":
        from cattus_amd.evaluator import HipEvaluator

        base = CHESS if args.game == "chess" else TTT if args.game in ("ttt", "tictactoe") else hex_game(info["board"])
        d = NetDesc(**base, blocks=args.blocks, filters=args.filters, vhc=8, phc=8)
        ev = HipEvaluator(seeded_blob(d, 2), batch_size=args.batch_size, plane_words=info["plane_words"], dtype=args.dtype, device=device)
        net = sp.Net.hip(ev)
    else:
        net = sp.Net.stub(args.game)

    # fault injection (tests): this rank dies, abruptly, once `k` of its games are complete on disk
    fault_rank, fault_after = os.environ.get("CATTUS_FAULT_RANK"), os.environ.get("CATTUS_FAULT_AFTER_GAMES")
    if fault_rank is not None and int(fault_rank) == rank and not requeue and progress is not None:
        k = int(fault_after or 1)

        def die_after_k():
            while True:
                if len(supervisor.read_progress([progress])) >= k:
                    os._exit(17)
                time.sleep(0)  # a tight poll: the games of the CPU plumbing tests take well under a millisecond each

        threading.Thread(target=die_after_k, daemon=True).start()

    if os.environ.get("CATTUS_HANG_RANK") is not None and int(os.environ["CATTUS_HANG_RANK"]) == rank and not requeue:
        time.sleep(3600)  # fault injection (tests): a rank wedged before it plays -- only --rank-timeout ends it
    if use_pg and os.environ.get("CATTUS_SUPERVISED") != "1":
        dist.barrier()  # a common start for the timing.  Not under the supervisor: a peer that died or hangs before it would take every
        # healthy rank down with it (round-4 review) -- there each rank plays its shard at once and meets the others in the collective
    t0 = time.perf_counter()
    res = sp.run_self_play(args.game, cfg, net, None, local_games, out1, out2)
    t_play = time.perf_counter() - t0
    if work is not None:  # this rank's games are complete on disk: its counters, for the supervisor
        (work / f"{tag}.json").write_text(json.dumps({k: int(res[k]) for k in ("player1_wins", "player2_wins", "draws", "positions", "node_evals",
                                                                              "activation_count", "cache_hits", "cache_misses")}
                                                     | {"seconds_play": t_play}))
    recs = meta = tot = None
    pooled_ok = False
    if use_pg:
        if work is not None and (work / supervisor.ABORT_FLAG).exists():
            print(f"rank {rank}: a peer has died (supervisor's flag): skipping the pooling collective; the records are in {work}", file=sys.stderr, flush=True)
        else:
            try:
                recs, meta = cdist.pool_records(res["record_bytes"], res["record_meta"], device=dev)  # gathered on rank 0
                tot = cdist.reduce_counters(res, device=dev)
                t = torch.tensor([t_play, time.perf_counter() - t0], dtype=torch.float64, device=dev or "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                pooled_ok = True
            except Exception as exc:  # noqa: BLE001 - a dead peer or a deadline: this rank's games are on disk already
                print(f"rank {rank}: pooling collective failed ({type(exc).__name__}: {exc})", file=sys.stderr, flush=True)
                if work is None:
                    raise
    else:
        recs, meta, tot, pooled_ok = res["record_bytes"], res["record_meta"], res, True
        t = torch.tensor([t_play, time.perf_counter() - t0], dtype=torch.float64)
    if rank == 0 and pooled_ok and not requeue:
        play_s, total_s = t.tolist()
        out = {
            "game": args.game, "n_gpus": world, "games": args.games_num, "sim_num": args.sim_num, "net": args.net,
            "player1_wins": tot["player1_wins"], "player2_wins": tot["player2_wins"], "draws": tot["draws"],
            "records_pooled": int(len(recs)), "record_bytes": int(recs.shape[1]) if len(recs) else info["record_bytes"],
            "node_evals": tot["node_evals"], "seconds_play": play_s, "seconds_total": total_s,
            "node_evals_per_sec": tot["node_evals"] / play_s, "games_per_hour": args.games_num * 3600 / total_s,
            "pool_seconds": total_s - play_s,
            "ranks": dist.get_world_size() if use_pg else 1, "collective_backend": dist.get_backend() if use_pg else None,
            "threads_per_rank": args.threads,
        }
        assert len(recs) == tot["positions"]
        if work is not None:
            np.savez(work / "pooled.npz", recs=recs, meta=meta)
        if os.environ.get("CATTUS_SUPERVISED") != "1":  # under the supervisor the one line on stdout is the supervisor's
            print(json.dumps(out), flush=True)
            if args.out:
                Path(args.out).write_text(json.dumps(out))
    if ev is not None:
        ev.close()
    if use_pg:
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 - a peer that is gone must not turn a complete shard into a failed rank
            pass


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.game_list_file is None:
        # no launcher: supervise.  This process imports neither torch.cuda nor the HIP libraries.
        raise SystemExit(supervisor_main(args, argv))
    rank_main(args)


if __name__ == "__main__":
    main()
