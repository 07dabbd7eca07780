#!/usr/bin/env python3
"""Headline benchmark: MCTS node-evaluations per second on the chess 20x256 network, batch 256.

One "step" = one pass of the leaf-evaluation hot path over one batch of 256 synthetic leaf
positions: bitboard planes (already resident in HBM) -> plane-pack -> 41 fused 3x3 conv+BN+ReLU
launches -> policy/value heads -> logits + values in HBM.  ``value`` is leaves evaluated per
second summed over all ranks (weak scaling: every GPU runs its own batch stream, as self-play
games shard across GPUs with no collective on the evaluation path).

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "chess20x256": dict(game="chess", blocks=20, filters=256, vhc=8, phc=8, batch=256, seed=2),
    # configs[1] and configs[4], selectable for extra measurements
    "hex7_6x64": dict(game="hex7", blocks=6, filters=64, vhc=16, phc=16, batch=128, seed=1),
    "chess40x384": dict(game="chess", blocks=40, filters=384, vhc=8, phc=8, batch=512, seed=3),
    # the headline net at other batch sizes (how the fixed cost per launch amortises)
    "chess20x256_b512": dict(game="chess", blocks=20, filters=256, vhc=8, phc=8, batch=512, seed=2),
    "chess20x256_b1024": dict(game="chess", blocks=20, filters=256, vhc=8, phc=8, batch=1024, seed=2),
    "chess20x256_b128": dict(game="chess", blocks=20, filters=256, vhc=8, phc=8, batch=128, seed=2),
}


def make_workload(name: str):
    from cattus_amd import synth
    from cattus_amd.weights import CHESS, NetDesc, hex_game, seeded_blob

    w = WORKLOADS[name]
    if w["game"] == "chess":
        d = NetDesc(**CHESS, blocks=w["blocks"], filters=w["filters"], vhc=w["vhc"], phc=w["phc"])
        planes = synth.random_chess_planes(w["batch"], w["seed"])
    else:
        d = NetDesc(**hex_game(7), blocks=w["blocks"], filters=w["filters"], vhc=w["vhc"], phc=w["phc"])
        planes = synth.random_hex_planes(w["batch"], 7, w["seed"])
    return d, seeded_blob(d, w["seed"]), planes


def cpu_baseline(blob, planes, budget_s: float = 20.0):
    """The oracle (a plain C port of the same arithmetic) timed on this host's cores, on a bounded
    sample of the same workload."""
    from oracle import oracle

    net = oracle.OracleNet(blob)
    threads = oracle.default_threads()
    n = min(len(planes), threads)
    t0 = time.perf_counter()
    net.forward(planes[:n], threads=threads)
    dt = time.perf_counter() - t0
    # second, larger sample sized to the remaining budget
    n2 = int(min(len(planes), max(n, (budget_s - dt) * n / dt * 0.8)))
    n2 = max(threads, n2 // threads * threads)
    t0 = time.perf_counter()
    net.forward(planes[:n2], threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": n2 / dt,
        "unit": "node-evals/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{n2} of the {len(planes)} bench leaves, oracle/oracle_net.c f32 on {threads} threads, {dt:.1f} s",
    }


def pack_roofline(torch, ev_lib, dev, stream, leaves: int = 262144, reps: int = 20):
    """HBM roofline of the plane-pack kernel (reference layout, f32 NCHW): chess planes for `leaves`
    positions (1.25 GB of output, far beyond the 256 MiB Infinity Cache), timed with events on the
    launch stream.  Algorithmic bytes per leaf = 18*8 in + 18*64*4 out = 4752."""
    import ctypes as C

    planes = torch.randint(0, 2**62, (leaves, 18, 1), dtype=torch.int64, device=dev)
    out = torch.empty((leaves, 18, 8, 8), dtype=torch.float32, device=dev)

    def launch():
        rc = ev_lib.cattus_hip_planes_to_tensor_device(planes.data_ptr(), leaves, 18, 1, 8, leaves, out.data_ptr(), C.c_void_p(stream.cuda_stream))
        assert rc == 0

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        launch()
    e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    nbytes = leaves * 4752
    # spot check against the definition
    host = out[:4].cpu().numpy().reshape(4, 18, 64)
    bits = planes[:4].cpu().numpy().view("uint64").reshape(4, 18)
    for b in range(4):
        for c in range(18):
            want = [(int(bits[b, c]) >> i) & 1 for i in range(64)]
            assert host[b, c].astype(int).tolist() == want
    achieved = nbytes / (us * 1e-6) / 1e9
    return {
        "kernel": "planes_to_tensor_nchw_kernel",
        "bound": "hbm",
        "achieved": achieved,
        "peak": 8000.0,
        "unit": "GB/s",
        "frac": achieved / 8000.0,
        "traffic": measured_traffic("planes_to_tensor_nchw64_kernel"),
        "avg_launch_us": us,
        "leaves_per_launch": leaves,
        "bytes_per_leaf": 4752,
    }


def selfplay_leg(ev_blob, d, dtype: str, local_rank: int, rank: int, world: int, games: int, sims: int):
    """End-to-end leg: real self-play (C++ search + evaluation cache + this GPU's evaluator) with the
    reference's self-play settings (temperature 1.0 for 30 moves, Dirichlet 0.03/0.25:
    training/config/chess_dev.yaml:52-68,83-88).  Every rank plays its own shard of the games."""
    from cattus_amd import selfplay as sp
    from cattus_amd.evaluator import HipEvaluator

    threads = max(1, sp.available_cpus() // max(1, min(world, 8)) - 1)
    # keep the leg near a minute whatever CPU share this rank has: a host thread sustains roughly 25 k
    # simulations/s of chess search, the GPU roughly 350 k evaluations/s, a game costs about 200 * sims leaves
    capacity = min(350e3, threads * 25e3)
    games = int(min(games, max(64, 60.0 * capacity / (200.0 * sims)))) // 2 * 2
    slots = min(1024, games)  # >= 4 batches of leaves in flight keeps the batches full while other slots search
    with HipEvaluator(ev_blob, batch_size=256, plane_words=1, dtype=dtype, device=local_rank) as ev:
        cfg = sp.make_config(sim_num=sims, batch_size=256, threads=threads, concurrent_games=slots, cache_size=1000000,
                             temperature_policy=[(30, 1.0), (9999, 0.0)], prior_noise_alpha=0.03, prior_noise_epsilon=0.25,
                             first_game=rank, game_stride=world, seed=1)  # random streams are per global game index
        t0 = time.perf_counter()
        res = sp.run_self_play("chess", cfg, sp.Net.hip(ev), None, games, keep_records=False)
        dt = time.perf_counter() - t0
    return dict(seconds=dt, games=games, node_evals=res["node_evals"], batches=res["activation_count"], positions=res["positions"],
                threads=threads, slots=slots, sims=sims, steady_rate=res["steady_node_evals"] / max(res["steady_seconds"], 1e-9),
                steady_seconds=res["steady_seconds"])


def measured_traffic(kernel: str):
    """HBM-side bytes per launch from the committed rocprofv3 PMC pass (profiles/), or None."""
    try:
        with open(ROOT / "profiles" / "r01_pmc_hbm_traffic.json") as f:
            return json.load(f)[kernel]["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="chess20x256")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, choices=[1, 2], default=1,
                    help="2 = also time the K steps with two batches in flight (extra object; `value` stays single-stream)")
    ap.add_argument("--selfplay-games", type=int, default=1024, help="games per GPU of the end-to-end self-play leg (0 = skip)")
    ap.add_argument("--selfplay-sims", type=int, default=64)
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the leaf evaluator has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cattus_amd.evaluator import HipEvaluator

    d, blob, planes = make_workload(args.workload)
    batch = len(planes)
    plane_words = planes.shape[2]
    ev = HipEvaluator(blob, batch_size=batch, plane_words=plane_words, dtype=args.dtype, device=local_rank)

    dev = torch.device("cuda", local_rank)
    d_planes = torch.from_numpy(planes.view(np.int64)).to(dev)
    d_policy = torch.empty((batch, d.moves), dtype=torch.float32, device=dev)
    d_value = torch.empty((batch,), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()

    def step():
        ev.eval_device(d_planes.data_ptr(), batch, d_policy.data_ptr(), d_value.data_ptr(), stream.cuda_stream)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the timed kernels produced real numbers
    assert bool(torch.isfinite(d_policy).all()) and bool(torch.isfinite(d_value).all())

    # roofline of the dominant kernel (3x3 conv tower launch): event-stamped launch durations of the same
    # forward, taken right behind the timed region so that the device is in the same state as for `value`
    launch_us, launches = ev.time_tower(batch, 20) if rank == 0 else (0.0, 1)

    # ---- the same K steps with two batches in flight (evaluator lanes 0/1 on two streams), as the self-play
    # driver runs the evaluator: one batch's kernel tails overlap the other's heads.  Reported beside `value`.
    def time_two_lanes():
        side = torch.cuda.Stream(device=dev)
        d_policy2, d_value2 = torch.empty_like(d_policy), torch.empty_like(d_value)
        lanes = [(0, stream, d_policy, d_value), (1, side, d_policy2, d_value2)]

        def step2(i):
            lane, st, pol, val = lanes[i & 1]
            ev.eval_device(d_planes.data_ptr(), batch, pol.data_ptr(), val.data_ptr(), st.cuda_stream, lane=lane)

        for i in range(max(2, args.warmup // 2 * 2)):
            step2(i)
        sync_all()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step2(i)
        sync_all()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert bool((d_policy2 == d_policy).all()) and bool((d_value2 == d_value).all())  # lanes agree bit for bit
        return dt

    elapsed2 = time_two_lanes() if args.lanes == 2 else None

    # ---- secondary measurement: end-to-end self-play games/hour (same network, real search on the host) ----
    sp_out = None
    if args.selfplay_games > 0 and args.workload == "chess20x256":
        leg = selfplay_leg(blob, d, args.dtype, local_rank, rank, world, args.selfplay_games, args.selfplay_sims)
        t = torch.tensor([leg["seconds"], leg["games"], leg["node_evals"], leg["batches"], leg["positions"], leg["steady_rate"]],
                         dtype=torch.float64, device=dev)
        tmax = t.clone()
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        secs = float(tmax[0].item())
        sp_out = {
            "games_per_hour": float(t[1].item()) * 3600.0 / secs,
            "node_evals_per_sec": float(t[2].item()) / secs,
            # while >= 3/4 of the concurrent-game slots still have a game to play, i.e. without the drain at
            # the end of this fixed-size run when batches can no longer be filled (sum over GPUs)
            "steady_node_evals_per_sec": float(t[5].item()),
            "games": int(t[1].item()),
            "sims_per_move": leg["sims"],
            "plies_per_game": float(t[4].item()) / max(1.0, float(t[1].item())),
            "batch_fill": float(t[2].item()) / max(1.0, float(t[3].item())),
            "concurrent_games_per_gpu": leg["slots"],
            "host_threads_per_gpu": leg["threads"],
            "seconds": secs,
            "settings": "temperature 1.0 for 30 moves then 0, Dirichlet noise 0.03/0.25, cache 1e6, batch 256",
        }

    if rank == 0:
        value = world * batch * args.steps / elapsed
        flop_per_launch = d.conv_flops_per_position() * batch / launches
        achieved = flop_per_launch / (launch_us * 1e-6) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        out = {
            "metric": "MCTS node-evals/sec (chess 20x256 net, batch 256)",
            "value": value,
            "unit": "node-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic leaf positions + seeded random-init weights",
            "config": {
                "workload": f"{args.workload}: ConvNetV1 {d.blocks}x{d.filters}, board {d.board}, "
                f"{d.planes} planes, {d.moves} moves, batch {batch} leaves/GPU, evaluator-only",
                "per_gpu_batch": batch,
                "flop_per_leaf": d.flops_per_position(),
            },
            "per_gpu_value": value / world,
            "roofline": {
                "kernel": "tower_persistent_kernel" if launches == 1 else "conv3x3_mfma_v2_kernel",
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                "traffic": measured_traffic("tower_persistent_kernel" if launches == 1 else "conv3x3_mfma_v2_kernel")
                if args.workload == "chess20x256" and args.dtype == "bf16"
                else None,
                "avg_launch_us": launch_us,
                "launches_per_step": launches,
                "flop_per_launch": flop_per_launch,
            },
        }
        if args.workload == "chess20x256":
            from cattus_amd import evaluator as ev_mod

            out["roofline_plane_pack"] = pack_roofline(torch, ev_mod.load_library(), dev, stream)
        if elapsed2 is not None:
            out["two_batches_in_flight"] = {
                "value": batch * world * args.steps / elapsed2,
                "ms_per_step": elapsed2 / args.steps * 1e3,
                "note": "same K steps alternating between the evaluator's two lanes on two streams",
            }
        if sp_out is not None:
            out["selfplay"] = sp_out
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(blob, planes)
        print(json.dumps(out), flush=True)

    ev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
