"""Multi-process path on CPU: 2 ranks over gloo shard the games, pool records with all-gather and the
counters with all-reduce; the pooled result equals the single-process run."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from cattus_amd import dist as cdist
from cattus_amd import selfplay as sp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, games, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, stride, local = cdist.shard_games(games, rank, world)
    cfg = sp.make_config(sim_num=25, temperature_policy=[(9999, 0.0)], first_game=first, game_stride=stride, batch_size=2, threads=1)
    res = sp.run_self_play("tictactoe", cfg, sp.Net.stub("tictactoe"), None, local)
    recs, meta = cdist.pool_records(res["record_bytes"], res["record_meta"])
    tot = cdist.reduce_counters(res)
    if rank == 0:
        np.savez(out, recs=recs, meta=meta, w=np.array([tot["player1_wins"], tot["player2_wins"], tot["draws"], tot["positions"]]))
    dist.destroy_process_group()


def test_two_rank_gloo_pooling_matches_single_process(tmp_path):
    games, out = 8, str(tmp_path / "pooled.npz")
    mp.spawn(_worker, args=(2, _free_port(), games, out), nprocs=2, join=True)
    z = np.load(out)
    cfg = sp.make_config(sim_num=25, temperature_policy=[(9999, 0.0)], batch_size=2, threads=1)
    whole = sp.run_self_play("tictactoe", cfg, sp.Net.stub("tictactoe"), None, games)
    assert (z["meta"] == whole["record_meta"]).all()
    assert (z["recs"] == whole["record_bytes"]).all()
    assert list(z["w"]) == [whole["player1_wins"], whole["player2_wins"], whole["draws"], whole["positions"]]


def test_shard_games_requires_even_share():
    assert cdist.shard_games(16, 3, 4) == (3, 4, 4)
    with pytest.raises(ValueError):
        cdist.shard_games(12, 0, 4)


def test_multi_gpu_launcher_with_stub_network_on_gloo(tmp_path):
    """scripts/selfplay_multi_gpu.py through torch.distributed.run, 2 ranks, CPU stand-in network."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    out = tmp_path / "summary.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "scripts" / "selfplay_multi_gpu.py"), "--game", "hex4", "--net", "stub",
           "--games-num", "8", "--sim-num", "20", "--batch-size", "4", "--concurrent-games", "4", "--threads", "1", "--out", str(out)]
    subprocess.check_call(cmd, cwd=str(root), timeout=300)
    s = json.loads(out.read_text())
    assert s["n_gpus"] == 2 and s["player1_wins"] + s["player2_wins"] + s["draws"] == 8
    whole = sp.run_self_play("hex4", sp.make_config(sim_num=20, batch_size=4, threads=1), sp.Net.stub("hex4"), None, 8)
    # each rank has its own evaluation cache, so node_evals differ from the one-process run; records do not
    assert s["records_pooled"] == whole["positions"] and s["node_evals"] >= whole["node_evals"]
