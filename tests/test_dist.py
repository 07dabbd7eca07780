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
