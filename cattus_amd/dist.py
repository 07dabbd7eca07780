"""Multi-GPU self-play: game sharding and record pooling.

Games are independent (training/self-play/src/self_play.rs:109,184), so each rank (one process per
GPU) plays the global game indices ``g = rank (mod world)`` with its own evaluator; there is no
collective on the evaluation path.  The reference "pools" records by having every thread write one
file per position into a shared directory (self_play.rs:60); across GPUs the fixed-size records are
pooled once per self-play round with an all-gather (counts first, then zero-padded payload), and the
win counters with an all-reduce.  With backend "nccl" this is RCCL over xGMI; tests use "gloo".
"""

from __future__ import annotations

import numpy as np


def shard_games(games_num: int, rank: int, world: int) -> tuple[int, int, int]:
    """-> (first_game, game_stride, local_games).  Every rank needs an even count
    (self_play.rs:100), so games_num must be a multiple of 2*world."""
    if games_num % (2 * world) != 0:
        raise ValueError(f"games_num {games_num} must be a multiple of 2*world_size = {2 * world}")
    return rank, world, games_num // world


def pool_records(record_bytes: np.ndarray, record_meta: np.ndarray, device=None):
    """All-gather the records of every rank; returns (bytes [N, R] uint8, meta [N, 3] uint32) sorted
    by (game, ply), identical on all ranks."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    rec = torch.from_numpy(np.ascontiguousarray(record_bytes, dtype=np.uint8))
    meta = torch.from_numpy(np.ascontiguousarray(record_meta, dtype=np.uint32).view(np.int32))
    dev = torch.device(device) if device is not None else torch.device("cpu")
    n_local = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nmax, r = max(counts + [1]), rec.shape[1] if rec.ndim == 2 else 0
    r_t = torch.tensor([r], dtype=torch.int64, device=dev)
    dist.all_reduce(r_t, op=dist.ReduceOp.MAX)
    r = int(r_t.item())
    pad_rec = torch.zeros((nmax, r), dtype=torch.uint8, device=dev)
    pad_meta = torch.zeros((nmax, 3), dtype=torch.int32, device=dev)
    if rec.shape[0]:
        pad_rec[: rec.shape[0]] = rec.to(dev)
        pad_meta[: rec.shape[0]] = meta.to(dev)
    all_rec = [torch.empty_like(pad_rec) for _ in range(world)]
    all_meta = [torch.empty_like(pad_meta) for _ in range(world)]
    dist.all_gather(all_rec, pad_rec)
    dist.all_gather(all_meta, pad_meta)
    recs = np.concatenate([t[:c].cpu().numpy() for t, c in zip(all_rec, counts)]) if sum(counts) else np.zeros((0, r), np.uint8)
    metas = (
        np.concatenate([t[:c].cpu().numpy() for t, c in zip(all_meta, counts)]).view(np.uint32)
        if sum(counts)
        else np.zeros((0, 3), np.uint32)
    )
    order = np.lexsort((metas[:, 1], metas[:, 0]))
    return recs[order], metas[order]


def reduce_counters(summary: dict, keys=("player1_wins", "player2_wins", "draws", "positions", "node_evals", "activation_count",
                                          "cache_hits", "cache_misses"), device=None) -> dict:
    """All-reduce(sum) of the win counters and integer metrics (self_play.rs:65-69)."""
    import torch
    import torch.distributed as dist

    dev = torch.device(device) if device is not None else torch.device("cpu")
    t = torch.tensor([int(summary[k]) for k in keys], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = dict(summary)
    for k, v in zip(keys, t.tolist()):
        out[k] = int(v)
    return out
