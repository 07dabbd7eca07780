"""Multi-GPU self-play: game sharding and record pooling.

Games are independent (training/self-play/src/self_play.rs:109,184), so each rank (one process per
GPU) plays the global game indices ``g = rank (mod world)`` with its own evaluator; there is no
collective on the evaluation path.  The reference "pools" records by having every thread write one
file per position into a shared directory (self_play.rs:60); across GPUs the fixed-size records are
pooled once per self-play round on rank 0 (counts all-gathered first, then one gather of the zero-padded payloads),
and the win counters with an all-reduce.  With backend "nccl" this is RCCL over xGMI; tests use "gloo".
Failure containment (a rank that dies; cattus_amd/supervisor.py): every rank also writes its records to the out
directories and a progress line per finished game BEFORE any collective, so a hung or failed collective loses nothing.
"""

from __future__ import annotations

import numpy as np


def shard_games(games_num: int, rank: int, world: int) -> tuple[int, int, int]:
    """-> (first_game, game_stride, local_games).  Every rank needs an even count
    (self_play.rs:100), so games_num must be a multiple of 2*world."""
    if games_num % (2 * world) != 0:
        raise ValueError(f"games_num {games_num} must be a multiple of 2*world_size = {2 * world}")
    return rank, world, games_num // world


def pool_records(record_bytes: np.ndarray, record_meta: np.ndarray, device=None, dst: int | None = 0):
    """Pool the records of every rank on rank `dst` (default 0: the one that writes / hands them to the trainer); returns
    (bytes [N, R] uint8, meta [N, 3] uint32) sorted by (game, ply) there and (None, None) on the other ranks.  Counts travel
    first (all-gather of one int64), then each rank's payload zero-padded to the largest shard in ONE gather: only `dst` holds
    world x largest shard, and only `dst` copies anything back to the host.  dst=None: every rank gets the pooled set
    (all-gather; world x the memory and the host copies on every rank -- tests and small runs)."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    rec = torch.from_numpy(np.ascontiguousarray(record_bytes, dtype=np.uint8))
    meta = torch.from_numpy(np.ascontiguousarray(record_meta, dtype=np.uint32).view(np.uint8).reshape(len(record_meta), 12))
    dev = torch.device(device) if device is not None else torch.device("cpu")
    # counts and record width of every rank (a rank that played nothing has width 0)
    mine = torch.tensor([rec.shape[0], rec.shape[1] if rec.ndim == 2 else 0], dtype=torch.int64, device=dev)
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    counts = [int(c[0].item()) for c in everyone]
    r = max(int(c[1].item()) for c in everyone)
    nmax = max(counts + [1])
    # one payload per rank: [nmax][r record bytes | 12 meta bytes]
    pad = torch.zeros((nmax, r + 12), dtype=torch.uint8, device=dev)
    if rec.shape[0]:
        pad[: rec.shape[0], :r] = rec.to(dev)
        pad[: rec.shape[0], r:] = meta.to(dev)
    if dst is None:
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
    else:
        parts = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, parts, dst=dst)
        if rank != dst:
            return None, None
    if not sum(counts):
        return np.zeros((0, r), np.uint8), np.zeros((0, 3), np.uint32)
    flat = np.concatenate([t[:c].cpu().numpy() for t, c in zip(parts, counts) if c])
    recs = np.ascontiguousarray(flat[:, :r])
    metas = np.ascontiguousarray(flat[:, r:]).view(np.uint32).reshape(-1, 3)
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
