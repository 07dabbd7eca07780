"""libcattus_pool.so (include/cattus_pool.h): the RCCL pooling entry points a non-Python host calls.  On CPU: the library
loads and exports every symbol the header declares.  On the GPU box (one GPU): a one-rank communicator pools a self-play
shard and reduces its counters; the multi-rank path needs one GPU per rank and has not run anywhere yet (RCCL refuses two
ranks on one device)."""

import re
from pathlib import Path

import numpy as np
import pytest

from cattus_amd import pool
from cattus_amd import selfplay as sp

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "cattus_pool.h").read_text()
    declared = sorted(set(re.findall(r"\b(cattus_pool_\w+)\s*\(", header)))
    assert declared == sorted(pool.ABI_SYMBOLS)
    lib = pool.load_library()
    for name in declared:
        getattr(lib, name)


@pytest.mark.gpu
def test_one_rank_pooling_over_rccl():
    res = sp.run_self_play("hex4", sp.make_config(sim_num=20, batch_size=4, threads=1), sp.Net.stub("hex4"), None, 6)
    size = sp.game_info("hex4")["record_bytes"]
    rng = np.random.default_rng(0)
    order = rng.permutation(len(res["record_bytes"]))  # handed over unsorted: the pool returns them by (game, ply)
    with pool.Pool(pool.unique_id(), 0, 1, 0) as p:
        recs, meta, n = p.pool_records(res["record_bytes"][order], res["record_meta"][order], size)
        assert n == res["positions"]
        assert (recs == res["record_bytes"]).all() and (meta == res["record_meta"]).all()
        empty_r, empty_m, n0 = p.pool_records(np.zeros((0, size), np.uint8), np.zeros((0, 3), np.uint32), size)
        assert n0 == 0 and empty_r.shape == (0, size)
        assert p.reduce_counters([res["player1_wins"], res["player2_wins"], res["draws"], 2**40 + 5]) == [
            res["player1_wins"], res["player2_wins"], res["draws"], 2**40 + 5]
    with pytest.raises(RuntimeError):
        pool.Pool(pool.unique_id(), 1, 1, 0)  # rank out of range


@pytest.mark.gpu
def test_a_collective_nobody_matches_returns_a_timeout_status_instead_of_hanging():
    """The failure path (include/cattus_pool.h): the communicator is non-blocking and every wait has a deadline.  A collective
    that does not complete -- what a pooling collective is to the survivors when a peer died before calling it (the reference's TODO,
    training/self-play/src/self_play.rs:128); with one rank: an all-reduce behind a held stream, RCCL refuses a lone self-receive --
    returns CATTUS_POOL_E_TIMEOUT once a 0-second deadline has passed, the communicator
    is aborted, the handle answers CATTUS_POOL_E_STATE from then on and is destroyed without a hang; a new pool works."""
    import time

    p = pool.Pool(pool.unique_id(), 0, 1, 0)
    assert p.reduce_counters([1, 2, 3]) == [1, 2, 3]  # alive
    p.set_timeout(0.0)
    t0 = time.monotonic()
    with pytest.raises(pool.PoolError) as ei:
        p.debug_stalled_collective()
    assert ei.value.status == pool.E_TIMEOUT, str(ei.value)
    assert time.monotonic() - t0 < 30
    with pytest.raises(pool.PoolError) as ei:
        p.reduce_counters([1])
    assert ei.value.status == pool.E_STATE
    p.close()
    with pool.Pool(pool.unique_id(), 0, 1, 0) as q:
        assert q.reduce_counters([5]) == [5]
