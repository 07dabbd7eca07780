"""Rank-failure containment (cattus_amd/supervisor.py, scripts/selfplay_multi_gpu.py), on CPU with the stand-in network
over gloo: the counterpart the reference leaves as a TODO (training/self-play/src/self_play.rs:128 -- a worker that dies
is not detected and its games are missing from the round)."""

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from cattus_amd import selfplay as sp
from cattus_amd import supervisor

ROOT = Path(__file__).resolve().parent.parent
SCRIPT = ROOT / "scripts" / "selfplay_multi_gpu.py"


def _run(tmp_path, world, games, extra_env=None, extra_args=()):
    work = tmp_path / "round"
    out = tmp_path / "summary.json"
    cmd = [sys.executable, str(SCRIPT), "--gpus", str(world), "--game", "hex4", "--net", "stub", "--games-num", str(games), "--sim-num", "20",
           "--batch-size", "4", "--concurrent-games", "2", "--threads", "1", "--work-dir", str(work), "--out", str(out), "--pg-timeout", "60",
           *extra_args]
    env = dict(os.environ, **(extra_env or {}))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=300)
    return p, work, out


def _single_process(games):
    return sp.run_self_play("hex4", sp.make_config(sim_num=20, batch_size=4, threads=1), sp.Net.stub("hex4"), None, games)


def test_healthy_round_pools_through_the_collective(tmp_path):
    p, work, out = _run(tmp_path, 2, 8)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1  # the supervisor's line, not the ranks'
    s = json.loads(out.read_text())
    assert s["ranks"] == 2 and s["failed_ranks"] == [] and s["requeued_games"] == [] and s["pooled_via"] == "collective"
    whole = _single_process(8)
    z = np.load(work / "round.npz")
    assert (z["meta"] == whole["record_meta"]).all() and (z["recs"] == whole["record_bytes"]).all()
    assert [s["player1_wins"], s["player2_wins"], s["draws"], s["positions"]] == [whole[k] for k in ("player1_wins", "player2_wins", "draws", "positions")]
    # shard-first: the same records are also on disk, written before the collective
    recs, meta = supervisor.pool_from_dirs(work / "out1", work / "out2", sp.game_info("hex4")["record_bytes"])
    assert (recs == z["recs"]).all() and (meta == z["meta"]).all()


def test_rank_that_dies_after_k_games_is_requeued_on_a_fresh_process(tmp_path):
    """Rank 1 of 2 dies (os._exit, no clean-up) once 2 of its 60 games are complete: the run still finishes, the dead rank's
    unfinished global game indices -- exactly those -- are played by a fresh child process, the pooled records equal the
    single-process run byte for byte, and the summary names the re-queued games."""
    games = 120
    p, work, out = _run(tmp_path, 2, games, extra_env={"CATTUS_FAULT_RANK": "1", "CATTUS_FAULT_AFTER_GAMES": "2"}, extra_args=("--sim-num", "60"))
    assert p.returncode == 0, p.stderr[-3000:]
    s = json.loads(out.read_text())
    assert s["failed_ranks"] == [1] and s["rank_status"] == {"0": 0, "1": 17} and s["pooled_via"] == "files"
    first_life = supervisor.read_progress([work / "progress" / "rank1.txt"])
    assert 2 <= len(first_life) < games // 2  # it died with games unfinished (a few may complete between the poll and the exit)
    want_requeued = [g for g in range(1, games, 2) if g not in first_life]
    assert s["requeued_games"] == want_requeued and want_requeued
    assert [int(x) for x in (work / "requeue1_1.list").read_text().split()] == want_requeued
    whole = sp.run_self_play("hex4", sp.make_config(sim_num=60, batch_size=4, threads=1), sp.Net.stub("hex4"), None, games)
    z = np.load(work / "round.npz")
    assert (z["meta"] == whole["record_meta"]).all() and (z["recs"] == whole["record_bytes"]).all()
    assert [s["player1_wins"], s["player2_wins"], s["draws"], s["positions"]] == [whole[k] for k in ("player1_wins", "player2_wins", "draws", "positions")]


def test_explicit_game_list_plays_exactly_those_games(tmp_path):
    """cattus_sp_config.game_list (the re-queue's input): the listed global game indices, any parity, same records as the
    arithmetic shard that contains them; progress_path gets one line per finished game."""
    prog = tmp_path / "p.txt"
    cfg = sp.make_config(sim_num=20, batch_size=4, threads=1, game_list=[5, 2, 9], progress_path=prog)
    res = sp.run_self_play("hex4", cfg, sp.Net.stub("hex4"), None, 3)
    whole = _single_process(10)
    keep = np.isin(whole["record_meta"][:, 0], [2, 5, 9])
    assert (res["record_meta"] == whole["record_meta"][keep]).all() and (res["record_bytes"] == whole["record_bytes"][keep]).all()
    done = supervisor.read_progress([prog])
    assert sorted(done) == [2, 5, 9] and sum(v[0] for v in done.values()) == res["positions"]
    tally = [sum(1 for v in done.values() if v[1] == t) for t in (0, 1, 2)]
    assert tally == [res["draws"], res["player1_wins"], res["player2_wins"]]
    with pytest.raises(ValueError):
        sp.run_self_play("hex4", cfg, sp.Net.stub("hex4"), None, 4)  # the list holds 3 games


def test_read_progress_ignores_a_torn_last_line(tmp_path):
    f = tmp_path / "p.txt"
    f.write_text("3 7 1 0\n5 9 0 1\n7 1")  # the writer died inside the third line
    assert supervisor.read_progress([f, tmp_path / "missing.txt"]) == {3: (7, 1, 0), 5: (9, 0, 1)}


def test_launcher_world_size_must_match_gpus(tmp_path):
    """Under a launcher WORLD_SIZE != --gpus is an error, not a silently smaller job."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(supervisor.free_port()))
    p = subprocess.run([sys.executable, str(SCRIPT), "--gpus", "2", "--game", "hex4", "--net", "stub", "--games-num", "4", "--sim-num", "20"],
                       cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_a_second_round_in_the_same_work_directory_is_refused_or_cleaned(tmp_path):
    """A work directory serves one round (round-4 review): a second round there would append to the first one's progress files, pool its
    record files and count a dead rank's games as done -- old-network data for the trainer.  Refused by default; --clean removes the
    earlier state first, and then the round is the new seed's, byte for byte."""
    p, work, out = _run(tmp_path, 2, 8, extra_args=("--seed", "1"))
    assert p.returncode == 0, p.stderr[-3000:]
    first = np.load(work / "round.npz")["recs"].copy()
    assert supervisor.stale_state(work)
    # the same directory again, another seed, rank 1 dying at once: refused before any rank starts
    p2, _, _ = _run(tmp_path, 2, 8, extra_env={"CATTUS_FAULT_RANK": "1"}, extra_args=("--seed", "7"))
    assert p2.returncode != 0 and "earlier round" in p2.stderr
    assert (np.load(work / "round.npz")["recs"] == first).all()  # untouched
    p3, _, out3 = _run(tmp_path, 2, 8, extra_args=("--seed", "7", "--clean"))
    assert p3.returncode == 0, p3.stderr[-3000:]
    whole = sp.run_self_play("hex4", sp.make_config(sim_num=20, batch_size=4, threads=1, seed=7), sp.Net.stub("hex4"), None, 8)
    z = np.load(work / "round.npz")
    assert (z["meta"] == whole["record_meta"]).all() and (z["recs"] == whole["record_bytes"]).all()
    assert json.loads(out3.read_text())["requeued_games"] == []


def test_rank_that_hangs_is_killed_at_the_rank_timeout_and_its_games_requeued(tmp_path):
    """--rank-timeout reaches supervise(): a rank that never exits (here: SIGSTOPped by the fault hook's sibling, a sleep far beyond
    the timeout) is killed, its unfinished games are played by a fresh process, the round is whole."""
    games = 8
    p, work, out = _run(tmp_path, 2, games, extra_env={"CATTUS_HANG_RANK": "1"}, extra_args=("--rank-timeout", "8"))
    assert p.returncode == 0, p.stderr[-3000:]
    s = json.loads(out.read_text())
    assert s["failed_ranks"] == [1] and s["rank_status"]["1"] == -9 and s["pooled_via"] == "files"
    assert s["requeued_games"] == list(range(1, games, 2))
    whole = _single_process(games)
    z = np.load(work / "round.npz")
    assert (z["meta"] == whole["record_meta"]).all() and (z["recs"] == whole["record_bytes"]).all()
