"""Failure containment for self-play sharded over one process per GPU.

The reference fans self-play out over worker threads and leaves a dead worker undetected (the TODO at
training/self-play/src/self_play.rs:128: a panicking thread's games are silently missing from the round).  Here the unit of
failure is the per-GPU process, and a supervisor -- a plain parent process that never touches a GPU -- contains it:

* every rank writes each finished game's records to the out directories (the reference's own pooling: one file per
  position, self_play.rs:60,273) and then one line to its progress file, BEFORE any collective: a rank that dies, or a
  collective that hangs, loses nothing that was finished;
* when a rank exits non-zero the supervisor raises an abort flag (the survivors skip the pooling collective instead of
  waiting for a peer that is gone; every process group also carries a timeout as the backstop) and, once the survivors are
  done, re-queues exactly the dead rank's unfinished global game indices on a FRESH child process on the same device.
  A game is a function of (seed, global game index, networks) -- tests/test_selfplay.py asserts schedule invariance --
  so a re-played game rewrites the files `{g:08}_{ply:03}.traindata` with the same bytes: records are idempotent;
* the round's result is then pooled from the files and the progress lines; the summary carries `requeued_games`.

Pure host logic (subprocess, files): covered on CPU with the stand-in network over gloo (tests/test_supervisor.py).
"""

from __future__ import annotations

import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ABORT_FLAG = "pool_abort"  # created in the work directory when a rank has died: survivors skip the collective


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def read_progress(paths) -> dict[int, tuple[int, int, int]]:
    """Progress files (include/cattus_selfplay.h: `game plies tally adjudicated` per finished game) ->
    {game_idx: (plies, tally, adjudicated)}.  A torn last line (the process died while writing it) is ignored: that
    game counts as unfinished and is played again."""
    done: dict[int, tuple[int, int, int]] = {}
    for p in paths:
        try:
            text = Path(p).read_text()
        except OSError:
            continue
        for line in text.split("\n")[: -1 if not text.endswith("\n") else None]:
            f = line.split()
            if len(f) != 4:
                continue
            try:
                g, plies, tally, adj = (int(x) for x in f)
            except ValueError:
                continue
            done[g] = (plies, tally, adj)
    return done


def shard_of(rank: int, world: int, games_num: int) -> list[int]:
    """Global game indices of rank `rank`: g = rank (mod world) (cattus_amd/dist.py::shard_games)."""
    return list(range(rank, games_num, world))


def pool_from_dirs(out_dir1, out_dir2, record_bytes: int):
    """The reference's pooling: the round's records are the files of the two out directories.  -> (bytes [N, R] uint8,
    meta [N, 3] uint32 = game, ply, dir) sorted by (game, ply), the order of cattus_amd.dist.pool_records."""
    rows = []
    for d, (path) in enumerate((out_dir1, out_dir2)):
        for f in Path(path).glob("*.traindata"):
            g, ply = f.stem.split("_")
            rows.append((int(g), int(ply), d, f))
    rows.sort(key=lambda r: (r[0], r[1]))
    recs = np.zeros((len(rows), record_bytes), dtype=np.uint8)
    meta = np.zeros((len(rows), 3), dtype=np.uint32)
    for i, (g, ply, d, f) in enumerate(rows):
        b = f.read_bytes()
        if len(b) != record_bytes:
            raise ValueError(f"{f}: {len(b)} bytes, a record has {record_bytes}")
        recs[i] = np.frombuffer(b, dtype=np.uint8)
        meta[i] = (g, ply, d)
    return recs, meta


def stale_state(work_dir) -> list[str]:
    """What an earlier round left in a work directory: progress lines, counters, re-queue lists, pooled records, record files.
    Any of it would be read as THIS round's (progress files are appended to, a game with a progress line is never played again,
    pool_from_dirs pools every .traindata it finds): a second round in the same directory would hand the trainer the first round's
    records -- old-network data -- and count a dead rank's games as done."""
    work = Path(work_dir)
    found = [str(p.relative_to(work)) for pat in ("progress/*.txt", "rank*.json", "requeue*", "pooled.npz", "round.npz", ABORT_FLAG)
             for p in sorted(work.glob(pat))]
    for d in ("out1", "out2"):
        n = sum(1 for _ in (work / d).glob("*.traindata")) if (work / d).is_dir() else 0
        if n:
            found.append(f"{d}/ ({n} records)")
    return found


def wipe_state(work_dir) -> None:
    work = Path(work_dir)
    for pat in ("progress/*.txt", "rank*.json", "requeue*", "pooled.npz", "round.npz", ABORT_FLAG, "out1/*.traindata", "out2/*.traindata"):
        for p in work.glob(pat):
            p.unlink()


def supervise(rank_cmd, world: int, games_num: int, work_dir, *, max_requeues: int = 2, rank_timeout: float | None = None,
              env_extra: dict | None = None, log=sys.stderr, stale: str = "refuse") -> dict:
    """Run `world` rank processes, contain the ones that fail, return the round's summary.

    rank_cmd(rank, requeue_list_file | None) -> argv of one rank process.  A regular rank gets RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_* in its environment (what torch.distributed.run would set); a re-queue child runs alone
    (WORLD_SIZE=1) with CATTUS_LOCAL_DEVICE naming the dead rank's device, and takes its games from the list file.
    Progress files: <work_dir>/progress/rank<r>.txt and requeue<r>_<attempt>.txt; a rank writes <work_dir>/rank<r>.json
    (its own counters) when its games are done.

    A work directory serves ONE round: state of an earlier one (`stale_state`) is refused (stale="refuse", the default) or removed
    first (stale="wipe"); nothing of it is ever pooled.  rank_timeout: seconds after which ranks still running are killed and their
    unfinished games re-queued (a rank wedged in a GPU call never exits by itself)."""
    work = Path(work_dir)
    old = stale_state(work)
    if old and stale == "wipe":
        wipe_state(work)
    elif old:
        raise RuntimeError(f"{work} holds the state of an earlier round ({', '.join(old[:6])}{' ...' if len(old) > 6 else ''}): "
                           "a work directory serves one round -- use a fresh one, or stale='wipe' / --clean")
    (work / "progress").mkdir(parents=True, exist_ok=True)
    flag = work / ABORT_FLAG
    port = free_port()
    base_env = dict(os.environ)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base_env.update(env_extra or {})
    procs = {}
    for r in range(world):
        env = dict(base_env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CATTUS_SUPERVISED="1")
        procs[r] = subprocess.Popen(rank_cmd(r, None), env=env, stdout=sys.stderr)  # stdout carries the supervisor's one line
    t0 = time.monotonic()
    status: dict[int, int] = {}
    while len(status) < world:
        for r, p in procs.items():
            if r in status:
                continue
            rc = p.poll()
            if rc is None:
                continue
            status[r] = rc
            if rc != 0 and not flag.exists():
                print(f"supervisor: rank {r} exited with status {rc}: survivors will skip the pooling collective", file=log, flush=True)
                flag.write_text(f"rank {r} status {rc}\n")
        if rank_timeout is not None and time.monotonic() - t0 > rank_timeout:
            for r, p in procs.items():
                if r not in status:
                    print(f"supervisor: rank {r} exceeded {rank_timeout:.0f} s: killed", file=log, flush=True)
                    p.kill()
                    p.wait()
                    status[r] = -9
            flag.write_text("timeout\n")
        time.sleep(0.05)
    # a rank is complete when it reported its counters (written behind its last game: all of them are on disk) and either exited
    # cleanly or was only stopped while waiting for a dead peer in the pooling collective (killed at the rank timeout, a failed
    # collective): what it was there to do is done, nothing of its shard is re-queued
    done0 = read_progress(sorted((work / "progress").glob("*.txt")))
    failed = [r for r in range(world)
              if not (work / f"rank{r}.json").exists() or (status[r] != 0 and any(g not in done0 for g in shard_of(r, world, games_num)))]
    requeued: list[int] = []
    attempts = {r: 0 for r in failed}
    pending = list(failed)
    while pending:
        r = pending.pop(0)
        done = read_progress(sorted((work / "progress").glob("*.txt")))
        todo = [g for g in shard_of(r, world, games_num) if g not in done]
        if not todo:
            continue
        if attempts[r] >= max_requeues:
            raise RuntimeError(f"rank {r}: {len(todo)} games still unfinished after {max_requeues} re-queues")
        attempts[r] += 1
        lst = work / f"requeue{r}_{attempts[r]}.list"
        lst.write_text("\n".join(map(str, todo)) + "\n")
        print(f"supervisor: re-queueing {len(todo)} unfinished games of rank {r} on a fresh process (attempt {attempts[r]})", file=log, flush=True)
        env = dict(base_env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(free_port()), CATTUS_SUPERVISED="1", CATTUS_LOCAL_DEVICE=str(r),
                   CATTUS_REQUEUE_TAG=f"requeue{r}_{attempts[r]}")
        for k in ("CATTUS_FAULT_RANK", "CATTUS_FAULT_AFTER_GAMES", "CATTUS_HANG_RANK"):  # an injected fault (tests) belongs to the first life only
            env.pop(k, None)
        rc = subprocess.call(rank_cmd(r, str(lst)), env=env, stdout=sys.stderr)
        if rc != 0:
            print(f"supervisor: the re-queue of rank {r} exited with status {rc}", file=log, flush=True)
            pending.append(r)
        # only games the fresh process actually finished count as re-queued
        done_after = read_progress(sorted((work / "progress").glob("*.txt")))
        requeued += [g for g in todo if g in done_after]
    done = read_progress(sorted((work / "progress").glob("*.txt")))
    missing = [g for g in range(games_num) if g not in done]
    if missing:
        raise RuntimeError(f"{len(missing)} games unfinished after supervision: {missing[:8]}...")
    tally = [0, 0, 0]
    positions = adjudicated = 0
    for plies, t, adj in done.values():
        tally[t] += 1
        positions += plies
        adjudicated += adj
    counters = {"node_evals": 0, "activation_count": 0, "cache_hits": 0, "cache_misses": 0}
    for f in sorted(work.glob("rank*.json")) + sorted(work.glob("requeue*.json")):
        try:
            c = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        for k in counters:
            counters[k] += int(c.get(k, 0))
    return {
        "player1_wins": tally[1], "player2_wins": tally[2], "draws": tally[0], "positions": positions, "adjudicated": adjudicated,
        "games": games_num, "ranks": world, "failed_ranks": failed, "requeued_games": sorted(requeued),
        "rank_status": {str(r): status[r] for r in range(world)},
        # evaluations of a process that died are not counted: it took its counters with it
        **counters,
        "pooled_via": "files" if failed else "collective",
    }
