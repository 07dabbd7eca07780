"""Self-play driver tests (host library with CPU stand-in networks; no GPU needed).

Covers training/self-play/src/self_play.rs:94-275 and self_play_cmd.rs:55-153: output file names and
directories, record contents, win counters, summary JSON, even games_num, and that results do not
depend on threads / batch size / number of concurrent games (the reference's schedule is one game
per thread; ours batches leaves across games)."""

import json
import os

import numpy as np
import pytest

from cattus_amd import records
from cattus_amd import selfplay as sp
from cattus_amd.weights import NetDesc, hex_game, seeded_blob
from oracle import mcts_oracle as mo
from oracle import oracle


def _cfg(**kw):
    base = dict(sim_num=30, temperature_policy=[(9999, 0.0)], cache_size=1000, batch_size=1, threads=1)
    base.update(kw)
    return sp.make_config(**base)


def test_games_num_must_be_even():
    with pytest.raises(RuntimeError, match="multiple of 2"):
        sp.run_self_play("tictactoe", _cfg(), sp.Net.stub("tictactoe"), None, 3)


def test_output_files_records_and_counters(tmp_path):
    d1, d2 = tmp_path / "d1", tmp_path / "d2"
    res = sp.run_self_play("tictactoe", _cfg(), sp.Net.stub("tictactoe"), None, 4, d1, d2)
    assert res["player1_wins"] + res["player2_wins"] + res["draws"] == 4
    files1, files2 = sorted(os.listdir(d1)), sorted(os.listdir(d2))
    assert len(files1) + len(files2) == res["positions"] == res["records"]
    assert all(f.endswith(".traindata") and len(f) == len("00000000_000.traindata") for f in files1 + files2)
    # game 0, ply 0 has Player1 to move and an even game index -> out_dir1 (self_play.rs:256-259)
    assert "00000000_000.traindata" in files1 and "00000000_001.traindata" in files2
    assert "00000001_000.traindata" in files2 and "00000001_001.traindata" in files1
    # files hold exactly the in-memory records
    for rec, (g, p, d) in zip(res["record_bytes"], res["record_meta"]):
        path = (d1 if d == 0 else d2) / f"{g:08d}_{p:03d}.traindata"
        assert path.read_bytes() == rec.tobytes()
    # every record: Player1 to move, probabilities sum to 1 over legal moves, winner in {-1,0,1}
    for rec in res["record_bytes"]:
        e = records.parse_record("tictactoe", rec.tobytes())
        legal = e.probs >= 0
        assert abs(e.probs[legal].sum() - 1) < 1e-5
        occ = int(e.planes[0, 0]) | int(e.planes[1, 0])
        assert [bool(occ >> i & 1) for i in range(9)] == [not x for x in legal]
        assert e.winner in (-1.0, 0.0, 1.0)
        mine, theirs = bin(int(e.planes[0, 0])).count("1"), bin(int(e.planes[1, 0])).count("1")
        assert theirs - mine in (0, 1)  # flipped so that the side to move owns plane 0


def test_first_record_matches_search_trace():
    cfg = _cfg(sim_num=40)
    trace = sp.trace_game("hex4", cfg, sp.Net.stub("hex4"))
    res = sp.run_self_play("hex4", cfg, sp.Net.stub("hex4"), None, 2)
    recs = [(m, r) for m, r in zip(res["record_meta"], res["record_bytes"]) if m[0] == 0]
    assert len(recs) == len(trace)
    e0 = records.parse_record("hex4", recs[0][1].tobytes())
    visits = dict(trace[0][1])
    total = sum(visits.values())
    for m, n in visits.items():
        assert e0.probs[m] == np.float32(n) / np.float32(total)


@pytest.mark.parametrize("game", ["tictactoe", "hex4"])
def test_results_independent_of_schedule(game):
    ref = sp.run_self_play(game, _cfg(), sp.Net.stub(game), None, 8)
    for kw in (dict(threads=4, batch_size=4), dict(threads=3, batch_size=2, concurrent_games=8), dict(threads=2, batch_size=8, concurrent_games=5),
               dict(threads=3, batch_size=2, concurrent_games=8, eval_threads=1), dict(threads=2, batch_size=2, concurrent_games=8, eval_threads=4)):
        got = sp.run_self_play(game, _cfg(**kw), sp.Net.stub(game), None, 8)
        assert (got["record_meta"] == ref["record_meta"]).all()
        assert (got["record_bytes"] == ref["record_bytes"]).all()
        assert [got[k] for k in ("player1_wins", "player2_wins", "draws")] == [ref[k] for k in ("player1_wins", "player2_wins", "draws")]


def test_game_index_sharding_matches_single_process():
    # rank r of W plays games r, r+W, ...: the union equals the single-process run
    whole = sp.run_self_play("tictactoe", _cfg(), sp.Net.stub("tictactoe"), None, 8)
    parts = [sp.run_self_play("tictactoe", _cfg(first_game=r, game_stride=2), sp.Net.stub("tictactoe"), None, 4) for r in range(2)]
    meta = np.concatenate([p["record_meta"] for p in parts])
    recs = np.concatenate([p["record_bytes"] for p in parts])
    order = np.lexsort((meta[:, 1], meta[:, 0]))
    assert (meta[order] == whole["record_meta"]).all() and (recs[order] == whole["record_bytes"]).all()
    assert sum(p["player1_wins"] for p in parts) == whole["player1_wins"]


def test_two_models_use_separate_networks():
    calls = {"a": 0, "b": 0}

    def mk(tag, bias):
        def net(planes):
            calls[tag] += len(planes)
            n = len(planes)
            pol = np.tile(np.arange(9, dtype=np.float32) * bias, (n, 1))
            return pol, np.zeros(n, dtype=np.float32)

        return net

    res = sp.run_self_play("tictactoe", _cfg(sim_num=10), sp.Net.python(mk("a", 0.1)), sp.Net.python(mk("b", -0.1)), 2)
    assert calls["a"] > 0 and calls["b"] > 0
    assert res["node_evals"] == calls["a"] + calls["b"]


def test_summary_json_has_reference_metric_names(tmp_path):
    res = sp.run_self_play("tictactoe", _cfg(), sp.Net.stub("tictactoe"), None, 2)
    path = tmp_path / "summary.json"
    sp.write_summary(path, res)
    s = json.loads(path.read_text())
    assert set(s) == {"player1_wins", "player2_wins", "draws", "metrics"}
    for name in ("model.activation_count", "model.run_duration", "mcts.search_duration", "cache.hits", "cache.misses"):
        assert name in s["metrics"]
    assert s["metrics"]["cache.misses"] == res["node_evals"]
    with pytest.raises(FileExistsError):
        sp.write_summary(path, res)  # create_new semantics (self_play_cmd.rs:148)


def test_temperature_and_noise_paths_run():
    cfg = _cfg(sim_num=20, temperature_policy=[(2, 1.0), (9999, 0.0)], prior_noise_alpha=0.3, prior_noise_epsilon=0.25, seed=7)
    a = sp.run_self_play("hex4", cfg, sp.Net.stub("hex4"), None, 2)
    b = sp.run_self_play("hex4", cfg, sp.Net.stub("hex4"), None, 2)
    assert (a["record_bytes"] == b["record_bytes"]).all()  # seeded, unlike the reference's thread RNG
    assert a["positions"] > 0


def test_config1_plumbing_run_hex4_with_oracle_network():
    """BASELINE config 1 restated: hex4, ConvNetV1 7x16 (seeded), 100 sims/move, noise off,
    temperature 0, batch 1, 1 thread, 2 games -- on the CPU oracle network, cross-checked against the
    pure-Python search with the same network."""
    d = NetDesc(**hex_game(4), blocks=7, filters=16, vhc=16, phc=16)
    net = oracle.OracleNet(seeded_blob(d, 0))

    def eval_planes(planes):
        return net.forward(planes.reshape(len(planes), 3, 2), threads=1)

    cfg = sp.make_config(sim_num=100, explore_factor=1.41421, temperature_policy=[(9999, 0.0)], cache_size=1000, batch_size=1, threads=1)
    trace = sp.trace_game("hex4", cfg, sp.Net.python(eval_planes))

    def py_net(words, moves):
        p, v = net.forward(np.array(words, dtype=np.uint64).reshape(1, 3, 2), threads=1)
        return p[0], v[0]

    want, _ = mo.trace_game(mo.make_hex(4), 100, 1.41421, net=py_net)
    assert trace == want
    res = sp.run_self_play("hex4", cfg, sp.Net.python(eval_planes), None, 2)
    assert res["player1_wins"] + res["player2_wins"] + res["draws"] == 2
    assert len([m for m in res["record_meta"] if m[0] == 0]) == len(trace)


def _hashed_logits(planes, moves):
    # cheap deterministic "network": logits and value from the plane words
    h = (planes.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).sum(axis=1)
    k = np.arange(moves, dtype=np.uint64)[None, :]
    x = (h[:, None] ^ (k * np.uint64(0xC2B2AE3D27D4EB4F))) * np.uint64(0xD6E8FEB86659FD93)
    pol = ((x >> np.uint64(40)).astype(np.float32) / np.float32(1 << 24) - 0.5) * 6.0
    val = (((h >> np.uint64(40)).astype(np.float32)) / np.float32(1 << 24) - 0.5) * 0.6
    return pol.astype(np.float32), val.astype(np.float32)


@pytest.mark.parametrize("game,moves", [("tictactoe", 9), ("hex5", 25), ("chess", 1880)])
def test_legal_softmax_network_gives_the_same_games(game, moves):
    """A network that returns the softmax over the legal moves itself (cattus_net_eval_legal_fn, the
    host side of cattus_hip_eval_legal) must see the legal moves of the flipped position, in move order,
    and lead to the same records as the host's own calc_moves_probs (net/mod.rs:100-119)."""
    seen = {"leaves": 0, "max_cnt": 0}

    def plain(planes):
        return _hashed_logits(planes, moves)

    def legal(planes, idx, cnt):
        pol, val = _hashed_logits(planes, moves)
        probs = np.zeros(idx.shape, dtype=np.float32)
        for i in range(len(planes)):
            c = int(cnt[i])
            assert 0 < c <= idx.shape[1] and len(set(idx[i, :c].tolist())) == c
            probs[i, :c] = oracle.softmax_legal(pol[i], idx[i, :c].astype(np.uint32))
            seen["leaves"] += 1
            seen["max_cnt"] = max(seen["max_cnt"], c)
        return probs, val

    sims = 12 if game == "chess" else 30
    kw = dict(sim_num=sims, batch_size=4, threads=2, concurrent_games=4)
    a = sp.run_self_play(game, _cfg(**kw), sp.Net.python(plain), None, 2)
    b = sp.run_self_play(game, _cfg(**kw), sp.Net.python_legal(legal), None, 2)
    assert (a["record_meta"] == b["record_meta"]).all()
    assert (a["record_bytes"] == b["record_bytes"]).all()
    assert a["node_evals"] == b["node_evals"] == seen["leaves"]
    assert seen["max_cnt"] <= moves


def test_noisy_games_do_not_depend_on_schedule_or_sharding():
    """With temperature sampling and Dirichlet noise on, the random streams are seeded per game (not per
    slot or thread), so the same games come out of any schedule and any sharding."""
    kw = dict(sim_num=20, temperature_policy=[(4, 1.0), (9999, 0.0)], prior_noise_alpha=0.3, prior_noise_epsilon=0.25, cache_size=10000)
    ref = sp.run_self_play("hex5", _cfg(**kw), sp.Net.stub("hex5"), None, 12)
    for extra in (dict(threads=4, batch_size=8, concurrent_games=5), dict(threads=3, batch_size=4, concurrent_games=12, eval_threads=1)):
        got = sp.run_self_play("hex5", _cfg(**kw, **extra), sp.Net.stub("hex5"), None, 12)
        assert (got["record_meta"] == ref["record_meta"]).all()
        assert (got["record_bytes"] == ref["record_bytes"]).all()
    parts = [sp.run_self_play("hex5", _cfg(**kw, threads=2, batch_size=4, first_game=r, game_stride=2), sp.Net.stub("hex5"), None, 6) for r in range(2)]
    meta = np.concatenate([p["record_meta"] for p in parts])
    rec = np.concatenate([p["record_bytes"] for p in parts])
    order = np.lexsort((meta[:, 2], meta[:, 1], meta[:, 0]))
    rorder = np.lexsort((ref["record_meta"][:, 2], ref["record_meta"][:, 1], ref["record_meta"][:, 0]))
    assert (meta[order] == ref["record_meta"][rorder]).all()
    assert (rec[order] == ref["record_bytes"][rorder]).all()
    # and games differ from each other (the noise is really on)
    firsts = {ref["record_bytes"][i].tobytes() for i in range(len(ref["record_bytes"])) if ref["record_meta"][i][1] == 1}
    assert len(firsts) > 1


@pytest.mark.parametrize("game", ["hex5", "chess"])
def test_leaves_in_flight_fills_batches_and_stays_schedule_independent(game):
    """mcts.leaves_in_flight > 1 (virtual loss; not in the reference): one tree keeps several leaves at the
    network, so a handful of games fills batches; the games are still a pure function of the configuration
    (whatever the threads / batch size / evaluation threads), every record is a proper distribution over legal
    moves, and the default (1) remains the sequential search."""
    sims = 40 if game == "hex5" else 24
    base = dict(sim_num=sims, cache_size=100000, concurrent_games=4)
    seq = sp.run_self_play(game, _cfg(**base, threads=2, batch_size=32), sp.Net.stub(game), None, 4)
    a = sp.run_self_play(game, _cfg(**base, threads=2, batch_size=32, leaves_in_flight=8), sp.Net.stub(game), None, 4)
    b = sp.run_self_play(game, _cfg(**base, threads=4, batch_size=5, eval_threads=1, leaves_in_flight=8), sp.Net.stub(game), None, 4)
    assert (a["record_meta"] == b["record_meta"]).all() and (a["record_bytes"] == b["record_bytes"]).all()
    # 4 games alone give at most 4 leaves per batch; 8 leaves per tree give clearly more
    fill_seq = seq["node_evals"] / seq["activation_count"]
    fill_par = a["node_evals"] / a["activation_count"]
    assert fill_seq <= 4.0 and fill_par > 1.5 * fill_seq, (fill_seq, fill_par)
    for rec in a["record_bytes"][:40]:
        e = records.parse_record(game, rec.tobytes())
        legal = e.probs >= 0
        assert legal.sum() >= 1 and abs(e.probs[legal].sum() - 1) < 1e-4
    assert a["player1_wins"] + a["player2_wins"] + a["draws"] == 4


@pytest.mark.timeout(120)
@pytest.mark.parametrize("batch,lif", [(1, 8), (2, 16), (3, 16)])
def test_small_batches_with_many_leaves_in_flight_do_not_stall(batch, lif):
    """One slot's leaves may outnumber a whole fixed-size ring of batch buffers (batch_size 1 is the
    make_config default): the ring is sized from the configuration, so the run completes and gives the
    games of a roomy schedule."""
    base = dict(sim_num=40, cache_size=100000, concurrent_games=4, leaves_in_flight=lif)
    ref = sp.run_self_play("hex5", _cfg(**base, threads=2, batch_size=64), sp.Net.stub("hex5"), None, 4)
    got = sp.run_self_play("hex5", _cfg(**base, threads=2, batch_size=batch), sp.Net.stub("hex5"), None, 4)
    assert (got["record_meta"] == ref["record_meta"]).all() and (got["record_bytes"] == ref["record_bytes"]).all()


def test_config_validation_follows_the_reference_asserts():
    """TemperaturePolicy::scheduled and MctsPlayer::new assert these (mcts/mod.rs:106-110,470-474); the C ABI
    reports them instead of clamping."""
    net = sp.Net.stub("tictactoe")
    bad = _cfg()
    bad.temperature_count = 0
    with pytest.raises(RuntimeError, match="temperature_count"):
        sp.run_self_play("tictactoe", bad, net, None, 2)
    bad = _cfg()
    bad.temperature_count = 9
    with pytest.raises(RuntimeError, match="temperature_count"):
        sp.run_self_play("tictactoe", bad, net, None, 2)
    with pytest.raises(RuntimeError, match="strictly increasing"):
        sp.run_self_play("tictactoe", _cfg(temperature_policy=[(5, 1.0), (5, 0.5), (9999, 0.0)]), net, None, 2)
    with pytest.raises(RuntimeError, match=">= 0"):
        sp.run_self_play("tictactoe", _cfg(temperature_policy=[(5, -1.0), (9999, 0.0)]), net, None, 2)
    with pytest.raises(RuntimeError, match="epsilon"):
        sp.run_self_play("tictactoe", _cfg(prior_noise_epsilon=1.5), net, None, 2)
    with pytest.raises(RuntimeError, match="temperature_count"):
        sp.trace_game("tictactoe", bad, net)
    # the last entry's threshold is unused (only its temperature is): it need not be increasing
    sp.run_self_play("tictactoe", _cfg(temperature_policy=[(5, 1.0), (0, 0.0)]), net, None, 2)


def test_cache_is_one_fifo_of_max_size():
    """ValueFuncCache keeps ONE deque of max_size positions (mcts/cache.rs:62-70).  Sequential search, one
    thread: the hit / miss counters are then a pure function of the evaluation order, and they must equal a
    replay of that order through a plain FIFO of the same size."""
    seen = []

    def net(planes):
        seen.extend(p.tobytes() for p in planes)
        return _hashed_logits(planes, 25)

    for size in (1, 7, 50):
        seen.clear()
        res = sp.run_self_play("hex5", _cfg(sim_num=30, cache_size=size), sp.Net.python(net), None, 2)
        assert res["cache_misses"] == res["node_evals"] == len(seen)
        # replay: every network call was a miss and inserted its position (planes identify the flipped position)
        from collections import deque

        fifo, live = deque(), set()
        for key in seen:
            if key in live:
                raise AssertionError("a cached position went to the network")
            while len(fifo) >= size:
                live.discard(fifo.popleft())
            fifo.append(key)
            live.add(key)
        big = sp.run_self_play("hex5", _cfg(sim_num=30, cache_size=100000), sp.Net.python(net), None, 2)
        assert (big["record_bytes"] == res["record_bytes"]).all()  # results never depend on the cache
        assert big["cache_misses"] <= res["cache_misses"]


def test_self_player_executables_exist_for_every_game_and_start_from_any_directory(tmp_path):
    """bin/<game>_self_player (the names training/cattus_train/self_play.py:44 builds): present for every game the
    reference has a binary for, executable, and importable from a foreign working directory (`--help` needs no GPU)."""
    import subprocess
    from pathlib import Path

    root = Path(sp.__file__).resolve().parent.parent
    for game in ("tictactoe", "hex4", "hex5", "hex7", "hex9", "hex11", "hex", "chess"):
        exe = root / "bin" / f"{game}_self_player"
        assert exe.exists() and exe.stat().st_mode & 0o111, exe
    out = subprocess.run([str(root / "bin" / "chess_self_player"), "--help"], cwd=str(tmp_path), capture_output=True, text=True)
    assert out.returncode == 0 and "--model1-path" in out.stdout and "--summary-file" in out.stdout
    # a missing required flag is an argument error (clap exits non-zero as well)
    bad = subprocess.run([str(root / "bin" / "hex4_self_player"), "--games-num=2"], cwd=str(tmp_path), capture_output=True, text=True)
    assert bad.returncode != 0
