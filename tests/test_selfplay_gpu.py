"""End-to-end GPU tests: the C++ self-play driver calling the HIP evaluator through the raw C function
pointer, checked against the same search running on the CPU oracle network."""

import json
import subprocess
import sys

import numpy as np
import pytest

from cattus_amd import records
from cattus_amd import selfplay as sp
from cattus_amd.evaluator import HipEvaluator
from cattus_amd.weights import CHESS, NetDesc, hex_game, seeded_blob
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_visit_distributions_identical_with_hip_f32_and_oracle_network():
    """north_star: 'move-visit distributions bit-identical under fixed seed + greedy argmax'."""
    d = NetDesc(**hex_game(7), blocks=6, filters=64, vhc=16, phc=16)
    blob = seeded_blob(d, 1)
    cfg = sp.make_config(sim_num=60, temperature_policy=[(9999, 0.0)], cache_size=10000, batch_size=1, threads=1)
    net = oracle.OracleNet(blob)
    want = sp.trace_game("hex7", cfg, sp.Net.python(lambda pl: net.forward(pl.reshape(len(pl), 3, 2), threads=1)), max_plies=8)
    with HipEvaluator(blob, batch_size=16, plane_words=2, dtype="f32") as ev:
        got = sp.trace_game("hex7", cfg, sp.Net.hip(ev), max_plies=8)
        assert got == want
        # and whole games through the batched driver: identical records whatever the batching
        a = sp.run_self_play("hex7", sp.make_config(sim_num=30, batch_size=16, threads=4, concurrent_games=16), sp.Net.hip(ev), None, 8)
        b = sp.run_self_play("hex7", sp.make_config(sim_num=30, batch_size=3, threads=2, concurrent_games=5), sp.Net.hip(ev), None, 8)
        assert (a["record_bytes"] == b["record_bytes"]).all() and (a["record_meta"] == b["record_meta"]).all()
        assert a["node_evals"] > 0 and a["activation_count"] < a["node_evals"]  # leaves were batched


def test_chess_self_play_on_bf16_evaluator():
    d = NetDesc(**CHESS, blocks=2, filters=64, vhc=8, phc=8)
    blob = seeded_blob(d, 31)
    with HipEvaluator(blob, batch_size=8, plane_words=1, dtype="bf16") as ev:
        cfg = sp.make_config(sim_num=16, batch_size=8, threads=4, concurrent_games=8, cache_size=100000)
        res = sp.run_self_play("chess", cfg, sp.Net.hip(ev), None, 8)
        again = sp.run_self_play("chess", cfg, sp.Net.hip(ev), None, 8)
    assert res["player1_wins"] + res["player2_wins"] + res["draws"] == 8
    assert (res["record_bytes"] == again["record_bytes"]).all()  # per-leaf results are batch-independent
    for rec in res["record_bytes"][:50]:
        e = records.parse_record("chess", rec.tobytes())
        legal = e.probs >= 0
        assert 1 <= legal.sum() <= 225 and abs(e.probs[legal].sum() - 1) < 1e-4
        assert int(e.planes[17, 0]) == 2**64 - 1


def test_self_player_cli_contract(tmp_path):
    """Same flags / files as the reference's <game>_self_player (self_play_cmd.rs:15-32,135-149)."""
    d = NetDesc(**hex_game(5), blocks=1, filters=64, vhc=4, phc=4)
    model = tmp_path / "model.cattus"
    model.write_bytes(seeded_blob(d, 9))
    cfg = {
        "model": {"inference": {"engine": "hip", "device": 0, "dtype": "f32"}, "batch_size": 8},
        "mcts": {"sim_num": 12, "explore_factor": 1.41421, "temperature_policy": [[9999, 0.0]], "prior_noise_alpha": 0.0,
                 "prior_noise_epsilon": 0.0, "cache_size": 1000},
        "threads": 2,
    }
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    cmd = [sys.executable, "-m", "cattus_amd.selfplay", "--game", "hex5", "--model1-path", str(model), "--model2-path", str(model),
           "--games-num", "4", "--out-dir1", str(tmp_path / "o1"), "--out-dir2", str(tmp_path / "o2"),
           "--summary-file", str(tmp_path / "summary.json"), "--config-file", str(tmp_path / "cfg.json")]
    subprocess.check_call(cmd, cwd=str(sp._PKG.parent))
    s = json.loads((tmp_path / "summary.json").read_text())
    assert s["player1_wins"] + s["player2_wins"] + s["draws"] == 4
    assert s["metrics"]["model.activation_count"] > 0 and s["metrics"]["cache.misses"] > 0
    n = len(list((tmp_path / "o1").iterdir())) + len(list((tmp_path / "o2").iterdir()))
    assert n > 0
    for f in (tmp_path / "o1").iterdir():
        assert f.stat().st_size == records.record_nbytes("hex5")


def test_self_player_executable_as_the_trainer_launches_it(tmp_path):
    """bin/<game>_self_player exactly as training/cattus_train/train_process.py:159-186 runs the reference's binary:
    `--flag=value` arguments, no --game, a foreign working directory, the engine JSON written with json.dump, and the
    summary read back through the same keys the trainer reads (default dtype: the split-precision tower)."""
    d = NetDesc(**hex_game(5), blocks=1, filters=64, vhc=4, phc=4)
    model_dir = tmp_path / "models" / "model_x" / "self_play"
    model_dir.mkdir(parents=True)
    model_path = model_dir / "model.cattus"
    model_path.write_bytes(seeded_blob(d, 9))
    engine_cfg = {
        "model": {"inference": {"engine": "hip"}, "batch_size": 8},
        "mcts": {"sim_num": 12, "explore_factor": 1.41421, "temperature_policy": [[2, 1.0], [9999, 0.0]], "prior_noise_alpha": 0.3,
                 "prior_noise_epsilon": 0.25, "cache_size": 1000},
        "threads": 2,
    }
    cfg_file = tmp_path / "config.json"
    cfg_file.write_text(json.dumps(engine_cfg, indent=2))
    summary_file = tmp_path / "selfplay_summary.json"
    data_entries_dir = tmp_path / "games" / "run" / "250101_000000_000000"
    foreign_cwd = tmp_path / "self-play-crate"
    foreign_cwd.mkdir()
    exe = sp._PKG.parent / "bin" / "hex5_self_player"
    subprocess.check_call(
        [
            str(exe),
            f"--model1-path={model_path}",
            f"--model2-path={model_path}",
            "--games-num=4",
            f"--out-dir1={data_entries_dir}",
            f"--out-dir2={data_entries_dir}",
            f"--summary-file={summary_file}",
            f"--config-file={cfg_file}",
        ],
        cwd=str(foreign_cwd),
    )
    summary = json.loads(summary_file.read_text())
    metrics = {  # train_process.py:176-186
        "net_activations_count": summary["metrics"]["model.activation_count"],
        "net_run_duration_average_us": summary["metrics"]["model.run_duration"],
        "search_duration": summary["metrics"]["mcts.search_duration"],
        "cache_hit_ratio": summary["metrics"]["cache.hits"] / (summary["metrics"]["cache.hits"] + summary["metrics"]["cache.misses"]),
    }
    assert metrics["net_activations_count"] > 0 and metrics["net_run_duration_average_us"] > 0 and metrics["search_duration"] > 0
    assert summary["player1_wins"] + summary["player2_wins"] + summary["draws"] == 4
    files = sorted(data_entries_dir.iterdir())
    assert files and all(f.name.endswith(".traindata") and f.stat().st_size == records.record_nbytes("hex5") for f in files)
    # the summary file is created with create_new (self_play_cmd.rs:139-143): a second run on the same path fails
    assert subprocess.call([str(exe), f"--model1-path={model_path}", f"--model2-path={model_path}", "--games-num=2",
                            f"--out-dir1={data_entries_dir}", f"--out-dir2={data_entries_dir}", f"--summary-file={summary_file}",
                            f"--config-file={cfg_file}"], cwd=str(foreign_cwd), stderr=subprocess.DEVNULL) != 0


@pytest.mark.parametrize("game,desc,words", [("hex7", dict(**hex_game(7), blocks=2, filters=64, vhc=16, phc=16), 2), ("chess", dict(**CHESS, blocks=2, filters=64, vhc=8, phc=8), 1)])
def test_device_softmax_games_equal_host_restatement(game, desc, words):
    """cattus_hip_eval_legal inside the driver: the records equal those of a driver whose network is
    'HIP logits + the oracle's restatement of the device softmax' (bit-exact), and stay within 1e-6 of
    the probabilities the host's libm softmax gives."""
    d = NetDesc(**desc)
    blob = seeded_blob(d, 5)
    cfg = sp.make_config(sim_num=20, batch_size=8, threads=3, concurrent_games=8, cache_size=100000)
    with HipEvaluator(blob, batch_size=8, plane_words=words, dtype="f32") as ev:

        def restated(planes, idx, cnt):
            pol, val = ev.eval(planes.reshape(len(planes), d.planes, words))
            probs = np.zeros(idx.shape, dtype=np.float32)
            for i in range(len(planes)):
                probs[i, : cnt[i]] = oracle.softmax_legal_det(pol[i], idx[i, : cnt[i]])
            return probs, val

        got = sp.run_self_play(game, cfg, sp.Net.hip(ev, device_softmax=True), None, 4)
        want = sp.run_self_play(game, cfg, sp.Net.python_legal(restated), None, 4)
    assert got["node_evals"] == want["node_evals"] > 0
    assert (got["record_meta"] == want["record_meta"]).all()
    assert (got["record_bytes"] == want["record_bytes"]).all()


@pytest.mark.parametrize("game,size", [("tictactoe", 3), ("hex4", 4), ("hex5", 5), ("hex7", 7), ("hex9", 9), ("hex11", 11), ("chess", 8)])
def test_full_loop_smoke_every_game(game, size):
    """The reference's smoke loops (training/tests/test_convnetv1.py:10-58): every game with a tiny ConvNetV1,
    sim_num 10, noise off -- here through the HIP evaluator (16 filters padded to 64; boards above 8x8 on 128 pixel slots) and
    checked for well-formed records and against the same run on the CPU oracle network."""
    info = sp.game_info(game)
    d = NetDesc(planes=info["planes"], board=info["board"], moves=info["moves"], blocks=2, filters=16, vhc=4, phc=4)
    blob = seeded_blob(d, 13)
    words = info["plane_words"]
    cfg = sp.make_config(sim_num=10, batch_size=4, threads=2, concurrent_games=4, cache_size=1000)
    net = oracle.OracleNet(blob)
    want = sp.run_self_play(game, cfg, sp.Net.python(lambda pl: net.forward(pl.reshape(len(pl), d.planes, words), threads=1)), None, 2)
    with HipEvaluator(blob, batch_size=4, plane_words=words, dtype="f32") as ev:
        got = sp.run_self_play(game, cfg, sp.Net.hip(ev), None, 2)
    assert got["player1_wins"] + got["player2_wins"] + got["draws"] == 2
    assert (got["record_meta"] == want["record_meta"]).all()
    assert (got["record_bytes"] == want["record_bytes"]).all()  # f32 mode is bit-exact against the oracle
    for rec in got["record_bytes"][:10]:
        e = records.parse_record(game, rec.tobytes())
        legal = e.probs >= 0
        assert legal.sum() >= 1 and abs(e.probs[legal].sum() - 1) < 1e-4


def test_config4_shape_64_games_with_rccl_pooling():
    """BASELINE config 4's per-GPU shape as a test: chess 20x256, 64 concurrent games on this GPU (sequential search: at
    most 64 leaves per batch), the records all-gathered and the counters all-reduced through torch.distributed's `nccl`
    backend (= RCCL) -- with the one rank this box has; the 2-rank path is tests/test_dist.py on gloo."""
    import os

    import torch
    import torch.distributed as dist

    from cattus_amd import dist as cdist

    d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
    blob = seeded_blob(d, 2)
    started = not dist.is_initialized()
    if started:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29300 + os.getpid() % 500}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        first, stride, local_games = cdist.shard_games(64, dist.get_rank(), dist.get_world_size())
        with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2") as ev:
            cfg = sp.make_config(sim_num=100, batch_size=256, threads=8, concurrent_games=64, cache_size=1000000, first_game=first,
                                 game_stride=stride, seed=1, max_game_plies=3, temperature_policy=[(30, 1.0), (9999, 0.0)],
                                 prior_noise_alpha=0.03, prior_noise_epsilon=0.25)
            res = sp.run_self_play("chess", cfg, sp.Net.hip(ev), None, local_games)
            st = ev.stats()
        dev = torch.device("cuda", 0)
        recs, meta = cdist.pool_records(res["record_bytes"], res["record_meta"], device=dev)
        tot = cdist.reduce_counters(res, device=dev)
        assert dist.get_backend() == "nccl"
        assert len(recs) == tot["positions"] == 64 * 3 and recs.shape[1] == records.record_nbytes("chess")
        assert tot["player1_wins"] + tot["player2_wins"] + tot["draws"] == 64
        assert sorted(set(meta[:, 0].tolist())) == list(range(64))  # every global game index once per ply
        assert st["positions"] / st["batches"] <= 64.0  # one leaf per tree in flight: a batch holds at most 64 leaves
        e = records.parse_record("chess", recs[0].tobytes())
        assert abs(e.probs[e.probs >= 0].sum() - 1) < 1e-4
    finally:
        if started:
            dist.destroy_process_group()
