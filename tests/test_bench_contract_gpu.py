"""bench.py's output contract: stdout is exactly ONE line, a JSON object with the driver's keys plus `roofline` and
`cpu_baseline`; everything else (RCCL's version banner included -- the config-4 leg brings a communicator up even on one
rank) goes to stderr."""

import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_bench_prints_one_json_line_on_stdout():
    cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "2", "--settle-seconds", "0", "--selfplay-seconds", "3",
           "--agreement-plies", "0", "--no-bf16", "--no-f16", "--no-f32", "--lanes", "1"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[:2000]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 6 and out["dtype"] == "f16x2" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] in ("mfma", "l2_delivery") and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel"] == "tower_wino4_kernel"  # batch 256 on 8x8 boards: the split tower in Winograd form, every layer in one launch
    assert r["delivery"]["workgroups"] == 256 and 0 < r["delivery"]["frac_of_roof"][0] < 1.5
    c = out["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert abs(out["value"] - 256 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-6


@pytest.mark.gpu
def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` with no torchrun in front: the parent (which has not touched the GPU) starts two fresh ranks
    and relays rank 0's line.  This box has one GPU, so the rehearsal switch puts both ranks on device 0 and runs the
    collectives over gloo -- the plumbing is what is under test, never the number."""
    import os

    env = dict(os.environ, BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--settle-seconds", "0", "--selfplay-seconds", "2",
           "--agreement-plies", "0", "--no-bf16", "--no-f16", "--no-f32", "--lanes", "1", "--no-smi"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[:2000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["per_rank_value"]) == 2 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 256 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-6
    assert out["process_group"]["ranks"] == 2
    for leg in ("selfplay", "selfplay_full_games", "selfplay_config4"):
        assert out[leg]["ranks"] == 2 and len(out[leg]["node_evals_per_sec_per_rank"]) == 2, leg
    assert out["selfplay_config4"]["pooled"]["records"] == out["selfplay_config4"]["games"] * 6


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Under a launcher, WORLD_SIZE != --gpus exits non-zero before anything is measured (no GPU needed to see it)."""
    import os

    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and p.stdout.strip() == ""


def test_bench_gpus_n_fails_loudly_where_the_ranks_cannot_run():
    """`--gpus 2` on a box without a GPU: the parent starts the ranks, they refuse (no CPU path), and the parent exits
    non-zero with an empty stdout instead of a one-rank line."""
    import os

    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and p.stdout.strip() == "" and "starting 2 ranks" in p.stderr


@pytest.mark.gpu
def test_sustained_mfma_rate_is_a_sane_fraction_of_the_nominal_peak():
    """cattus_hip_mfma_sustained (the `roofline.sustained` figure): between 40 % and 100 % of the nominal peak for every tower
    kind, and the exact-f32 MFMA -- the least power-hungry -- closest to its own."""
    import numpy as np  # noqa: F401

    from cattus_amd.evaluator import HipEvaluator
    from cattus_amd.weights import TTT, NetDesc, seeded_blob

    blob = seeded_blob(NetDesc(**TTT, blocks=1, filters=64, vhc=8, phc=8), 3)
    frac = {}
    for dtype, nominal in (("f16x2", 2500.0), ("bf16", 2500.0), ("f32", 157.3)):
        with HipEvaluator(blob, batch_size=4, plane_words=1, dtype=dtype) as ev:
            frac[dtype] = ev.mfma_sustained(0.3) / nominal
        assert 0.4 < frac[dtype] <= 1.02, (dtype, frac[dtype])
    assert frac["f32"] > frac["f16x2"]
