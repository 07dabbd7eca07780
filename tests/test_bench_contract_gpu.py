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
           "--agreement-plies", "0", "--no-bf16", "--no-f32", "--lanes", "1"]
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
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["kernel"] == "conv3x3_splitw_kernel"
    c = out["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert abs(out["value"] - 256 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-6


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
