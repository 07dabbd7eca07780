"""integration/c/consumer.c on the GPU box: the Rust binding's call sequence (create -> N threads blocking in
cattus_hip_apply, one leaf each -> stats -> destroy) from plain C, no ctypes -- against the Python binding's results."""

import json
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

from cattus_amd import synth
from cattus_amd.evaluator import HipEvaluator
from cattus_amd.weights import NetDesc, hex_game, seeded_blob

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("dtype,code", [("f16x2", 2), ("f32", 0)])
def test_c_consumer_replays_the_binding(tmp_path, dtype, code):
    exe = tmp_path / "consumer"
    libdir = ROOT / "cattus_amd"
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-D_POSIX_C_SOURCE=200809L", f"-I{ROOT / 'include'}", str(ROOT / "integration" / "c" / "consumer.c"),
                           "-o", str(exe), f"-L{libdir}", "-lcattus_hip", "-lpthread", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"])
    d = NetDesc(**hex_game(7), blocks=2, filters=64, vhc=16, phc=16)
    blob = seeded_blob(d, 3)
    planes = synth.random_hex_planes(50, 7, 4)
    (tmp_path / "model.cattus").write_bytes(blob)
    planes.tofile(tmp_path / "planes.bin")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe), str(tmp_path / "model.cattus"), str(tmp_path / "planes.bin"), str(tmp_path / "out.bin"), str(code), "16", "16", "2"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(50, d.moves + 1)
    with HipEvaluator(blob, batch_size=16, plane_words=2, dtype=dtype) as ev:
        want_p, want_v = ev.eval(planes[:16])
        rest = [ev.eval(planes[i : i + 16]) for i in range(16, 50, 16)]
    want_p = np.concatenate([want_p] + [r[0] for r in rest])
    want_v = np.concatenate([want_v] + [r[1] for r in rest])
    assert (got[:, : d.moves] == want_p).all() and (got[:, d.moves] == want_v).all()  # per-leaf results do not depend on the batch
    assert stats["leaves"] == 50 and stats["positions"] == 50 and 4 <= stats["batches"] <= 50 and stats["moves"] == d.moves
    assert stats["empty_eval_status"] == -1  # CATTUS_E_INVALID, not an abort
