#!/usr/bin/env python3
"""Per-leaf error of the evaluator's dtypes against the float64 run of the reference network (tests/golden), and
the time of a chess 20x256 batch in each.  GPU box:  python scripts/split_check.py [steps]"""
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import blob_for, golden_names, outputs_equal_ref_tol  # noqa: E402

from cattus_amd.evaluator import HipEvaluator  # noqa: E402

out = {"fixtures": {}}
for name in golden_names():
    d, blob, z = blob_for(name)
    planes = z["planes"]
    row = {}
    p64, v64 = z["policy_f64"], z["value_f64"]
    row["reference_f32"] = dict(dp=float(np.abs(z["policy"] - p64).max()), dv=float(np.abs(z["value"] - v64).max()))
    for dtype in ("f32", "f16x2", "f16", "bf16"):
        with HipEvaluator(blob, batch_size=len(planes), plane_words=planes.shape[2], dtype=dtype) as ev:
            p, v = ev.eval(planes)
        row[dtype] = dict(dp=float(np.abs(p - p64).max()), dv=float(np.abs(v - v64).max()),
                          dp_vs_ref=float(np.abs(p - z["policy"]).max()), dv_vs_ref=float(np.abs(v - z["value"]).max()),
                          within_reference_tolerance=bool(outputs_equal_ref_tol(p, v, z["policy"], z["value"])))
    out["fixtures"][name] = row
    print(name, json.dumps(row), flush=True)

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

d, blob, planes = bench.make_workload("chess20x256")
dev = torch.device("cuda", 0)
d_planes = torch.from_numpy(planes.view(np.int64)).to(dev)
stream = torch.cuda.Stream(device=dev)
out["timing"] = {}
for dtype in ("bf16", "f16", "f16x2", "f32"):
    ev = HipEvaluator(blob, batch_size=256, plane_words=1, dtype=dtype)
    pol = torch.empty((256, d.moves), dtype=torch.float32, device=dev)
    val = torch.empty((256,), dtype=torch.float32, device=dev)
    n = steps if dtype != "f32" else max(5, steps // 10)
    for _ in range(max(3, n // 5)):
        ev.eval_device(d_planes.data_ptr(), 256, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ev.eval_device(d_planes.data_ptr(), 256, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    us, launches = ev.time_tower(256, 10)
    out["timing"][dtype] = dict(ms_per_batch=ms, node_evals_per_s=256 / ms * 1e3, conv_launch_us=us, launches=launches)
    print(dtype, json.dumps(out["timing"][dtype]), flush=True)
    ev.close()
print(json.dumps(out))
