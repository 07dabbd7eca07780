#!/usr/bin/env python3
"""The f16x2 tower with its 256-filter layers in Winograd form (CATTUS_WINOGRAD=1) against the direct f16x2 tower, the exact-f32
tower and the reference network's float64 run; and the launch time of both (GPU box):  python scripts/wino_check.py [workload]"""
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np
import torch  # noqa: F401

import bench
from cattus_amd.evaluator import HipEvaluator
from helpers import blob_for, outputs_equal_ref_tol


def run(blob, planes, dtype, wino, words=1):
    os.environ["CATTUS_WINOGRAD"] = "1" if wino else "0"
    with HipEvaluator(blob, batch_size=len(planes), plane_words=words, dtype=dtype) as ev:
        p, v = ev.eval(planes)
        p2, v2 = ev.eval(planes)
        assert (p == p2).all() and (v == v2).all(), "not reproducible"
        us, launches = ev.time_tower(len(planes), 20)
        sat = ev.stats()["saturated"]
    return p, v, us, launches, sat


out = {}
d, blob, z = blob_for("chess_20x256")
pw, vw, *_ = run(blob, z["planes"], "f16x2", True)
pd, vd, *_ = run(blob, z["planes"], "f16x2", False)
out["fixture chess_20x256"] = dict(
    wino_vs_f64=dict(dp=float(np.abs(pw - z["policy_f64"]).max()), dv=float(np.abs(vw - z["value_f64"]).max())),
    direct_vs_f64=dict(dp=float(np.abs(pd - z["policy_f64"]).max()), dv=float(np.abs(vd - z["value_f64"]).max())),
    wino_within_reference_tolerance=bool(outputs_equal_ref_tol(pw, vw, z["policy"], z["value"])))
print(json.dumps(out), flush=True)
for wl in sys.argv[1:] or ["chess20x256", "chess40x384", "chess20x256_b128"]:
    d, blob, planes = bench.make_workload(wl)
    p32, v32, us32, *_ = run(blob, planes, "f32", False)
    pw, vw, usw, lw, satw = run(blob, planes, "f16x2", True)
    pd, vd, usd, ld, _ = run(blob, planes, "f16x2", False)
    out[wl] = dict(wino_vs_f32=dict(dp=float(np.abs(pw - p32).max()), dv=float(np.abs(vw - v32).max())),
                   direct_vs_f32=dict(dp=float(np.abs(pd - p32).max()), dv=float(np.abs(vd - v32).max())),
                   wino_within_reference_tolerance_of_f32=bool(outputs_equal_ref_tol(pw, vw, p32, v32)),
                   launch_us=dict(wino=usw, direct=usd, f32=us32), launches=lw, saturated=satw)
    print(wl, json.dumps(out[wl]), flush=True)
print(json.dumps(out))
