#!/usr/bin/env python3
"""Whole step (planes, logits, values resident in HBM) of the f16x2 evaluator in its two tower forms by batch size, one process, alternating:
where the Winograd form (K1w4, one launch for the tower) starts to pay.   python scripts/by_batch_forms.py [workload] [batches...]"""
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "chess20x256"
batches = [int(a) for a in sys.argv[2:]] or [32, 64, 96, 128, 160, 192, 256]
d, blob, planes = bench.make_workload(workload)
dev = torch.device("cuda", 0)
d_planes = torch.from_numpy(planes.view("int64")).to(dev)
stream = torch.cuda.Stream(device=dev)
out = {}
for n in batches:
    pol = torch.empty((n, d.moves), dtype=torch.float32, device=dev)
    val = torch.empty((n,), dtype=torch.float32, device=dev)
    evs = {}
    for form in ("direct", "winograd"):
        try:
            evs[form] = HipEvaluator(blob, batch_size=n, plane_words=planes.shape[2], dtype="f16x2", tower_form=form, switches={})
        except Exception as exc:  # noqa: BLE001 - a shape the form does not cover
            print(n, form, exc, file=sys.stderr)
    row = {}
    for rep in range(2):
        for form, ev in evs.items():
            for _ in range(100):
                ev.eval_device(d_planes.data_ptr(), n, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                ev.eval_device(d_planes.data_ptr(), n, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 300 * 1e3
            row.setdefault(form, {"kernel": ev.tower_kernel(), "ms_per_step": []})["ms_per_step"].append(round(ms, 4))
    for form, r in row.items():
        r["node_evals_per_s"] = round(n / min(r["ms_per_step"]) * 1e3)
    out[n] = row
    print(n, json.dumps(row), flush=True)
    for ev in evs.values():
        ev.close()
