#!/usr/bin/env python3
"""One process, one library build (CATTUS_HIP_LIB): ms per batch forward of the default f16x2 evaluator as bench.py times it (planes,
logits, values resident in HBM), a sha256 of its outputs, the tower kernel's name.  For alternating A/B runs of two BUILDS on one box:
    scripts/ab_step_libs.sh ROUNDS LIB [LIB ...]
    python scripts/step_time.py [workload] [steps]"""
import hashlib
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
d, blob, planes = bench.make_workload(workload)
batch = len(planes)
dev = torch.device("cuda", 0)
d_planes = torch.from_numpy(planes.view("int64")).to(dev)
stream = torch.cuda.Stream(device=dev)
ev = HipEvaluator(blob, batch_size=batch, plane_words=planes.shape[2], dtype="f16x2", switches={})
pol = torch.empty((batch, d.moves), dtype=torch.float32, device=dev)
val = torch.empty((batch,), dtype=torch.float32, device=dev)
times = []
for rep in range(3):
    for _ in range(150):
        ev.eval_device(d_planes.data_ptr(), batch, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ev.eval_device(d_planes.data_ptr(), batch, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    times.append(round((time.perf_counter() - t0) / steps * 1e3, 4))
h = hashlib.sha256(pol.cpu().numpy().tobytes() + val.cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps({"lib": os.environ.get("CATTUS_HIP_LIB", "default"), "kernel": ev.tower_kernel(), "ms_per_step": times,
                  "node_evals_per_s": round(batch / min(times) * 1e3), "sha256": h}))
