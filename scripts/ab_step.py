#!/usr/bin/env python3
"""Whole-step A/B on ONE box, in ONE process: ms per batch-256 forward (planes, logits, values resident in HBM, as bench.py times it) of
the f16x2 evaluator under each set of diagnostic switches, alternating, several rounds.

    python scripts/ab_step.py [rounds] [workload]      # default 3 rounds of chess20x256"""
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
workload = sys.argv[2] if len(sys.argv) > 2 else "chess20x256"
d, blob, planes = bench.make_workload(workload)
batch = len(planes)
dev = torch.device("cuda", 0)
d_planes = torch.from_numpy(planes.view("int64")).to(dev)
stream = torch.cuda.Stream(device=dev)
CONFIGS = {
    "k16 (conv3x3_wino_kernel, per layer)": {"CATTUS_WINO_KERNEL": "k16"},
    "k4 per layer (conv3x3_wino4_kernel)": {"CATTUS_WINO_KERNEL": "k4", "CATTUS_WINO_PERSIST": "0"},
    "k4 one launch (tower_wino4_kernel)": {"CATTUS_WINO_KERNEL": "k4"},
}
evs = {}
for name, sw in CONFIGS.items():
    try:
        evs[name] = HipEvaluator(blob, batch_size=batch, plane_words=planes.shape[2], dtype="f16x2", switches=sw)
    except Exception as exc:  # noqa: BLE001 - a shape one of the kernels does not cover
        print(f"{name}: {exc}", file=sys.stderr)
pol = torch.empty((batch, d.moves), dtype=torch.float32, device=dev)
val = torch.empty((batch,), dtype=torch.float32, device=dev)
out = {name: [] for name in evs}
ref = None
for r in range(rounds):
    for name, ev in evs.items():
        for _ in range(100):
            ev.eval_device(d_planes.data_ptr(), batch, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            ev.eval_device(d_planes.data_ptr(), batch, pol.data_ptr(), val.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        out[name].append(round((time.perf_counter() - t0) / 300 * 1e3, 4))
        bits = (pol.cpu().numpy().tobytes(), val.cpu().numpy().tobytes())
        assert ref is None or bits == ref, f"{name}: outputs differ from the first configuration's"
        ref = ref or bits
print(json.dumps({"workload": workload, "batch": batch, "ms_per_step": out,
                  "node_evals_per_s": {k: round(batch / (min(v) * 1e-3)) for k, v in out.items()},
                  "kernels": {k: ev.tower_kernel() for k, ev in evs.items()}, "outputs": "bit-identical across configurations"}))
