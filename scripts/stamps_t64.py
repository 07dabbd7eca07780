#!/usr/bin/env python3
"""Diagnostic only: cycle stamps of tower64_lds_kernel on the stamped build (python -m cattus_amd.build --diag).
    python scripts/stamps_t64.py [WORKLOAD]     (default hex7_6x64)"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("CATTUS_HIP_LIB", os.path.join(ROOT, "cattus_amd", "libcattus_hip_diag.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from cattus_amd import evaluator as ev_mod  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "hex7_6x64"
d, blob, planes = bench.make_workload(wl)
ev = ev_mod.HipEvaluator(blob, batch_size=len(planes), plane_words=planes.shape[2], dtype="bf16")
for _ in range(30):
    ev.eval(planes)
L = ev_mod.load_library()
n = 256 * 16 * 8
buf = (C.c_ulonglong * n)()
assert L.cattus_hip_debug_stamps(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(256, 16, 8)
used = st[:, 0, 0] != 0
cons = st[used][:, :4]
layers = 1 + 2 * d.blocks
tot = cons[..., 3] - cons[..., 0]
print(f"{wl}: {used.sum()} workgroups, {layers} layers; consumer wave lifetime median {np.median(tot):.0f} cycles")
print(f"  start -> first barrier passed (plane expansion) {np.median(cons[...,1]-cons[...,0]):.0f}")
print(f"  layers {np.median(cons[...,2]-cons[...,1]):.0f} = {np.median(cons[...,2]-cons[...,1])/layers:.0f} per layer; of it at barriers "
      f"{np.median(cons[...,4])/layers:.0f}, in epilogues {np.median(cons[...,5])/layers:.0f} per layer")
print(f"  heads + store {np.median(cons[...,3]-cons[...,2]):.0f}")
