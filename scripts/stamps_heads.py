#!/usr/bin/env python3
"""Diagnostic only: cycle stamps of head_fc_pair_kernel on the stamped build (python -m cattus_amd.build --diag).
    python scripts/stamps_heads.py [WORKLOAD]     (default hex7_6x64)"""
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
n = 256 * 4 * 8
buf = (C.c_ulonglong * n)()
assert L.cattus_hip_debug_head_stamps(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(256, 4, 8)
used = st[:, 0, 0] != 0
t0 = st[used][:, :, 0].min()
print(f"{wl}: {used.sum()} blocks stamped; cycles relative to the earliest wave start (memtime, 100 MHz? see ratio below)")
names = ["start", "chunk 1: top", "staged", "barrier + next fetch issued", "MFMAs done", "k loop done", "end"]
for b in np.nonzero(used)[0][:12]:
    w = st[b, 0]
    print(f"  block {b:3d} wave0:", " ".join(f"{nm}={w[i] - t0}" for i, nm in enumerate(names) if w[i]))
last = st[used][:, :, 6].max()
print(f"  all waves: first start -> last end {last - t0}")
