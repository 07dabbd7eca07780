#!/usr/bin/env python3
"""Diagnostic: prologue / loop / epilogue cycles of conv3x3_wino8_kernel on the stamped build (libcattus_hip_diag.so)."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
os.environ.setdefault("CATTUS_HIP_LIB", os.path.join(ROOT, "cattus_amd", "libcattus_hip_diag.so"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cattus_amd import evaluator as ev_mod, synth  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
ev = ev_mod.HipEvaluator(seeded_blob(d, 2), batch_size=256, plane_words=1, dtype="f16x2", switches={"CATTUS_WINO_KERNEL": "k8"})
planes = synth.random_chess_planes(256, 2)
for _ in range(40):
    ev.eval(planes)
L = ev_mod.load_library()
n = 1024 * 8 * 4
buf = (C.c_ulonglong * n)()
assert L.cattus_hip_debug_stamps_w8(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(1024, 8, 4)[:256]
for half, name in ((slice(0, 4), "lh=0"), (slice(4, 8), "lh=1")):
    s = st[:, half]
    print(name, "prologue", int(np.median(s[..., 1] - s[..., 0])), "loop", int(np.median(s[..., 2] - s[..., 1])), "epilogue", int(np.median(s[..., 3] - s[..., 2])),
          "total", int(np.median(s[..., 3] - s[..., 0])))
print("launch us", ev.time_tower(256, 20))
