#!/usr/bin/env python3
"""Diagnostic only: run the chess 20x256 batch-256 forward on the stamped build
(cattus_amd/libcattus_hip_diag.so, -DCATTUS_STAMPS; build it with `python -m cattus_amd.build --diag`)
and print where the tower kernel's cycles go.
Never quote this build's run time; read the shares."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("CATTUS_HIP_LIB", os.path.join(ROOT, "cattus_amd", "libcattus_hip_diag.so"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cattus_amd import evaluator as ev_mod, synth  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
ev = ev_mod.HipEvaluator(seeded_blob(d, 2), batch_size=256, plane_words=1, dtype=dtype)
planes = synth.random_chess_planes(256, 2)
for _ in range(30):
    ev.eval(planes)  # warm clocks; the stamps of the LAST tower launch (a residual conv) remain
L = ev_mod.load_library()
n = 256 * 16 * 8
buf = (C.c_ulonglong * n)()
assert L.cattus_hip_debug_stamps(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(256, 16, 8)
nload = int((st[0, 4:, 0] != 0).sum())
cons, load = st[:, :4], st[:, 4 : 4 + nload]
print("loader waves per workgroup:", nload)
tot = cons[..., 3] - cons[..., 0]
rt = (cons[..., 6] - cons[..., 5]) / 100e6  # seconds (100 MHz)
clk = np.median(tot / np.maximum(rt, 1e-12)) / 1e9
print(f"dtype {dtype}: in-kernel clock ~{clk:.2f} GHz; consumer wave lifetime median {np.median(tot):.0f} cycles = {np.median(tot)/clk/1e3:.2f} us")
print(f"  consumers: prologue (start->first barrier passed) {np.median(cons[...,1]-cons[...,0]):.0f}, loop {np.median(cons[...,2]-cons[...,1]):.0f}, "
      f"epilogue {np.median(cons[...,3]-cons[...,2]):.0f}; of the loop, blocked at barriers {np.median(cons[...,4]):.0f}")
print(f"  loaders  : lifetime {np.median(load[...,3]-load[...,0]):.0f}, waiting for data (vmcnt) {np.median(load[...,4]):.0f}, "
      f"waiting at barriers {np.median(load[...,7]):.0f}; prologue {np.median(load[...,1]-load[...,0]):.0f}")
mfma = {"bf16": 12 * 48 * 32, "f32": 24 * 192 * 64, "f16x2": 24 * 72 * 32}[dtype]
print(f"  MFMA-bound loop time {mfma} cycles -> loop efficiency {mfma/np.median(cons[...,2]-cons[...,1]):.2f}, kernel efficiency {mfma/np.median(tot):.2f}")
span = (cons[..., 6].max() - cons[..., 5].min()) / 100.0  # us on the 100 MHz real-time counter, one time base for all XCDs
starts = (cons[..., 5] - cons[..., 5].min()) / 100.0
ends = (cons[..., 6].max() - cons[..., 6]) / 100.0
us, launches = ev.time_tower(256, 50)
print(f"  first wave start -> last wave end over all workgroups: {span:.2f} us; event-stamped launch {us:.2f} us "
      f"(the difference is dispatch before the first wave and completion after the last)")
print(f"  wave start after the first one: median {np.median(starts):.2f} us, max {starts.max():.2f} us; "
      f"wave end before the last one: median {np.median(ends):.2f} us, max {ends.max():.2f} us")
# who ends late: by XCD (blockIdx % 8 share one), and the spread of the consumer phases over workgroups
endt = (cons[..., 6].max(axis=1) - cons[..., 5].min()) / 100.0
print("  workgroup end (us after the first wave start) by blockIdx % 8:", " ".join(f"{endt[x::8].mean():.2f}" for x in range(8)),
      f"| min {endt.min():.2f} median {np.median(endt):.2f} max {endt.max():.2f}")
for name, a, b in (("prologue", 0, 1), ("loop", 1, 2), ("epilogue", 2, 3)):
    ph = (cons[..., b] - cons[..., a]).max(axis=1)
    print(f"  {name}: per-workgroup cycles min {ph.min():.0f} median {np.median(ph):.0f} p90 {np.percentile(ph, 90):.0f} max {ph.max():.0f}")
