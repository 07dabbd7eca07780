#!/usr/bin/env python3
"""Diagnostic only: where the cycles of conv3x3_wino4_kernel go, on the stamped build (cattus_amd/libcattus_hip_diag.so,
-DCATTUS_STAMPS: `python -m cattus_amd.build --diag`).  Never quote this build's run time; read the shares.
    python scripts/stamps_w4.py [blocks filters batch]      # default 20 256 256"""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("CATTUS_HIP_LIB", os.path.join(ROOT, "cattus_amd", "libcattus_hip_diag.so"))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cattus_amd import evaluator as ev_mod, synth  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

blocks, filters, batch = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (20, 256, 256)
d = NetDesc(**CHESS, blocks=blocks, filters=filters, vhc=8, phc=8)
persist = os.environ.get("W4_PERSIST", "1")  # W4_PERSIST=0: the per-layer launches
ev = ev_mod.HipEvaluator(seeded_blob(d, 2), batch_size=batch, plane_words=1, dtype="f16x2", switches={"CATTUS_WINO_KERNEL": "k4", "CATTUS_WINO_PERSIST": persist})
print("tower kernel:", ev.tower_kernel())
planes = synth.random_chess_planes(batch, 2)
for _ in range(40):
    ev.eval(planes)  # warm clocks; the stamps of the LAST tower launch (a layer with skip rows) remain
L = ev_mod.load_library()
wgs = min(1024, (batch // 4) * (filters // 64))
n = 1024 * 4 * 10
buf = (C.c_ulonglong * n)()
assert L.cattus_hip_debug_stamps_w4(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(1024, 4, 10)[:wgs]
tot = st[..., 5] - st[..., 0]
rt = (st[..., 7] - st[..., 6]) / 100e6
clk = np.median(tot / np.maximum(rt, 1e-12)) / 1e9
names = ["prologue (start -> loop)", "loop", "ring drain", "Z = M A + exchange writes (2 barriers)", "Y, bias, skip, stores"]
print(f"chess {blocks}x{filters} batch {batch}: {wgs} workgroups; in-kernel clock ~{clk:.2f} GHz; wave lifetime median {np.median(tot):.0f} cycles = {np.median(tot) / clk / 1e3:.2f} us")
for i, nm in enumerate(names):
    seg = st[..., i + 1] - st[..., i]
    print(f"  {nm:42s} median {np.median(seg):8.0f}  p10 {np.percentile(seg, 10):8.0f}  p90 {np.percentile(seg, 90):8.0f}  = {np.median(seg) / clk / 1e3:6.2f} us")
if st[..., 8].any():
    seg = st[..., 8] - st[..., 0]
    print(f"  of the prologue: start -> behind the hand-off wait   median {np.median(seg):8.0f}  p10 {np.percentile(seg, 10):8.0f}  p90 {np.percentile(seg, 90):8.0f}  = {np.median(seg) / clk / 1e3:6.2f} us")
if st[..., 9].any():
    seg = st[..., 9] - st[..., 4]
    print(f"  of the last segment: stores issued and drained        median {np.median(seg):8.0f}  p10 {np.percentile(seg, 10):8.0f}  p90 {np.percentile(seg, 90):8.0f}  = {np.median(seg) / clk / 1e3:6.2f} us")
nks = filters // 16
print(f"  MFMA floor of the loop: {nks * 48 * 32} cycles ({nks} k-steps x 48 MFMAs x 32); loop / floor = {np.median(st[..., 2] - st[..., 1]) / (nks * 48 * 32):.2f}")
span = (st[..., 7].max() - st[..., 6].min()) / 100.0
us, launches = ev.time_tower(batch, 30)
print(f"  first wave start -> last wave end: {span:.2f} us; event-stamped launch {us:.2f} us")
starts = (st[..., 6] - st[..., 6].min()) / 100.0
print(f"  wave start after the first: median {np.median(starts):.2f} us, max {starts.max():.2f} us")
