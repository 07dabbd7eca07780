#!/usr/bin/env python3
"""Fixed and per-layer cost of the resident towers: launch time of hex7 N x 64 networks for several N (GPU box).
    python scripts/t64s_slope.py [batch=128] [dtype=f16x2]"""
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401

from cattus_amd import synth
from cattus_amd.evaluator import HipEvaluator
from cattus_amd.weights import NetDesc, hex_game, seeded_blob

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
planes = synth.random_hex_planes(batch, 7, 1)
pts = []
for blocks in (0, 2, 6, 12, 20):
    d = NetDesc(**hex_game(7), blocks=blocks, filters=64, vhc=16, phc=16)
    with HipEvaluator(seeded_blob(d, 1), batch_size=batch, plane_words=2, dtype=dtype) as ev:
        ev.time_tower(batch, 20)
        us, launches = ev.time_tower(batch, 100)
    pts.append((1 + 2 * blocks, us * launches))
    print(f"{dtype} batch {batch}: {1 + 2 * blocks:3d} layers: {us * launches:7.2f} us ({launches} launch)", flush=True)
x, y = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts], float)
slope, icpt = np.polyfit(x, y, 1)
print(f"fit: {icpt:.2f} us fixed + {slope:.3f} us per layer")
