#!/usr/bin/env python3
"""Resident split tower against the per-layer launches, for every ring depth / heads switch and a few batch sizes (GPU box)."""
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401

from cattus_amd.evaluator import HipEvaluator
from cattus_amd.weights import NetDesc, hex_game, seeded_blob

d = NetDesc(**hex_game(7), blocks=3, filters=64, vhc=16, phc=16)
blob = seeded_blob(d, 17)
rng = np.random.default_rng(5)
for n in (128, 300, 600, 1100):
    planes = np.zeros((n, d.planes, 2), dtype=np.uint64)
    bits = rng.integers(0, 2, size=(n, d.planes, 49), dtype=np.uint64)
    for i in range(49):
        planes[:, :, i >> 6] |= bits[:, :, i] << np.uint64(i & 63)
    os.environ["CATTUS_TOWER64"] = "0"
    with HipEvaluator(blob, batch_size=n, plane_words=2, dtype="f16x2") as ev:
        want_p, want_v = ev.eval(planes)
    del os.environ["CATTUS_TOWER64"]
    for depth in ("1", "2", "9"):
        for heads in ("1", "0"):
            os.environ["CATTUS_T64S_SHAPE"], os.environ["CATTUS_T64S_HEADS"] = depth, heads
            with HipEvaluator(blob, batch_size=n, plane_words=2, dtype="f16x2") as ev:
                bad = []
                for rep in range(3):
                    p, v = ev.eval(planes)
                    wrong = np.nonzero((p != want_p).any(axis=1) | (v != want_v))[0]
                    bad.append(len(wrong))
                print(f"n={n} shape={depth} fused_heads={heads}: wrong leaves per pass {bad}", "first:", wrong[:12].tolist() if len(wrong) else "", flush=True)
