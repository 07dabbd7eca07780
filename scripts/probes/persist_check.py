#!/usr/bin/env python3
"""One-launch Winograd tower against the per-layer launches on a ladder of shapes: which leaves differ, by how much."""
import sys

sys.path.insert(0, ".")
import numpy as np

from cattus_amd import synth
from cattus_amd.evaluator import HipEvaluator
from cattus_amd.weights import CHESS, NetDesc, seeded_blob

for blocks, filters, n in ((1, 128, 4), (1, 128, 8), (1, 128, 256), (2, 128, 4), (2, 128, 256), (1, 256, 256), (3, 256, 256), (20, 256, 256)):
    d = NetDesc(**CHESS, blocks=blocks, filters=filters, vhc=8, phc=8)
    blob = seeded_blob(d, 31)
    planes = synth.random_chess_planes(n, 17)
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2", switches={"CATTUS_WINO_KERNEL": "k4", "CATTUS_WINO_PERSIST": "0"}) as ev:
        want = ev.eval(planes)
    with HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2", switches={"CATTUS_WINO_KERNEL": "k4"}) as ev:
        k = ev.tower_kernel()
        got = ev.eval(planes)
        got2 = ev.eval(planes)
    bad = np.where((got[0] != want[0]).any(axis=1) | (got[1] != want[1]))[0]
    bad2 = np.where((got2[0] != want[0]).any(axis=1) | (got2[1] != want[1]))[0]
    print(f"{blocks}x{filters} n={n} {k}: {len(bad)} leaves differ (second pass {len(bad2)}), max |dp| {np.abs(got[0] - want[0]).max():.3g}, first bad {bad[:8].tolist()}", flush=True)
