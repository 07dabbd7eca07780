#!/usr/bin/env python3
"""sha256 of the f16x2 Winograd tower's outputs on the bench's 256 chess leaves + the tower launch time: run once per build
(CATTUS_HIP_LIB selects the library) to show that a change of the kernel leaves every output bit where it was.

    python scripts/wino_bits.py ; CATTUS_HIP_LIB=cattus_amd/libcattus_hip_prev.so python scripts/wino_bits.py
"""
import hashlib
import json
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("CATTUS_WINOGRAD", "1")
sys.path.insert(0, ".")
import torch  # noqa: F401,E402

from cattus_amd import synth  # noqa: E402
from cattus_amd.evaluator import LIB_PATH, HipEvaluator  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

out = {"lib": str(LIB_PATH)}
CASES = ((20, 256, 256, 2), (4, 128, 192, 5), (40, 384, 512, 3), (10, 256, 256, 2), (5, 256, 256, 2), (15, 256, 256, 2), (17, 256, 256, 2), (24, 256, 256, 2), (30, 256, 256, 2))
if len(sys.argv) > 1:  # e.g. "0" = the bench net only
    CASES = tuple(CASES[int(a)] for a in sys.argv[1:])
for blocks, filters, n, seed in CASES:
    d = NetDesc(**CHESS, blocks=blocks, filters=filters, vhc=8, phc=8)
    planes = synth.random_chess_planes(n, seed)
    with HipEvaluator(seeded_blob(d, seed), batch_size=n, plane_words=1, dtype="f16x2") as ev:
        p, v = ev.eval(planes)
        for _ in range(3):
            ev.time_tower(n, 20)
        us = [ev.time_tower(n, 20)[0] for _ in range(5)]
        out[f"chess{blocks}x{filters}_b{n}"] = dict(kernel=ev.tower_kernel(), sha256=hashlib.sha256(p.tobytes() + v.tobytes()).hexdigest()[:16],
                                                   launch_us=[round(u, 2) for u in us], saturated=ev.stats()["saturated"])
print(json.dumps(out))
