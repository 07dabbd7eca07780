#!/usr/bin/env python3
"""A/B of the two f16x2 conv kernels (CATTUS_SPLIT_W=1: weights in a register ring, =0: through the LDS ring):
outputs must agree bit for bit (same MFMA sequence per accumulator); event-stamped tower launch durations of both.

    python scripts/splitw_ab.py [WORKLOAD[:BATCH] ...]      (bench.py workload names; default set below)
"""
import json
import os
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import bench  # noqa: E402
from cattus_amd import synth  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, hex_game, seeded_blob  # noqa: E402

cases = []
for a in sys.argv[1:] or ["chess20x256", "chess20x256:128", "chess20x256:64", "chess20x256:17"]:
    name, _, b = a.partition(":")
    d, blob, planes = bench.make_workload(name)
    cases.append((a, d, blob, planes[: int(b)] if b else planes))
# shapes the headline does not touch: 128-slot boards (hex 9 / 11), 96 -> 128 padded filters, ragged batches
for tag, d, planes in [
    ("hex11_4x96:37", NetDesc(**hex_game(11), blocks=4, filters=96, vhc=8, phc=8), synth.random_hex_planes(37, 11, 5)),
    ("hex9_3x64:130", NetDesc(**hex_game(9), blocks=3, filters=64, vhc=4, phc=4), synth.random_hex_planes(130, 9, 6)),
    ("chess_0x64:5", NetDesc(**CHESS, blocks=0, filters=64, vhc=8, phc=8), synth.random_chess_planes(5, 7)),
]:
    cases.append((tag, d, seeded_blob(d, 77), planes))

out = {}
for tag, d, blob, planes in cases:
    res = {}
    for mode in ("1", "0"):
        os.environ["CATTUS_SPLIT_W"] = mode  # read by cattus_hip_create
        with HipEvaluator(blob, batch_size=len(planes), plane_words=planes.shape[2], dtype="f16x2") as ev:
            p, v = ev.eval(planes)
            ev.time_tower(len(planes), 5)
            us, launches = ev.time_tower(len(planes), 20)
        res[mode] = (p, v, us, launches)
    same = bool((res["1"][0] == res["0"][0]).all() and (res["1"][1] == res["0"][1]).all())
    out[tag] = dict(identical=same, register_ring_us=round(res["1"][2], 2), lds_ring_us=round(res["0"][2], 2), launches=res["1"][3],
                    max_abs_dlogit=float(np.abs(res["1"][0] - res["0"][0]).max()))
    print(tag, out[tag], flush=True)
os.environ.pop("CATTUS_SPLIT_W", None)
print(json.dumps(out))
sys.exit(0 if all(o["identical"] for o in out.values()) else 1)
