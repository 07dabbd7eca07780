#!/usr/bin/env python3
"""Event-stamped tower launch duration of bench workloads: python scripts/time_tower.py [--batch N] [--warm REPS] WORKLOAD[:DTYPE]...
(DTYPE f16x2 by default)"""
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402

args = sys.argv[1:]
warm = 20
batch = None
if args and args[0] == "--batch":  # time the tower on the first BATCH leaves of the workload
    batch, args = int(args[1]), args[2:]
if args and args[0] == "--warm":
    warm, args = int(args[1]), args[2:]
for wl in args:
    d, blob, planes = bench.make_workload(wl.split(":")[0])
    n = batch or len(planes)
    dtype = wl.split(":")[1] if ":" in wl else "f16x2"
    ev = HipEvaluator(blob, batch_size=max(n, len(planes)), plane_words=planes.shape[2], dtype=dtype)
    ev.time_tower(n, warm)
    us, launches = ev.time_tower(n, 200 if dtype == "bf16" else 20)
    print(wl, "tower launch us:", round(us, 2), "x", launches, end="  |  ")
    ev.close()
print()
