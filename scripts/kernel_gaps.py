#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV: python scripts/kernel_gaps.py TRACE.csv [name-substring]
Prints, per kernel name, the average duration and the average idle time between a kernel's end and the next kernel's start."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur, gap_after = defaultdict(list), defaultdict(list)
for a, b in zip(rows, rows[1:]):
    name = a["Kernel_Name"].split("(")[0][:70]
    dur[name].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    gap_after[name].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for name in dur:
    if len(dur[name]) < 20 or (len(sys.argv) > 2 and sys.argv[2] not in name):
        continue
    d, g = dur[name], sorted(gap_after[name])
    print(f"{name:72s} n={len(d):5d} dur {sum(d) / len(d) / 1e3:7.2f} us  gap to next: median {g[len(g) // 2] / 1e3:6.2f} us  mean {sum(g) / len(g) / 1e3:7.2f}")
