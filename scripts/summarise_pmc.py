#!/usr/bin/env python3
"""Average per-launch PMC values per kernel from the rocprofv3 counter CSVs written by collect_profiles.sh."""
import csv
import glob
import json
import sys
from collections import defaultdict

out_dir = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(f"{out_dir}/pmc_*/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
            a = acc[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
res = {}
for k, counters in sorted(acc.items()):
    if not any(s in k for s in ("conv3x3", "planes_to_tensor", "head_gemm", "head_conv", "pack_planes", "tower64", "tower_wino4", "head_fc")):
        continue
    e = {c: {"avg_per_launch": v[0] / v[1], "launches": v[1]} for c, v in counters.items()}
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # both counters are in KiB; on gfx950 FETCH_SIZE counts half of a wide read (MI355X_MICROARCH.md)
        e["traffic_bytes_per_launch"] = int(2 * e["FETCH_SIZE"]["avg_per_launch"] * 1024 + e["WRITE_SIZE"]["avg_per_launch"] * 1024)
    res[k] = e
json.dump(res, sys.stdout, indent=1)
