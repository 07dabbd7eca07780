#!/usr/bin/env python3
"""Copy what scripts/collect_profiles.sh wrote under gpurun_out/TAG into profiles/ (round 4 names):
    python scripts/install_profiles.py gpurun_out/TAG
bench line, per-dtype rocprofv3 kernel stats and PMC summaries, the HBM traffic file bench.py reads, MFMA utilisation."""
import csv
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = Path(sys.argv[1])
prof = ROOT / "profiles"
line = (tag / "bench.json").read_text().strip().splitlines()[-1]
R = "r05"
for dt in ("f16x2", "bf16", "f16", "f32"):
    if (tag / f"bench_{dt}_kernel_stats.csv").exists():
        shutil.copy(tag / f"bench_{dt}_kernel_stats.csv", prof / f"{R}_bench_{dt}_kernel_stats.csv")
        shutil.copy(tag / f"pmc_summary_{dt}.json", prof / f"{R}_pmc_summary_{dt}.json")
        # the profiled command's OWN line beside its kernel statistics: the sum of the kernels of a step can be held against THAT run's
        # ms_per_step (profiled passes clock lower than the un-profiled bench: round-4 review)
        pl = (tag / f"stats_bench_{dt}.json").read_text().strip().splitlines()
        if pl:
            json.dump(json.loads(pl[-1]), open(prof / f"{R}_bench_{dt}_kernel_stats_line.json", "w"), indent=1)
for wl in ("hex7_6x64", "chess40x384"):
    if (tag / f"bench_{wl}.json").exists():
        json.dump(json.loads((tag / f"bench_{wl}.json").read_text().strip().splitlines()[-1]), open(prof / f"{R}_bench_{wl}.json", "w"), indent=1)
        for dt in ("f16x2", "bf16"):
            if (tag / f"bench_{wl}_{dt}_kernel_stats.csv").exists():
                shutil.copy(tag / f"bench_{wl}_{dt}_kernel_stats.csv", prof / f"{R}_bench_{wl}_{dt}_kernel_stats.csv")
                pl = (tag / f"stats_bench_{wl}_{dt}.json").read_text().strip().splitlines() if (tag / f"stats_bench_{wl}_{dt}.json").exists() else []
                if pl:
                    json.dump(json.loads(pl[-1]), open(prof / f"{R}_bench_{wl}_{dt}_kernel_stats_line.json", "w"), indent=1)
with open(prof / f"{R}_pmc_hbm_traffic.json", "w") as f:
    subprocess.check_call([sys.executable, str(ROOT / "scripts" / "make_traffic_json.py"), str(tag)], stdout=f)
# the bench line goes in LAST and is re-made from the collected one with the traffic figures of this very collection filled in
# (the collection's own bench run came before its PMC passes; nothing in the tracked summary is "withheld")
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

b = json.loads(line)
for obj, dt in ((b, b["dtype"]), (b.get("bf16"), "bf16"), (b.get("f16"), "f16"), (b.get("f32"), "f32")):
    if obj and "roofline" in obj:
        obj["roofline"]["traffic"], obj["roofline"]["traffic_source"] = bench.measured_traffic(obj["roofline"]["kernel"], dt)
if "roofline_plane_pack" in b:
    b["roofline_plane_pack"]["traffic"], b["roofline_plane_pack"]["traffic_source"] = bench.measured_traffic("planes_to_tensor_nchw64_kernel", "any")
json.dump(b, open(prof / f"{R}_bench.json", "w"), indent=1)
out = {"f16x2": {}, "bf16": {}}
out["source"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES (its own pass) and rocprofv3 --kernel-trace --stats of python3 bench.py --dtype D "
                 f"--steps 50 --warmup 5 (scripts/collect_profiles.sh {tag.name}), one MI355X box, round 5; tower_wino4_kernel: per layer (its launch holds 40)")
LAYERS = {"tower_wino4_kernel": 40}  # every layer behind the stem in one launch: per-layer figures
for dt, kern in (("f16x2", "tower_wino4_kernel"), ("bf16", "conv3x3_mfma_v2_kernel")):
    pm = json.load(open(prof / f"{R}_pmc_summary_{dt}.json"))
    # the 256 -> 256 layers: the STEM variants (third template flag) are left out
    busy = [(v["SQ_VALU_MFMA_BUSY_CYCLES"]["avg_per_launch"], v["SQ_VALU_MFMA_BUSY_CYCLES"]["launches"])
            for k, v in pm.items() if kern in k and "ELb1ELi" not in k and "SQ_VALU_MFMA_BUSY_CYCLES" in v]
    b = sum(a * n for a, n in busy) / sum(n for _, n in busy) / LAYERS.get(kern, 1)
    rows = [r for r in csv.DictReader(open(prof / f"{R}_bench_{dt}_kernel_stats.csv")) if kern in r["Name"] and "ELb1ELi" not in r["Name"]]
    avg = sum(float(r["TotalDurationNs"]) for r in rows) / sum(int(r["Calls"]) for r in rows) / 1e3 / LAYERS.get(kern, 1)
    out[dt].update({"mfma_busy_cycles_per_simd_and_launch": b / 1024, "avg_launch_us_rocprof": avg,
                    "utilisation_at_2.4GHz": b / 1024 / (avg * 2400), "utilisation_at_2.0GHz": b / 1024 / (avg * 2000)})
    print(dt, b / 1024, round(avg, 2), round(b / 1024 / (avg * 2400), 4))
json.dump(out, open(prof / f"{R}_pmc_mfma_utilisation.json", "w"), indent=1)
