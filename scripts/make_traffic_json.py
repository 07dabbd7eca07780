#!/usr/bin/env python3
"""profiles/rNN_pmc_hbm_traffic.json from the per-dtype PMC summaries of scripts/collect_profiles.sh:
    python scripts/make_traffic_json.py gpurun_out/TAG > profiles/r05_pmc_hbm_traffic.json
The file carries the sha256 of the code of cattus_amd/csrc/kernels.hip (comments and white space removed) the passes ran on; bench.py withholds the traffic figure
when the kernels have changed since (it cannot collect PMC counters inside its own process)."""
import hashlib
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
tag = Path(sys.argv[1])
# algorithmic HBM-side bytes per launch, chess 20x256 at batch 256, averaged over the 41 launches of a step: per launch the
# 16384 x 256 activations in and out (+ the skip rows in 20 of them) at the dtype's bytes per channel, the layer's weights once
ACT = 16384 * 256
ALG = {}
for dtype, abytes, wbytes in (("bf16", 2, 2), ("f16", 2, 2), ("f16x2", 4, 4), ("f32", 4, 4)):
    w = 9 * 256 * 256 * wbytes
    stem = 256 * 144 + 9 * 256 * (64 if abytes == 2 else 32) * wbytes + ACT * abytes
    ALG[dtype] = (stem + 20 * (2 * ACT * abytes + w) + 20 * (3 * ACT * abytes + w)) / 41
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/collect_profiles.sh) -- python3 bench.py --dtype D --steps 5 "
              "--warmup 2 --lanes 1 --settle-seconds 0 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 --no-bf16 --no-f16, MI355X",
    "units": "FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced read, so fetched bytes = "
             "2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section). Fabric-side L2 requests: Infinity Cache hits are included.",
    "kernels_sha256": bench.kernels_sha256(),  # of the code: comments and white space removed
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
    "by_dtype": {},
}
def base_name(k: str) -> str:
    """Kernel name without namespace and template arguments, from a demangled or an Itanium-mangled name."""
    import re

    m = re.match(r"_ZN6cattus(\d+)", k)
    if m:
        n = int(m.group(1))
        return k[m.end() : m.end() + n]
    return k.replace("void ", "").replace("cattus::", "").split("<")[0].split("(")[0].strip()


LAYERS_PER_LAUNCH = {"tower_wino4_kernel": 40}  # chess 20x256: 2 x 20 layers behind the stem
TOWER = {"f16x2": "tower_wino4_kernel", "bf16": "conv3x3_mfma_v2_kernel", "f16": "conv3x3_mfma_v2_kernel", "f32": "conv3x3_mfma_v2_kernel"}
# the f16x2 tower at batch 256 runs its 40 layers behind the stem in Winograd form: f32 rows in and out (+ the skip rows in 20 of them),
# the layer's transformed weights U (16 frequencies x 256 x 256 pairs = 4.19 MB) once.  tower_wino4_kernel runs all 40 in ONE launch:
# its counters are divided by LAYERS_PER_LAUNCH so that the figure is per layer, like bench.py's roofline.achieved
ALG["f16x2"] = (20 * (2 * ACT * 4 + 16 * 256 * 256 * 4) + 20 * (3 * ACT * 4 + 16 * 256 * 256 * 4)) / 40
for dtype in ("f16x2", "bf16", "f16", "f32"):
    f = tag / f"pmc_summary_{dtype}.json"
    if not f.exists():
        continue
    summ = json.loads(f.read_text())
    acc = {}  # base name -> [sum of traffic x launches, launches, variants]
    for k, v in summ.items():
        if "traffic_bytes_per_launch" not in v:
            continue
        n = v["FETCH_SIZE"]["launches"]
        a = acc.setdefault(base_name(k), [0.0, 0, {}])
        a[0] += v["traffic_bytes_per_launch"] * n
        a[1] += n
        a[2][k[:110]] = {"traffic_bytes_per_launch": v["traffic_bytes_per_launch"], "launches": n}
    entry = {}
    for name, (tot, n, variants) in acc.items():
        per = LAYERS_PER_LAUNCH.get(name, 1)
        entry[name] = {"traffic_bytes_per_launch": int(tot / n / per), "launches": n}
        if per > 1:
            entry[name]["layers_per_launch"] = per
            entry[name]["traffic_bytes_per_whole_launch"] = int(tot / n)
        if len(variants) > 1:
            entry[name]["by_variant"] = variants
        if name == TOWER[dtype]:
            entry[name]["algorithmic_bytes_per_launch"] = int(ALG[dtype])
            entry[name]["note"] = ("average over the launches of a step (20 convs without and 20 with the skip rows; bf16 / f16 / f32: the stem too); the excess "
                                   "over the algorithmic bytes is the layer's weight set fetched once per XCD L2 (8 of them) instead of once")
    out["by_dtype"][dtype] = entry
    if "planes_to_tensor_nchw64_kernel" in entry:
        out["by_dtype"].setdefault("any", {})["planes_to_tensor_nchw64_kernel"] = dict(entry["planes_to_tensor_nchw64_kernel"], algorithmic_bytes_per_launch=262144 * 4752)
json.dump(out, sys.stdout, indent=1)
