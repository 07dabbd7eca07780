#!/bin/bash
# Evaluator-only bench line and the rocprofv3 kernel stats of the same command on ONE box (boxes differ by up to 12 %):
#   bash scripts/collect_stats_only.sh TAG
set -u
TAG=${1:-stats}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
F="--lanes 1 --settle-seconds 0 --no-bf16 --no-f16 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32"
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 > "$OUT/bench_evaluator_only.json" 2> "$OUT/bench.err" || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py --steps 50 --warmup 5 $F > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || exit 1
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/stats"
python3 - "$OUT" <<'PY'
import csv, json, sys
out = sys.argv[1]
d = json.loads(open(out + "/bench_evaluator_only.json").read().strip().splitlines()[-1])
tot = calls = 0
for row in csv.DictReader(open(out + "/kernel_stats.csv")):
    if "conv3x3_split" in row["Name"] or "conv3x3_mfma_v2_kernel" in row["Name"]:
        tot += int(row["TotalDurationNs"]); calls += int(row["Calls"])
print(f"value {d['value']:.0f} ms/step {d['ms_per_step']:.4f} launch_us(events) {d['roofline']['avg_launch_us']:.2f} frac {d['roofline']['frac']:.4f} "
      f"two-lane {d['two_batches_in_flight']['value']:.0f} | rocprof conv avg {tot / max(calls, 1) / 1e3:.2f} us over {calls} calls")
PY
