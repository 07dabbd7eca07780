#!/bin/bash
# rocprofv3 kernel stats of short evaluator-only bench runs (hex7 6x64 b128 and chess 20x256 b256): the head kernels' durations
#   bash scripts/head_stats.sh TAG
set -u
TAG=${1:-heads}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
F="--lanes 1 --settle-seconds 0 --no-bf16 --no-f16 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32"
for wl in hex7_6x64 chess20x256; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -o bench -- python3 bench.py --workload $wl --steps 50 --warmup 5 $F > "$OUT/bench_$wl.json" 2> "$OUT/stats_$wl.err" || exit 1
  find "$OUT/stats_$wl" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$wl.csv" \;
  rm -rf "$OUT/stats_$wl"
  echo "== $wl"; awk -F'","' 'NR>1 && NR<9 {printf "%-60.60s calls %s avg_ns %s\n", $1, $2, $4}' "$OUT/kernel_stats_$wl.csv"
  python3 -c "import json,sys; d=json.loads(open('$OUT/bench_$wl.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'], 'value', d['value'])"
done
