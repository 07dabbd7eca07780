#!/bin/bash
# Collect the round's measured evidence on a GPU box (run from the repository root):
#   bash scripts/collect_profiles.sh [TAG]
# Writes gpurun_out/<TAG>/: the default bench line, rocprofv3 kernel stats of the same command, and the
# separate PMC passes (HBM-side traffic, MFMA busy).  scripts/summarise_pmc.py turns the PMC CSVs into JSON.
set -u
TAG=${1:-final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== bench (defaults)"; timeout -k 10 600 python3 bench.py > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err" || exit 1
tail -c 600 "$OUT/bench_bf16.json"; echo
echo "== bench, two batches in flight"; timeout -k 10 600 python3 bench.py --lanes 2 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 > "$OUT/bench_bf16_two_lanes.json" 2> "$OUT/bench_two_lanes.err" || exit 1
echo "== bench f32"; timeout -k 10 600 python3 bench.py --dtype f32 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --steps 50 --warmup 5 > "$OUT/bench_f32.json" 2> "$OUT/bench_f32.err" || exit 1
echo "== rocprofv3 kernel stats"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py --steps 50 --warmup 5 --lanes 1 --no-long-run --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || exit 1
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_bf16_kernel_stats.csv" \;
for pmc in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  echo "== rocprofv3 --pmc $pmc"
  timeout -k 10 600 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/pmc_$pmc" -o bench -- python3 bench.py --steps 5 --warmup 2 --lanes 1 --no-long-run --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 > /dev/null 2> "$OUT/pmc_$pmc.err" || exit 1
done
python3 scripts/summarise_pmc.py "$OUT" > "$OUT/pmc_summary.json" && head -c 1500 "$OUT/pmc_summary.json"
echo "== config 2 (hex7 6x64, batch 128): bench line and kernel stats of the resident tower"
timeout -k 10 600 python3 bench.py --workload hex7_6x64 --no-cpu-baseline > "$OUT/bench_hex7_6x64.json" 2> "$OUT/bench_hex7.err" || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_hex7" -o bench -- python3 bench.py --workload hex7_6x64 --steps 50 --warmup 5 --lanes 1 --no-long-run --no-cpu-baseline > /dev/null 2> "$OUT/stats_hex7.err" || exit 1
find "$OUT/stats_hex7" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_hex7_6x64_kernel_stats.csv" \;
rm -rf "$OUT"/stats_hex7/*/*.db 2>/dev/null
echo "== config 5 shape on one GPU (chess 40x384, batch 512)"
timeout -k 10 600 python3 bench.py --workload chess40x384 --no-cpu-baseline --steps 50 > "$OUT/bench_chess40x384.json" 2> "$OUT/bench_40x384.err" || exit 1
rm -rf "$OUT"/stats/*/*.db "$OUT"/stats/*kernel_trace.csv "$OUT"/stats_hex7/*kernel_trace.csv 2>/dev/null
du -sh "$OUT"
