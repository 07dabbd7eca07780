#!/bin/bash
# Collect the round's measured evidence on a GPU box (run from the repository root):
#   bash scripts/collect_profiles.sh [TAG]
# Writes gpurun_out/<TAG>/: the default bench line, rocprofv3 kernel stats of the same command per dtype, and the
# separate PMC passes (HBM-side traffic, MFMA busy) per dtype.  scripts/summarise_pmc.py turns the PMC CSVs into JSON,
# scripts/make_traffic_json.py into the file bench.py reads for roofline.traffic.
set -u
TAG=${1:-final}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
QUIET="--lanes 1 --settle-seconds 0 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 --no-bf16 --no-f16 --no-smi"
# PARTS (default "bench dtypes workloads") selects what runs: one gpurun call holds 20 minutes, the whole collection takes more
PARTS=${PARTS:-bench dtypes workloads}
if [[ " $PARTS " == *" bench "* ]]; then
echo "== bench (defaults)"; timeout -k 10 900 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
tail -c 400 "$OUT/bench.json"; echo
fi
if [[ " $PARTS " == *" dtypes "* ]]; then
for dt in ${DTYPES:-f16x2 bf16 f16 f32}; do
  echo "== rocprofv3 kernel stats, $dt"
  STEPS=50; [ $dt = f32 ] && STEPS=10
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$dt" -o bench -- python3 bench.py --dtype $dt --steps $STEPS --warmup 5 $QUIET > "$OUT/stats_bench_$dt.json" 2> "$OUT/stats_$dt.err" || exit 1
  find "$OUT/stats_$dt" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_${dt}_kernel_stats.csv" \;
  rm -rf "$OUT/stats_$dt"
  PMCS="FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; [ $dt = f32 -o $dt = f16 ] && PMCS="FETCH_SIZE WRITE_SIZE"
  for pmc in $PMCS; do
    echo "== rocprofv3 --pmc $pmc, $dt"
    timeout -k 10 600 rocprofv3 --pmc $pmc --output-format csv -d "$OUT/$dt/pmc_$pmc" -o bench -- python3 bench.py --dtype $dt --steps 5 --warmup 2 $QUIET > /dev/null 2> "$OUT/pmc_${pmc}_$dt.err" || exit 1
  done
  python3 scripts/summarise_pmc.py "$OUT/$dt" > "$OUT/pmc_summary_$dt.json" && head -c 600 "$OUT/pmc_summary_$dt.json"
  rm -rf "$OUT/$dt"
done
python3 scripts/make_traffic_json.py "$OUT" > "$OUT/pmc_hbm_traffic.json"
fi
# BASELINE configs 2 and 5 as the driver would run them (evaluator-only line + kernel stats), every tower
if [[ " $PARTS " == *" workloads "* ]]; then
for wl in hex7_6x64 chess40x384; do
  echo "== bench --workload $wl"
  timeout -k 10 600 python3 bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-f32 > "$OUT/bench_$wl.json" 2> "$OUT/bench_$wl.err" || exit 1
  for dt in f16x2 bf16; do
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_${wl}_$dt" -o bench -- python3 bench.py --workload $wl --dtype $dt --steps 50 --warmup 5 $QUIET > "$OUT/stats_bench_${wl}_$dt.json" 2> "$OUT/stats_${wl}_$dt.err" || exit 1
    find "$OUT/stats_${wl}_$dt" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_${wl}_${dt}_kernel_stats.csv" \;
    rm -rf "$OUT/stats_${wl}_$dt"
  done
done
fi
du -sh "$OUT"
