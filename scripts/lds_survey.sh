#!/bin/bash
# LDS bank-conflict share of every tower kernel: scripts/lds_survey.sh OUTDIR
out=$1; mkdir -p $out
Q="--lanes 1 --settle-seconds 0 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 --no-bf16 --no-f16 --no-smi --steps 5 --warmup 2"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() {  # name, extra bench args
  name=$1; shift
  for pmc in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES; do
    timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/$name/pmc_$pmc" -o b -- python3 bench.py $Q "$@" > /dev/null 2> "$out/${name}_$pmc.err" || echo "$name $pmc failed"
  done
  python3 scripts/summarise_pmc.py "$out/$name" > "$out/$name.json"
  python3 - "$out/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    c = {n: x["avg_per_launch"] for n, x in v.items() if isinstance(x, dict)}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print(f"{sys.argv[2]:18s} {k[:60]:60s} conflict/active = {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f}  active/busy = {c['SQ_LDS_IDX_ACTIVE'] / c.get('SQ_BUSY_CYCLES', 1):.2f}")
PY
}
run bf16 --dtype bf16
run f16 --dtype f16
run f32 --dtype f32
CATTUS_WINOGRAD=0 run f16x2_direct --dtype f16x2
run f16x2_wino --dtype f16x2
run hex7_f16x2 --dtype f16x2 --workload hex7_6x64
run chess40_bf16 --dtype bf16 --workload chess40x384
