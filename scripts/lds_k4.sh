#!/bin/bash
out=$1; mkdir -p $out
Q="--lanes 1 --settle-seconds 0 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 --no-bf16 --no-f16 --no-smi --steps 5 --warmup 2"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for pmc in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS; do
  timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/k4/pmc_$pmc" -o b -- python3 bench.py $Q --dtype f16x2 > /dev/null 2> "$out/k4_$pmc.err" || echo "$pmc failed"
done
python3 scripts/summarise_pmc.py "$out/k4" > "$out/k4.json"
find "$out" -name '*.csv' -size +200k -delete 2> /dev/null
python3 - "$out/k4.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    c = {n: x["avg_per_launch"] for n, x in v.items() if isinstance(x, dict)}
    if c.get("SQ_LDS_IDX_ACTIVE"):
        print(k[:50], {n: round(x) for n, x in c.items()}, "conflict/active", round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3))
PY
