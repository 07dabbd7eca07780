#!/bin/bash
# Counters of the Winograd layer (one rocprofv3 --pmc pass each): scripts/lds_counters.sh OUTDIR [LIB] ["COUNTER ..."]
out=$1; lib=${2:-cattus_amd/libcattus_hip.so}
mkdir -p $out
export CATTUS_HIP_LIB=$lib
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
counters=${3:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM}
for pmc in $counters; do
  timeout -k 10 300 rocprofv3 --pmc $pmc --output-format csv -d "$out/pmc_$pmc" -o w -- python3 scripts/wino_bits.py 0 > /dev/null 2> "$out/pmc_$pmc.err" || echo "$pmc failed"
done
python3 scripts/summarise_pmc.py "$out" > "$out/lds_summary.json"
python3 - "$out/lds_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "wino" in k:
        print(k[:40], {c: round(x["avg_per_launch"]) for c, x in v.items() if isinstance(x, dict)})
PY
