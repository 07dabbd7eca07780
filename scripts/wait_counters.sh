#!/bin/bash
# Where a tower kernel's wave cycles go, from the SQ counters (one rocprofv3 --pmc pass per counter group; gfx950 has 8 SQ slots):
#   scripts/wait_counters.sh OUTDIR WORKLOAD [DTYPE]      e.g.  scripts/wait_counters.sh gpurun_out/wc_hex7 hex7_6x64 f16x2
# SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked at s_waitcnt / s_barrier) + SQ_WAIT_INST_ANY (issue stalls: a dependent MFMA, a busy pipe)
# + SQ_ACTIVE_INST_ANY, roughly (MI355X_MICROARCH.md, rocprofv3 PMC slots); SQ_WAIT_INST_LDS is the LDS-issue share of the second;
# TCP_PENDING_STALL_CYCLES counts the vector L1 stalled on its pending-request queue (the CU's memory pipeline backed up).
out=$1; wl=$2; dt=${3:-f16x2}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
QUIET="--lanes 1 --settle-seconds 0 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 --no-bf16 --no-f16 --no-smi"
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCP_PENDING_STALL_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc_$tag" -o w -- python3 bench.py --workload $wl --dtype $dt --steps 5 --warmup 2 $QUIET > /dev/null 2> "$out/pmc_$tag.err" || echo "$tag failed: $(tail -2 $out/pmc_$tag.err)"
done
python3 scripts/summarise_pmc.py "$out" > "$out/wait_summary.json"
python3 - "$out/wait_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "tower" in k or "wino" in k or "splitw" in k:
        print(k[:60], json.dumps({c: round(x["avg_per_launch"]) for c, x in v.items() if isinstance(x, dict)}))
PY
find "$out" -name '*.csv' -size +200k -delete 2> /dev/null
