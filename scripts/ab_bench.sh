#!/bin/bash
# A/B of two library builds over the bench's towers: scripts/ab_bench.sh OUTDIR ROUNDS VARIANT... ("" = the regular build)
out=$1; rounds=$2; shift 2
mkdir -p $out
Q="--lanes 1 --settle-seconds 0.2 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 --no-bf16 --no-f16 --no-smi --steps 100 --warmup 10"
one() {  # label, lib variant, env assignment, bench args...
  label=$1; v=$2; envs=$3; shift 3
  lib=cattus_amd/libcattus_hip${v:+_$v}.so
  env $envs CATTUS_HIP_LIB=$lib timeout -k 10 300 python bench.py $Q "$@" > $out/abb.json 2>$out/abb.err || { echo "$label ${v:-base} FAILED"; tail -3 $out/abb.err; return 1; }
  python - $out/abb.json "$label" "${v:-base}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:16s} {sys.argv[3]:6s} {d['value']:12.0f} {d['unit']}  {r['kernel']:28s} {r['avg_launch_us']:7.2f} us", flush=True)
PY
}
for r in $(seq $rounds); do
  for v in "$@"; do
    one bf16 "$v" X=1 --dtype bf16 || exit 1
    one f16 "$v" X=1 --dtype f16 || exit 1
    one f32 "$v" X=1 --dtype f32 --steps 20 || exit 1
    one f16x2_direct "$v" CATTUS_WINOGRAD=0 --dtype f16x2 || exit 1
    one hex7_f16x2 "$v" X=1 --dtype f16x2 --workload hex7_6x64 || exit 1
    one hex7_bf16 "$v" X=1 --dtype bf16 --workload hex7_6x64 || exit 1
  done
done
