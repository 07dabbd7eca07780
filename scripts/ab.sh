#!/bin/bash
# A/B of library builds on one box: scripts/ab.sh OUTDIR ROUNDS VARIANT... ("" = the regular build); prints launch_us per run
out=$1; rounds=$2; shift 2
mkdir -p $out
for r in $(seq $rounds); do
  for v in "$@"; do
    lib=cattus_amd/libcattus_hip${v:+_$v}.so
    CATTUS_HIP_LIB=$lib timeout -k 10 200 python scripts/wino_bits.py 0 > $out/ab_${v:-base}_$r.txt 2>$out/ab_${v:-base}_$r.err || exit 1
    python - "$out/ab_${v:-base}_$r.txt" "${v:-base}" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["chess20x256_b256"]
print(f"{sys.argv[2]:10s} {c['sha256']} {min(c['launch_us']):.2f} {sorted(c['launch_us'])[2]:.2f}", flush=True)
PY
  done
done
