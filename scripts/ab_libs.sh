#!/bin/bash
# Alternating A/B of library builds on ONE box: scripts/ab_libs.sh ROUNDS CASE LIB [LIB ...]  (CASE: an index of scripts/wino_bits.py; LIB: a
# path for CATTUS_HIP_LIB).  Prints the output hash and the launch times of every run: boxes differ by several per cent, runs on one box by 0.1.
rounds=$1; case=$2; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do
    echo -n "$lib: "
    CATTUS_HIP_LIB=$lib CATTUS_WINO_KERNEL=${CATTUS_WINO_KERNEL:-k4} timeout -k 10 120 python scripts/wino_bits.py $case 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(*[(k, v['kernel'], v['sha256'], v['launch_us']) for k,v in d.items() if k!='lib'])"
  done
done
