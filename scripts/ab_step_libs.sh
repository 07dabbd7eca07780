#!/bin/bash
# Whole-step A/B of library BUILDS on one box, alternating processes: scripts/ab_step_libs.sh ROUNDS LIB [LIB ...]   (WORKLOAD=chess20x256)
rounds=$1; shift
for r in $(seq $rounds); do
  for lib in "$@"; do
    CATTUS_HIP_LIB=$lib timeout -k 10 200 python scripts/step_time.py ${WORKLOAD:-chess20x256} 2>/dev/null || exit 1
  done
done
