#!/bin/bash
# A/B of two builds of libcattus_hip on one box: evaluator-only bench (chess 20x256, 200 batches), alternating.
#   bash scripts/ab_lib.sh cattus_amd/libcattus_hip_ab.so [ROUNDS]
F="--lanes 2 --no-long-run --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32"
AB=$PWD/$1
run() { timeout -k 5 120 env "$@" python3 bench.py --steps 200 --warmup 20 $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-70s ms/batch %.4f  launch_us %.2f  two-lane %.0f' % ('$*'[-70:], d['ms_per_step'], d['roofline']['avg_launch_us'], d['two_batches_in_flight']['value']))" || { echo "$* : failed, stopping"; exit 1; }; }
for i in $(seq ${2:-2}); do
  run X=base
  run CATTUS_HIP_LIB=$AB
done
