#!/bin/bash
# A/B of two environment settings (or two builds of libcattus_hip: CATTUS_HIP_LIB=path) on ONE box, alternating, evaluator-only
# bench of the chosen dtype (chess 20x256, 200 batches):
#   bash scripts/ab_env.sh DTYPE "A_ENV=..." "B_ENV=..." [ROUNDS]
# e.g. bash scripts/ab_env.sh f16x2 X=base CATTUS_HIP_LIB=$PWD/cattus_amd/libcattus_hip_ab.so 3
DT=$1; A=$2; B=$3
F="--dtype $DT --lanes 2 --settle-seconds 0.3 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32 --no-bf16 --no-f16"
run() { timeout -k 5 120 env "$1" python3 bench.py --steps 200 --warmup 20 $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s ms/batch %.4f  launch_us %.2f  two-lane %.0f' % ('$1'[-60:], d['ms_per_step'], d['roofline']['avg_launch_us'], d['two_batches_in_flight']['value']))" || { echo "$1 : failed, stopping"; exit 1; }; }
for i in $(seq ${4:-2}); do
  run "$A"
  run "$B"
done
