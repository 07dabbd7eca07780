#!/bin/bash
# (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the first synchronisation on this image: not in the list)
# Evaluator-only bench (chess 20x256, batch 256, 200 batches) under HIP runtime switches, one box, alternating with the default.
F="--lanes 1 --settle-seconds 0 --no-bf16 --no-f16 --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32"
run() { timeout -k 5 90 env "$@" python3 bench.py --steps 200 --warmup 20 $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s ms/batch %.4f  launch_us %.2f' % ('$*', d['ms_per_step'], d['roofline']['avg_launch_us']))" || { echo "$* : failed or timed out, stopping"; exit 1; }; }
run X=0
run AMD_OPT_FLUSH=0
run ROC_USE_FGS_KERNARG=0
run ROC_USE_FGS_KERNARG=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run ROC_SKIP_KERNEL_ARG_COPY=1
run X=0
run HIP_FORCE_DEV_KERNARG=0
run GPU_MAX_HW_QUEUES=1
run ROC_ACTIVE_WAIT_TIMEOUT=0
run X=0
