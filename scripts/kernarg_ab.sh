set -u
F="--lanes 1 --no-long-run --no-cpu-baseline --selfplay-seconds 0 --agreement-plies 0 --no-f32"
for v in 0 1 0 1; do
  for wl in hex7_6x64 chess20x256; do
    HIP_FORCE_DEV_KERNARG=$v python3 bench.py --workload $wl --steps 200 --warmup 20 $F 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DEV_KERNARG=$v', '$wl', 'ms/step', round(d['ms_per_step'],5), 'launch_us', round(d['roofline']['avg_launch_us'],2))"
  done
done
HIP_FORCE_DEV_KERNARG=0 python scripts/stamps_heads.py hex7_6x64 | sed -n 1,3p
HIP_FORCE_DEV_KERNARG=1 python scripts/stamps_heads.py hex7_6x64 | sed -n 1,3p
