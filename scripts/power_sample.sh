#!/bin/bash
# Samples rocm-smi power / clocks while the f16x2 (or DTYPE) tower runs back to back: is the conv kernel power-limited?
#   bash scripts/power_sample.sh [DTYPE] > gpurun_out/power_<dtype>.txt
DT=${1:-f16x2}
python3 - "$DT" <<'PY' &
import sys, time
sys.path.insert(0, ".")
import bench
from cattus_amd.evaluator import HipEvaluator
d, blob, planes = bench.make_workload("chess20x256")
ev = HipEvaluator(blob, batch_size=len(planes), plane_words=planes.shape[2], dtype=sys.argv[1])
t0 = time.time()
while time.time() - t0 < 14:
    us, n = ev.time_tower(len(planes), 200)
print("tower launch us (last 200 forwards):", round(us, 2), flush=True)
ev.close()
PY
PID=$!
sleep 5
for i in 1 2 3 4; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | grep -v "^=" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'
  echo
  sleep 2
done
wait $PID
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
