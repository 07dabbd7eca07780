#!/bin/bash
# 800-sim self-play: one process with 12 search threads, then two concurrent processes with 6 each, on the same GPU
python scripts/e2e_selfplay.py 12 1536 800 1536 diverse plies=10 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('one process :', d['evals_per_s'], d['steady_evals_per_s'], d['batches'], d['seconds'], d['gpu_busy_frac'])"
python scripts/e2e_selfplay.py 6 1536 800 1536 diverse plies=10 > /tmp/a.json &
P1=$!
python scripts/e2e_selfplay.py 6 1536 800 1536 diverse plies=10 > /tmp/b.json
wait $P1
python3 - <<'PY'
import json
a=json.load(open('/tmp/a.json')); b=json.load(open('/tmp/b.json'))
print('two processes:', a['evals_per_s'], b['evals_per_s'], 'sum', a['evals_per_s']+b['evals_per_s'], 'steady', a['steady_evals_per_s']+b['steady_evals_per_s'], 'batches', a['batches']+b['batches'], 'seconds', a['seconds'], b['seconds'], 'busy', a['gpu_busy_frac'], b['gpu_busy_frac'])
PY
