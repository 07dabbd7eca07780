#!/usr/bin/env python3
"""Soak of the one-launch Winograd tower: two host threads keep both lanes of ONE evaluator busy for SECONDS with batches of random
size (1 .. 256 leaves of chess 20x256), every result compared bit for bit with the per-layer launches' (a leaf's bits do not depend on
its batch); at the end the evaluator must still be on tower_wino4_kernel (a hand-off wait that gave up would have moved it to the
per-layer launches for good).      python scripts/probes/tower_soak.py [SECONDS] [blocks filters batch]"""
import json
import os
import sys
import threading
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from cattus_amd import synth  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
blocks, filters, batch = (int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (20, 256, 256)
d = NetDesc(**CHESS, blocks=blocks, filters=filters, vhc=8, phc=8)
blob = seeded_blob(d, 5)
planes = synth.random_chess_planes(batch, 5)
with HipEvaluator(blob, batch_size=batch, plane_words=1, dtype="f16x2", switches={"CATTUS_WINO_PERSIST": "0"}) as ref:
    assert ref.tower_kernel() == "conv3x3_wino4_kernel"
    want_p, want_v = ref.eval(planes)
ev = HipEvaluator(blob, batch_size=batch, plane_words=1, dtype="f16x2", switches={})
assert ev.tower_kernel() == "tower_wino4_kernel", ev.tower_kernel()
stop = time.time() + seconds
counts = [0, 0]
leaves = [0, 0]
bad = []


def worker(k):
    rng = np.random.default_rng(100 + k)
    while time.time() < stop and not bad:
        n = int(rng.integers(1, batch + 1)) if rng.random() < 0.7 else batch
        lo = int(rng.integers(0, batch - n + 1))
        p, v = ev.eval(planes[lo : lo + n])
        if not ((p == want_p[lo : lo + n]).all() and (v == want_v[lo : lo + n]).all()):
            bad.append((k, counts[k], n, lo))
        counts[k] += 1
        leaves[k] += n


t0 = time.time()
threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
for t in threads:
    t.start()
last = t0
while any(t.is_alive() for t in threads):
    time.sleep(1.0)
    if time.time() - last > 60:
        last = time.time()
        print(f"[{last - t0:.0f} s] batches {sum(counts)}", file=sys.stderr, flush=True)
for t in threads:
    t.join()
kernel = ev.tower_kernel()
ev.close()
print(json.dumps({"seconds": round(time.time() - t0, 1), "net": f"chess {blocks}x{filters}", "max_batch": batch, "threads": 2, "batches": sum(counts), "leaves": sum(leaves),
                  "mismatching_batches": len(bad), "first_mismatch": bad[:1], "tower_kernel_at_the_end": kernel,
                  "how": "scripts/probes/tower_soak.py: both lanes of one evaluator, batches of random size and offset, every output compared bit for bit with "
                         "the per-layer launches' outputs for the same leaves; one MI355X"}))
sys.exit(1 if bad or kernel != "tower_wino4_kernel" else 0)
