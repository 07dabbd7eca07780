#!/usr/bin/env python3
"""Does the Winograd tower gain from running as two half-size batches on the evaluator's two lanes?

conv3x3_wino_kernel at batch 256 is exactly one workgroup per CU, all of them in the same phase: operand-bound loop, then a store
burst every CU issues at once.  Two 128-leaf batches on two streams occupy half the chip each and drift apart in phase, so one
batch's store burst would overlap the other's loop.  This measures it: K steps of 256 leaves on one stream against K pairs of
128 + 128 on the two lane streams (same leaves, same bits).

    CATTUS_WINOGRAD=1 python scripts/lane_split.py [--steps 200]
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("CATTUS_WINOGRAD", "1")

from cattus_amd import synth  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dtype", default="f16x2")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    d = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
    blob = seeded_blob(d, 2)
    planes = synth.random_chess_planes(256, 2)
    d_planes = torch.from_numpy(planes.view(np.int64)).to(dev)
    out = {}
    ref = None
    for name, parts in (("1x256", 1), ("2x128", 2), ("4x64", 4)):
        n = 256 // parts
        with HipEvaluator(blob, batch_size=n, plane_words=1, dtype=args.dtype, device=0) as ev:
            pol = torch.empty((256, d.moves), dtype=torch.float32, device=dev)
            val = torch.empty((256,), dtype=torch.float32, device=dev)
            streams = [ev.lane_stream(i % 2) for i in range(parts)]

            def step():
                for i in range(parts):
                    ev.eval_device(d_planes[i * n:].data_ptr(), n, pol[i * n:].data_ptr(), val[i * n:].data_ptr(), streams[i], lane=i % 2)

            for _ in range(40):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            kern = ev.tower_kernel()
        if ref is None:
            ref = (pol.clone(), val.clone())
        same = bool((pol == ref[0]).all()) and bool((val == ref[1]).all())
        out[name] = dict(ms_per_256=dt / args.steps * 1e3, node_evals_per_s=256 * args.steps / dt, kernel=kern, same_bits_as_1x256=same)
        print(json.dumps({name: out[name]}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
