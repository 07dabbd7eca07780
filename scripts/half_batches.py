#!/usr/bin/env python3
"""Does a 256-leaf batch run faster as two 128-leaf halves on the evaluator's two lanes (two streams, 128 workgroups per conv
launch each when CATTUS_CONV_CB=2 is forced), whose layer boundaries then fall at different times?
    python scripts/half_batches.py            (prints leaves/s for: one batch of 256; two halves in flight, default tile; forced 64-cout tile)"""
import os
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
from cattus_amd.evaluator import HipEvaluator  # noqa: E402

d, blob, planes = bench.make_workload("chess20x256")
dev = torch.device("cuda", 0)
d_planes = torch.from_numpy(planes.view("int64")).to(dev)
half = d_planes[128:].contiguous()


def run(mode, cb=None, steps=200):
    if cb:
        os.environ["CATTUS_CONV_CB"] = cb
    else:
        os.environ.pop("CATTUS_CONV_CB", None)
    ev = HipEvaluator(blob, batch_size=256, plane_words=1, dtype="f16x2")
    pol = [torch.empty((256, d.moves), dtype=torch.float32, device=dev) for _ in range(2)]
    val = [torch.empty((256,), dtype=torch.float32, device=dev) for _ in range(2)]
    st = [ev.lane_stream(0), ev.lane_stream(1)]

    def step():
        if mode == "one":
            ev.eval_device(d_planes.data_ptr(), 256, pol[0].data_ptr(), val[0].data_ptr(), st[0], lane=0)
        else:
            ev.eval_device(d_planes.data_ptr(), 128, pol[0].data_ptr(), val[0].data_ptr(), st[0], lane=0)
            ev.eval_device(half.data_ptr(), 128, pol[1].data_ptr(), val[1].data_ptr(), st[1], lane=1)

    for _ in range(100):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ev.close()
    return 256 * steps / dt


for label, mode, cb in (("one batch of 256", "one", None), ("two halves of 128, default tile", "two", None),
                        ("two halves of 128, 64-cout tile forced", "two", "2"), ("one batch of 256", "one", None)):
    print(f"{label}: {run(mode, cb):.0f} leaves/s", flush=True)
