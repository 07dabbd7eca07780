#!/usr/bin/env python3
"""Gate A for a Winograd F(2x2, 3x3) form of the split-precision tower: numerics, on the CPU, before any kernel.

The split tower (dtype f16x2) computes a 3x3 conv as three f16 MFMA terms per product over K = 9 * cin.  Winograd
F(2x2, 3x3) would need 16 instead of 36 multiplies per 2x2 output tile and input channel (2.25x fewer MFMAs).  What it
does to the result is emulated here operation for operation in the arithmetic such a kernel would use:

  weights      U = G g G^T from the BatchNorm-folded f32 weights, in float64 on the host, scaled per output channel by a
               power of two (largest |U| in [2^10, 2^11)), split into hi = f16(U), lo = f16(U - hi)
  activations  stored as today: pairs (hi, lo) of f16, saturating at 65504
  input        V = B^T d B in f32 on d = hi + lo, then split into (hi, lo)
  products     M[f] = sum over cin of  U_hi V_hi + U_lo V_hi + U_hi V_lo  -- f16 x f16 products are exact in f32, the
               sums are f32 (torch f32 matmul stands in for the MFMA's f32 accumulation)
  output       Y = A^T M A in f32, * 2^-s, + bias, + skip, ReLU, stored as a pair again

and compared with (a) the same emulation of today's direct form and (b) the network's float64 run, on the reference-made
fixture tests/golden/chess_20x256.npz and on a seeded chess 40x384 batch.  The gate: proceed only if
the Winograd form stays inside the reference's cross-runtime tolerance (training/tests/test_net_output.py:28-33) and at
most 2x today's error against the float64 run.

    python scripts/winograd_gate_a.py [--leaves40 8]        # CPU only, about a minute
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from cattus_amd import synth  # noqa: E402
from cattus_amd.torch_model import PolicyValueNet  # noqa: E402
from cattus_amd.weights import CHESS, NetDesc, seeded_blob  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
F16_MAX = 65504.0


def split(x: torch.Tensor):
    """f32 -> (hi, lo) as f32 tensors holding f16 values; x saturates at +-65504."""
    x = x.clamp(-F16_MAX, F16_MAX)
    hi = x.to(torch.float16).to(torch.float32)
    lo = (x - hi).to(torch.float16).to(torch.float32)
    return hi, lo


def fold(conv_w, bn):
    """BatchNorm (eval) folded into the bias-free conv in f32, as cattus_amd/csrc/evaluator.hip::fold_conv does."""
    var, mean = bn.running_var.float(), bn.running_mean.float()
    g = bn.weight.float() if bn.affine else torch.ones_like(var)
    b = bn.bias.float() if bn.affine else torch.zeros_like(var)
    scale = g / torch.sqrt(var + torch.tensor(1e-5, dtype=torch.float32))
    return (conv_w.float() * scale[:, None, None, None]).contiguous(), b - mean * scale


def pow2_scale(w64: torch.Tensor):
    """Per output channel: 2^s with the channel's largest |w * 2^s| in [2^10, 2^11)."""
    m = w64.abs().flatten(1).max(dim=1).values.clamp_min(1e-30)
    s = 10 - torch.floor(torch.log2(m))
    return torch.pow(torch.tensor(2.0, dtype=torch.float64), s)


def conv_direct_split(x, w, bias, skip):
    """Today's form: out = conv(a_hi, w_hi) + conv(a_lo, w_hi) + conv(a_hi, w_lo), f32 accumulation."""
    sc = pow2_scale(w.double())
    wh, wl = split((w.double() * sc[:, None, None, None]).float())
    ah, al = split(x)
    acc = F.conv2d(al, wh, padding=1) + F.conv2d(ah, wl, padding=1) + F.conv2d(ah, wh, padding=1)
    y = acc * (1.0 / sc).float()[None, :, None, None] + bias[None, :, None, None]
    if skip is not None:
        y = y + skip
    return torch.relu(y)


def conv_winograd_split(x, w, bias, skip):
    B, C, S, _ = x.shape
    assert S % 2 == 0
    T = S // 2
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G)  # [cout, cin, 4, 4], float64 on the host
    sc = pow2_scale(U)
    Uh, Ul = split((U * sc[:, None, None, None]).float())
    xa = sum(split(x))  # what the kernel reads back: hi + lo, exact in f32
    xp = F.pad(xa, (1, 1, 1, 1))
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)  # [B, C, T, T, 4, 4]
    bt = BT.float()
    V = torch.einsum("ij,bctujk,lk->bctuil", bt, tiles, bt)  # f32 adds
    Vh, Vl = split(V)
    # per frequency f = (i, l): M[b, o, t, u] = sum_c U[o, c, f] V[b, c, t, u, f]
    def mm(Ux, Vx):
        return torch.einsum("ocil,bctuil->botuil", Ux, Vx)
    M = mm(Ul, Vh) + mm(Uh, Vl) + mm(Uh, Vh)
    at = AT.float()
    Y = torch.einsum("pi,botuil,ql->botupq", at, M, at)  # [B, O, T, T, 2, 2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, S, S)
    y = Y * (1.0 / sc).float()[None, :, None, None] + bias[None, :, None, None]
    if skip is not None:
        y = y + skip
    return torch.relu(y)


def tower(net, x, conv):
    w, b = fold(net._conv1._conv.weight, net._conv1._bn)
    a = sum(split(conv_direct_split(x, w, b, None)))  # the stem reads 0/1 planes of 18 channels: left direct in both forms
    for blk in net._residual_blocks:
        w1, b1 = fold(blk._conv1.weight, blk._bn1)
        w2, b2 = fold(blk._conv2.weight, blk._bn2)
        t = sum(split(conv(a, w1, b1, None)))
        a = sum(split(conv(t, w2, b2, a)))
    return a


def heads(net, a):
    v = net._cbr(net._value_head["0"], a).flatten(1)
    v = torch.tanh(net._value_head["4"](torch.relu(net._value_head["2"](v))))
    p = net._policy_head["2"](net._cbr(net._policy_head["0"], a).flatten(1))
    return p, v.flatten()


def ref_tol_ok(p, v, pr, vr):
    ok_p = np.isclose(p, pr, rtol=1e-3, atol=1e-6).all()
    ok_v = all(abs(a - b) <= max(1e-5 * max(abs(a), abs(b)), 1e-6) for a, b in zip(v, vr))
    return bool(ok_p and ok_v)


def run(name, blob, planes, p_ref32=None, v_ref32=None):
    net = PolicyValueNet.from_blob(blob)
    d = net.desc
    bits = np.unpackbits(planes.view(np.uint8).reshape(len(planes), d.planes, -1), axis=-1, bitorder="little")
    x = torch.from_numpy(bits[..., : d.hw].reshape(len(planes), d.planes, d.board, d.board).astype(np.float32))
    with torch.no_grad():
        p64, v64 = net.double()(x.double())
        p64, v64 = p64.numpy(), v64.flatten().numpy()
        net = net.float()
        p32, v32 = net(x)
        p32, v32 = p32.numpy(), v32.flatten().numpy()
        out = {"net": name, "leaves": len(planes), "torch_f32_vs_f64": dict(dp=float(np.abs(p32 - p64).max()), dv=float(np.abs(v32 - v64).max()))}
        if p_ref32 is None:
            p_ref32, v_ref32 = p32, v32
        for form, conv in (("direct_split", conv_direct_split), ("winograd_split", conv_winograd_split)):
            p, v = heads(net, tower(net, x, conv))
            p, v = p.numpy(), v.numpy()
            out[form] = dict(dp_vs_f64=float(np.abs(p - p64).max()), dv_vs_f64=float(np.abs(v - v64).max()),
                             dp_vs_ref_f32=float(np.abs(p - p_ref32).max()), dv_vs_ref_f32=float(np.abs(v - v_ref32).max()),
                             within_reference_tolerance=ref_tol_ok(p, v, p_ref32, v_ref32))
    out["winograd_over_direct"] = dict(dp=out["winograd_split"]["dp_vs_f64"] / out["direct_split"]["dp_vs_f64"],
                                       dv=out["winograd_split"]["dv_vs_f64"] / out["direct_split"]["dv_vs_f64"])
    out["gate_a_pass"] = bool(out["winograd_split"]["within_reference_tolerance"] and out["winograd_over_direct"]["dp"] <= 2.0
                              and out["winograd_over_direct"]["dv"] <= 2.0)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--leaves40", type=int, default=8, help="leaves of the seeded chess 40x384 batch")
    ap.add_argument("--leaves20", type=int, default=32, help="leaves of the seeded chess 20x256 batch (the bench weights)")
    args = ap.parse_args()
    torch.manual_seed(0)
    results = []
    from helpers import blob_for

    d, blob, z = blob_for("chess_20x256")
    results.append(run("chess_20x256 (reference-made fixture)", blob, z["planes"], z["policy"], z["value"]))
    print(json.dumps(results[-1]), flush=True)
    d20 = NetDesc(**CHESS, blocks=20, filters=256, vhc=8, phc=8)
    results.append(run("chess20x256 bench weights (seed 2)", seeded_blob(d20, 2), synth.random_chess_planes(args.leaves20, 2)))
    print(json.dumps(results[-1]), flush=True)
    d40 = NetDesc(**CHESS, blocks=40, filters=384, vhc=8, phc=8)
    results.append(run("chess40x384 (seed 3)", seeded_blob(d40, 3), synth.random_chess_planes(args.leaves40, 3)))
    print(json.dumps(results[-1]), flush=True)
    print(json.dumps({"gate_a": all(r["gate_a_pass"] for r in results), "results": results}))


if __name__ == "__main__":
    main()
