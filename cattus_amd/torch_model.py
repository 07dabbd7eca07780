"""A torch module with the reference checkpoint's ``state_dict`` key names.

The reference saves ``model.pt`` as the ``state_dict`` of its ``ConvNetV1``
(training/cattus_train/train_process.py:369; keys listed in SURVEY.md section 3.3).  This module has the
same parameter names and arithmetic, so such a checkpoint loads with ``load_state_dict`` and converts to
an evaluator blob with :func:`cattus_amd.weights.blob_from_state_dict`.  It also serves as the plain
PyTorch fp32 cross-check of the oracle on machines where the reference's Python is absent.
"""

from __future__ import annotations

import torch
from torch import nn

from .weights import FC_HIDDEN, NetDesc, state_dict_from_blob


def _conv_bn(cin: int, cout: int, k: int, affine: bool) -> nn.Module:
    m = nn.Module()
    m._conv = nn.Conv2d(cin, cout, k, padding=k // 2, bias=False)
    m._bn = nn.BatchNorm2d(cout, affine=affine)
    return m


class _Head(nn.Sequential):
    pass


class PolicyValueNet(nn.Module):
    def __init__(self, d: NetDesc):
        super().__init__()
        self.desc = d
        f, hw = d.filters, d.hw
        self._conv1 = _conv_bn(d.planes, f, 3, affine=True)
        blocks = []
        for _ in range(d.blocks):
            b = nn.Module()
            b._conv1 = nn.Conv2d(f, f, 3, padding=1, bias=False)
            b._bn1 = nn.BatchNorm2d(f, affine=False)
            b._conv2 = nn.Conv2d(f, f, 3, padding=1, bias=False)
            b._bn2 = nn.BatchNorm2d(f, affine=True)
            blocks.append(b)
        self._residual_blocks = nn.ModuleList(blocks)
        # index positions 0, 2, 4 carry parameters (1, 3, 5 are Flatten / ReLU / Tanh in the reference)
        self._value_head = nn.ModuleDict({"0": _conv_bn(f, d.vhc, 1, affine=False), "2": nn.Linear(d.vhc * hw, FC_HIDDEN), "4": nn.Linear(FC_HIDDEN, 1)})
        self._policy_head = nn.ModuleDict({"0": _conv_bn(f, d.phc, 1, affine=False), "2": nn.Linear(d.phc * hw, d.moves)})

    @staticmethod
    def _cbr(m, x):
        return torch.relu(m._bn(m._conv(x)))

    def forward(self, x):
        x = self._cbr(self._conv1, x)
        for b in self._residual_blocks:
            t = torch.relu(b._bn1(b._conv1(x)))
            x = torch.relu(x + b._bn2(b._conv2(t)))
        v = self._cbr(self._value_head["0"], x).flatten(1)
        v = torch.tanh(self._value_head["4"](torch.relu(self._value_head["2"](v))))
        p = self._policy_head["2"](self._cbr(self._policy_head["0"], x).flatten(1))
        return p, v

    @classmethod
    def from_blob(cls, blob: bytes):
        d, sd = state_dict_from_blob(blob)
        net = SimpleTwoHeaded(d) if d.simple else cls(d)
        net.load_state_dict(sd, strict=True)
        return net.eval()


class SimpleTwoHeaded(nn.Module):
    """The reference's other ``model.type`` (``SimpleTwoHeadedModel``, net_utils.py:92-121) with its checkpoint's key names."""

    def __init__(self, d: NetDesc):
        super().__init__()
        self.desc = d
        k = d.features
        self._dense1 = nn.Linear(k, k)
        self._dense2 = nn.Linear(k, k)
        self._value_head = nn.Linear(k, 1)
        self._policy_head = nn.Linear(k, d.moves)

    def forward(self, x):
        flow = torch.relu(self._dense2(torch.relu(self._dense1(x.flatten(1)))))
        return self._policy_head(flow), torch.tanh(self._value_head(flow))
