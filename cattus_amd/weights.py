"""Weight blobs for the ConvNetV1 policy/value network.

The reference keeps a network as a torch ``state_dict`` (``model.pt``,
training/cattus_train/train_process.py:369) and hands third-party runtimes an
exported graph.  The HIP evaluator instead takes one flat little-endian blob:
a 64-byte header followed by the raw f32 tensors of the ``state_dict`` in the
order listed by :func:`tensor_specs` (the key names follow
training/cattus_train/net_utils.py:45-89).  BatchNorm folding happens inside
the evaluator, so a blob is a loss-free copy of the reference's checkpoint.

``seeded_blob`` draws weights from a splitmix64 stream implemented here in
numpy, not from torch's RNG, so the GPU box can rebuild the large benchmark
networks (chess 20x256, 40x384) bit-for-bit from ``(desc, seed)`` alone.
"""

from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

MAGIC = b"CATTUSW1"
HEADER_BYTES = 64
FC_HIDDEN = 128  # nn.Linear(vhc*H*W, 128), net_utils.py:71
BN_EPS = 1e-5  # torch default, net_utils.py:14,30,33 do not override it


@dataclass(frozen=True)
class NetDesc:
    """Shape of one ConvNetV1 (net_utils.py:45-57)."""

    planes: int  # C, input planes
    board: int  # S, board edge
    moves: int  # M, policy logits
    blocks: int  # residual_block_num
    filters: int  # residual_filter_num
    vhc: int  # value_head_conv_output_channels_num
    phc: int  # policy_head_conv_output_channels_num

    @property
    def hw(self) -> int:
        return self.board * self.board

    @property
    def simple(self) -> bool:
        """``SimpleTwoHeadedModel`` (net_utils.py:92-121: two dense layers and two dense heads on the flattened planes), the
        reference's other ``model.type``; encoded as ``filters == 0`` (no conv tower)."""
        return self.filters == 0

    @property
    def features(self) -> int:
        return self.planes * self.hw

    def flops_per_position(self) -> int:
        """2*MAC count, BN/activations excluded."""
        if self.simple:
            k = self.features
            return 2 * k * k * 2 + 2 * k + 2 * k * self.moves
        c, f, n, hw = self.planes, self.filters, self.blocks, self.hw
        return (
            2 * c * f * 9 * hw
            + n * 4 * f * f * 9 * hw
            + 2 * f * self.vhc * hw
            + 2 * self.vhc * hw * FC_HIDDEN
            + 2 * FC_HIDDEN
            + 2 * f * self.phc * hw
            + 2 * self.phc * hw * self.moves
        )

    def conv_flops_per_position(self) -> int:
        """3x3 conv stack only: the work the MFMA tower kernels execute."""
        c, f, n, hw = self.planes, self.filters, self.blocks, self.hw
        return 2 * c * f * 9 * hw + n * 4 * f * f * 9 * hw


# game presets used across tests and the bench (SURVEY.md section 8a sizes)
CHESS = dict(planes=18, board=8, moves=1880)
TTT = dict(planes=3, board=3, moves=9)

# u64 words per bitboard plane at the C-ABI seam and in .traindata records: chess u64 -> 1,
# ttt u16 widened to u64 -> 1 (serialize/ttt.rs:19-22), hex u128 -> 2 as lo,hi (serialize/hex.rs:19-24)
PLANE_WORDS = {"chess": 1, "ttt": 1, "hex": 2}


def hex_game(size: int) -> dict:
    return dict(planes=3, board=size, moves=size * size)


def simple_desc(planes: int, board: int, moves: int) -> NetDesc:
    """Shape of a ``SimpleTwoHeadedModel`` for a game (net_utils.py:92-104)."""
    return NetDesc(planes, board, moves, 0, 0, 0, 0)


def tensor_specs(d: NetDesc) -> list[tuple[str, tuple[int, ...]]]:
    """(state_dict key, shape) in blob order."""
    if d.simple:  # net_utils.py:101-110
        k = d.features
        return [
            ("_dense1.weight", (k, k)),
            ("_dense1.bias", (k,)),
            ("_dense2.weight", (k, k)),
            ("_dense2.bias", (k,)),
            ("_value_head.weight", (1, k)),
            ("_value_head.bias", (1,)),
            ("_policy_head.weight", (d.moves, k)),
            ("_policy_head.bias", (d.moves,)),
        ]
    f, hw = d.filters, d.hw
    specs: list[tuple[str, tuple[int, ...]]] = [
        ("_conv1._conv.weight", (f, d.planes, 3, 3)),
        ("_conv1._bn.weight", (f,)),
        ("_conv1._bn.bias", (f,)),
        ("_conv1._bn.running_mean", (f,)),
        ("_conv1._bn.running_var", (f,)),
    ]
    for i in range(d.blocks):
        p = f"_residual_blocks.{i}."
        specs += [
            (p + "_conv1.weight", (f, f, 3, 3)),
            (p + "_bn1.running_mean", (f,)),
            (p + "_bn1.running_var", (f,)),
            (p + "_conv2.weight", (f, f, 3, 3)),
            (p + "_bn2.weight", (f,)),
            (p + "_bn2.bias", (f,)),
            (p + "_bn2.running_mean", (f,)),
            (p + "_bn2.running_var", (f,)),
        ]
    specs += [
        ("_value_head.0._conv.weight", (d.vhc, f, 1, 1)),
        ("_value_head.0._bn.running_mean", (d.vhc,)),
        ("_value_head.0._bn.running_var", (d.vhc,)),
        ("_value_head.2.weight", (FC_HIDDEN, d.vhc * hw)),
        ("_value_head.2.bias", (FC_HIDDEN,)),
        ("_value_head.4.weight", (1, FC_HIDDEN)),
        ("_value_head.4.bias", (1,)),
        ("_policy_head.0._conv.weight", (d.phc, f, 1, 1)),
        ("_policy_head.0._bn.running_mean", (d.phc,)),
        ("_policy_head.0._bn.running_var", (d.phc,)),
        ("_policy_head.2.weight", (d.moves, d.phc * hw)),
        ("_policy_head.2.bias", (d.moves,)),
    ]
    return specs


def _header(d: NetDesc) -> bytes:
    h = MAGIC + struct.pack(
        "<9I", 1, d.planes, d.board, d.moves, d.blocks, d.filters, d.vhc, d.phc, FC_HIDDEN
    )
    return h + b"\0" * (HEADER_BYTES - len(h))


def parse_header(blob: bytes) -> NetDesc:
    if len(blob) < HEADER_BYTES or blob[:8] != MAGIC:
        raise ValueError("not a cattus weight blob")
    ver, c, s, m, n, f, vhc, phc, hidden = struct.unpack_from("<9I", blob, 8)
    if ver != 1 or hidden != FC_HIDDEN:
        raise ValueError(f"unsupported blob version/hidden: {ver}/{hidden}")
    return NetDesc(c, s, m, n, f, vhc, phc)


def blob_nbytes(d: NetDesc) -> int:
    return HEADER_BYTES + 4 * sum(int(np.prod(s)) for _, s in tensor_specs(d))


def pack_tensors(d: NetDesc, tensors: dict[str, np.ndarray]) -> bytes:
    parts = [_header(d)]
    for name, shape in tensor_specs(d):
        t = np.ascontiguousarray(np.asarray(tensors[name], dtype="<f4"))
        if t.shape != shape:
            raise ValueError(f"{name}: shape {t.shape} != {shape}")
        parts.append(t.tobytes())
    return b"".join(parts)


def unpack_tensors(blob: bytes) -> tuple[NetDesc, dict[str, np.ndarray]]:
    d = parse_header(blob)
    if len(blob) != blob_nbytes(d):
        raise ValueError(f"blob size {len(blob)} != {blob_nbytes(d)}")
    off, out = HEADER_BYTES, {}
    for name, shape in tensor_specs(d):
        n = int(np.prod(shape))
        out[name] = np.frombuffer(blob, dtype="<f4", count=n, offset=off).reshape(shape).copy()
        off += 4 * n
    return d, out


def blob_from_state_dict(d: NetDesc, state_dict) -> bytes:
    """Convert a reference ``model.pt`` state_dict (train_process.py:369) to a blob."""
    tensors = {}
    for name, _ in tensor_specs(d):
        t = state_dict[name]
        tensors[name] = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return pack_tensors(d, tensors)


def desc_from_state_dict(state_dict, board: int) -> NetDesc:
    """The network's shape read off a ConvNetV1 ``state_dict`` (net_utils.py:45-89): planes / filters from the stem conv,
    blocks from the residual keys, head widths and move count from the head layers."""
    shape = lambda k: tuple(state_dict[k].shape)  # noqa: E731
    if "_dense1.weight" in state_dict:  # SimpleTwoHeadedModel
        k, moves = shape("_dense1.weight")[0], shape("_policy_head.weight")[0]
        if k % (board * board) != 0:
            raise ValueError("state_dict does not belong to a SimpleTwoHeadedModel on a %dx%d board" % (board, board))
        return simple_desc(k // (board * board), board, moves)
    filters, planes = shape("_conv1._conv.weight")[:2]
    blocks = 0
    while f"_residual_blocks.{blocks}._conv1.weight" in state_dict:
        blocks += 1
    vhc, phc = shape("_value_head.0._conv.weight")[0], shape("_policy_head.0._conv.weight")[0]
    moves = shape("_policy_head.2.weight")[0]
    d = NetDesc(planes, board, moves, blocks, filters, vhc, phc)
    if shape("_policy_head.2.weight")[1] != phc * d.hw or shape("_value_head.2.weight") != (FC_HIDDEN, vhc * d.hw):
        raise ValueError("state_dict does not belong to a ConvNetV1 on a %dx%d board" % (board, board))
    return d


def blob_from_module(model, input_shape) -> bytes:
    """What the trainer's ``export_model`` calls for the ``hip`` engine (integration/rust/cattus_hip.patch,
    training/cattus_train/self_play.py): the module's ``state_dict`` as an evaluator blob.  ``input_shape`` is the
    (1, planes, S, S) tuple the trainer passes around (train_process.py:381-383)."""
    sd = model.state_dict()
    return blob_from_state_dict(desc_from_state_dict(sd, int(input_shape[-1])), sd)


def state_dict_from_blob(blob: bytes):
    """Inverse of :func:`blob_from_state_dict`; values are torch tensors."""
    import torch

    d, tensors = unpack_tensors(blob)
    sd = {k: torch.from_numpy(v.copy()) for k, v in tensors.items()}
    # BatchNorm bookkeeping entries the reference checkpoint carries but inference ignores
    for name in list(sd):
        if name.endswith("running_var"):
            sd[name[: -len("running_var")] + "num_batches_tracked"] = torch.tensor(0)
    return d, sd


# ---------------------------------------------------------------------------
# seeded initialiser (splitmix64 -> uniform floats); independent of torch's RNG
# ---------------------------------------------------------------------------

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed: int, n: int) -> np.ndarray:
    """First ``n`` outputs of the splitmix64 stream started at ``seed``."""
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) * _GOLDEN) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, n: int) -> np.ndarray:
    """``n`` f32 values in [0,1) with 24 random bits each."""
    return ((splitmix64(seed, n) >> np.uint64(40)).astype(np.float32)) * np.float32(2.0**-24)


def seeded_tensors(d: NetDesc, seed: int) -> dict[str, np.ndarray]:
    """Random-init weights incl. randomised BN statistics.

    Conv/linear weights are U(-b, b) with b = gain * sqrt(3 / fan_in) (gain 1 for
    convs, 0.5 for the head linears, keeps activations O(1) through deep towers
    and tanh unsaturated); BN running stats and
    affine parameters are perturbed away from the (0, 1, 1, 0) defaults because
    fresh-init statistics would hide folding bugs.
    """
    out: dict[str, np.ndarray] = {}
    for idx, (name, shape) in enumerate(tensor_specs(d)):
        n = int(np.prod(shape))
        u = uniform01((seed << 20) + idx * 7919 + 1, n)
        if name.endswith("running_var"):
            t = 0.5 + u  # [0.5, 1.5)
        elif name.endswith("running_mean"):
            t = 0.4 * u - 0.2
        elif "_bn2.weight" in name:
            t = 0.25 + 0.25 * u  # small residual-branch gain: the skip path stays O(1) over 40 blocks
        elif "_bn" in name and name.endswith(".weight"):
            t = 0.75 + 0.5 * u
        elif "_bn" in name and name.endswith(".bias"):
            t = 0.2 * u - 0.1
        elif name.endswith(".bias"):
            t = 0.2 * u - 0.1
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = 0.5 if "_head.2." in name or "_head.4." in name or name.startswith("_value_head.w") else 1.0  # keep tanh unsaturated
            b = np.float32(gain * np.sqrt(3.0 / fan_in))
            t = (2.0 * u - 1.0) * b
        out[name] = t.astype(np.float32).reshape(shape)
    return out


def seeded_blob(d: NetDesc, seed: int) -> bytes:
    return pack_tensors(d, seeded_tensors(d, seed))
