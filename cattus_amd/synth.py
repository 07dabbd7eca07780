"""Seeded synthetic leaf positions for benches and parity tests.

Planes are laid out as the C-ABI takes them: uint64 ``[n][planes][plane_words]`` with bit
``h*S+w`` of a plane in bit ``i & 63`` of word ``i >> 6``.
"""

from __future__ import annotations

import numpy as np

from .weights import splitmix64, uniform01

_FULL = 0xFFFFFFFFFFFFFFFF


def random_hex_planes(n: int, size: int, seed: int) -> np.ndarray:
    """Each cell empty/red/blue with p = (.5,.25,.25); plane 2 all ones; u128 planes as lo,hi."""
    hw = size * size
    u = uniform01(seed * 1000003 + 17, n * hw).reshape(n, hw)
    out = np.zeros((n, 3, 2), dtype=np.uint64)
    full = (1 << hw) - 1
    for b in range(n):
        red = blue = 0
        for i in range(hw):
            if u[b, i] >= 0.75:
                blue |= 1 << i
            elif u[b, i] >= 0.5:
                red |= 1 << i
        for c, bits in enumerate((red, blue, full)):
            out[b, c, 0] = np.uint64(bits & _FULL)
            out[b, c, 1] = np.uint64(bits >> 64)
    return out


def random_chess_planes(n: int, seed: int) -> np.ndarray:
    """12 piece planes with ~28 squares occupied disjointly, each castle plane all-ones w.p. 0.5,
    en-passant plane one-hot w.p. 0.1, plane 17 all ones."""
    r = splitmix64(seed * 7778777 + 5, n * 80).reshape(n, 80)
    out = np.zeros((n, 18, 1), dtype=np.uint64)
    for b in range(n):
        occ = int(r[b, 0]) & int(r[b, 1])
        occ |= int(r[b, 2]) & int(r[b, 3]) & ~occ
        planes = [0] * 12
        k = 0
        for sq in range(64):
            if occ >> sq & 1:
                planes[int(r[b, 4 + k % 60]) % 12] |= 1 << sq
                k += 1
        for i in range(12):
            out[b, i, 0] = np.uint64(planes[i])
        for i in range(4):
            out[b, 12 + i, 0] = np.uint64(_FULL if int(r[b, 70 + i]) & 1 else 0)
        if int(r[b, 75]) % 10 == 0:
            out[b, 16, 0] = np.uint64(1 << (32 + int(r[b, 76]) % 8))
        out[b, 17, 0] = np.uint64(_FULL)
    return out


def random_ttt_planes(n: int, seed: int) -> np.ndarray:
    u = uniform01(seed * 31337 + 3, n * 9).reshape(n, 9)
    out = np.zeros((n, 3, 1), dtype=np.uint64)
    for b in range(n):
        x = o = 0
        for i in range(9):
            if u[b, i] >= 0.7:
                o |= 1 << i
            elif u[b, i] >= 0.4:
                x |= 1 << i
        out[b, :, 0] = [x, o, 511]
    return out
