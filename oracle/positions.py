"""Position strings -> network input planes (pure Python; TEST INFRASTRUCTURE ONLY).

Restates, for the fixtures the reference's own tests use:
  - ttt / hex position strings   training/self-play/src/test_util.rs:7-66
  - ttt planes                   engine/src/ttt/net.rs:14-24
  - hex planes                   engine/src/hex/net.rs:14-24 (+ serialize/hex.rs:19-24 lo,hi split)
  - chess planes                 engine/src/chess/net/mod.rs:19-60
The reference evaluates positions flipped so that Player1 is to move
(engine/src/net/mod.rs:79,158-164); ``*_planes`` here do that flip too.

Chess FEN parsing follows the un-vendored ``chess`` 3.2.0 crate as the reference uses
it: only the FILE of the FEN en-passant field is read, the recorded square is the
pawn's square (4th rank of the side that just moved), and it is kept only if an enemy
pawn stands next to it (``Board::set_ep``).  That crate is not in /root/reference, so
this detail is restated from its published behaviour (parity unpinned).
"""

from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- ttt


def ttt_from_str(s: str):
    """'xo_ ... ' 9 cells + turn char -> (board_x, board_o, turn) ; turn 1 = x, 2 = o."""
    assert len(s) == 10
    bx = bo = 0
    for i, ch in enumerate(s[:9]):
        if ch == "x":
            bx |= 1 << i
        elif ch == "o":
            bo |= 1 << i
        else:
            assert ch == "_"
    turn = {"x": 1, "o": 2}[s[9]]
    return bx, bo, turn


def ttt_planes(s: str) -> np.ndarray:
    bx, bo, turn = ttt_from_str(s)
    if turn == 2:  # flipped(): swap boards (ttt/core.rs:237-244)
        bx, bo = bo, bx
    return np.array([[bx], [bo], [(1 << 9) - 1]], dtype=np.uint64)


# --------------------------------------------------------------------------- hex


def hex_from_str(s: str, size: int):
    assert len(s) == size * size + 1
    red = blue = 0
    for i, ch in enumerate(s[: size * size]):
        if ch == "r":
            red |= 1 << i
        elif ch == "b":
            blue |= 1 << i
        else:
            assert ch == "e"
    turn = {"r": 1, "b": 2}[s[-1]]
    return red, blue, turn


def hex_transpose(bits: int, size: int) -> int:
    out = 0
    for r in range(size):
        for c in range(size):
            if bits >> (r * size + c) & 1:
                out |= 1 << (c * size + r)
    return out


def _u128_words(x: int) -> list[int]:
    return [x & 0xFFFFFFFFFFFFFFFF, (x >> 64) & 0xFFFFFFFFFFFFFFFF]


def hex_planes_raw(red: int, blue: int, turn: int, size: int) -> np.ndarray:
    if turn == 2:  # flipped(): transpose + swap colours (hex/core.rs:324-334)
        red, blue = hex_transpose(blue, size), hex_transpose(red, size)
    full = (1 << (size * size)) - 1
    # always two words per plane: the reference's hex bitboard is a u128 written lo,hi
    return np.array([_u128_words(p) for p in (red, blue, full)], dtype=np.uint64)


def hex_planes(s: str, size: int) -> np.ndarray:
    return hex_planes_raw(*hex_from_str(s, size), size)


# ------------------------------------------------------------------------- chess

_PIECES = "PNBRQK"


def chess_from_fen(fen: str):
    """-> dict(pieces[2][6] bitboards, white_to_move, castle 'KQkq' subset, ep pawn square or None)."""
    parts = fen.split()
    rows = [r for r in parts[0].split("/") if r != ""]
    assert len(rows) == 8
    pieces = [[0] * 6 for _ in range(2)]
    for ri, row in enumerate(rows):
        rank = 7 - ri
        file = 0
        for ch in row:
            if ch.isdigit():
                file += int(ch)
            else:
                color = 0 if ch.isupper() else 1
                pieces[color][_PIECES.index(ch.upper())] |= 1 << (rank * 8 + file)
                file += 1
        assert file == 8
    wtm = parts[1] == "w"
    castle = "" if parts[2] == "-" else parts[2]
    ep = None
    if parts[3] != "-":
        f = ord(parts[3][0]) - ord("a")
        # pawn square: 4th rank of the side that just moved (rank index 4 if white is to move)
        rank = 4 if wtm else 3
        sq = rank * 8 + f
        mover_pawns = pieces[0 if wtm else 1][0]
        adj = 0
        if f > 0:
            adj |= 1 << (sq - 1)
        if f < 7:
            adj |= 1 << (sq + 1)
        if adj & mover_pawns:
            ep = sq
    return dict(pieces=pieces, wtm=wtm, castle=castle, ep=ep)


def _flip_ranks(bb: int) -> int:
    out = 0
    for r in range(8):
        out |= ((bb >> (8 * r)) & 0xFF) << (8 * (7 - r))
    return out


def chess_planes_from_pos(pos) -> np.ndarray:
    pieces, castle, ep = pos["pieces"], pos["castle"], pos["ep"]
    if not pos["wtm"]:  # flipped(): mirror ranks, swap colours/castle rights, keep ep file (core.rs:366-399)
        pieces = [[_flip_ranks(b) for b in pieces[1]], [_flip_ranks(b) for b in pieces[0]]]
        castle = castle.swapcase()
        if ep is not None:
            ep = (7 - ep // 8) * 8 + ep % 8
    full = (1 << 64) - 1
    planes = [pieces[0][i] for i in range(6)] + [pieces[1][i] for i in range(6)]
    planes += [full if ch in castle else 0 for ch in "KQkq"]
    planes.append(0 if ep is None else 1 << ep)
    planes.append(full)
    return np.array(planes, dtype=np.uint64).reshape(18, 1)


def chess_planes(fen: str) -> np.ndarray:
    return chess_planes_from_pos(chess_from_fen(fen))


# ------------------------------------------------------------- reference fixtures

# training/tests/test_net_output.py:138-145 (same list test_serialize_encode.py:25-32)
TTT_TEST_POSITIONS = [
    "___x__o_ox",
    "o_xx_x__ox",
    "o__xo__xxx",
    "o___x_o__x",
    "oo__x____x",
    "oo__o__oxx",
]

# training/tests/test_net_output.py:153-189 (hex11)
HEX11_TEST_POSITIONS = [
    "r" + "e" * 120 + "r",
    ("rererererer" + "ererererere") * 5 + "rererererer" + "r",
    "".join("e" * i + "r" + "e" * (10 - i) for i in range(11)) + "r",
]

# training/tests/test_net_output.py:198-204
CHESS_TEST_FENS = [
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
    "nnqrkbbr/pppppppp/8/8/8/8/PPPPPPPP/NNQRKBBR w - - 0 1",
    "8/1p6/3QR3/6k1/1P2b3/2P3K1/6b1/6r1/ w - - 0 1",
    "4k2r/6r1/8/8/8/8/3R4/R3K3 w Qk - 0 1",
    "rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR w KQkq e3 0 1",
]
