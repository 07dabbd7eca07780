"""Game-rule tests for the C++ host library, restating the reference's own unit tests
(engine/src/ttt/core.rs:290-374, engine/src/hex/core.rs:386-548, engine/src/chess/core.rs:617-729)
plus public perft counts for the chess move generator."""

import hashlib
import random

import numpy as np
import pytest

from cattus_amd import selfplay as sp
from cattus_amd.evaluator import ABI_SYMBOLS as HIP_SYMBOLS
from oracle import positions

GOLDEN = __import__("pathlib").Path(__file__).resolve().parent / "golden"


def P(game, s=None):
    return sp.Position(game, s)


# ------------------------------------------------------------------------------------------ ABI


def test_selfplay_library_exports_every_declared_symbol():
    lib = sp.load_library()
    header = (GOLDEN.parent.parent / "include" / "cattus_selfplay.h").read_text()
    for name in sp.ABI_SYMBOLS:
        assert name in header
        getattr(lib, name)
    import re

    declared = set(re.findall(r"\b(cattus_sp_\w+)\s*\(", header))
    assert declared == set(sp.ABI_SYMBOLS)


def test_hip_library_loads_and_exports_every_declared_symbol():
    from cattus_amd import evaluator

    lib = evaluator.load_library()
    header = (GOLDEN.parent.parent / "include" / "cattus_hip.h").read_text()
    import re

    declared = set(re.findall(r"\b(cattus_hip_\w+)\s*\(", header))
    assert declared == set(HIP_SYMBOLS)
    for name in HIP_SYMBOLS:
        getattr(lib, name)
    assert b"gfx950" in lib.cattus_hip_version()


def test_the_native_library_leaves_the_environment_alone_and_the_package_default_is_opt_out():
    """HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) belongs to the host process: loading
    libcattus_hip.so must not change the environment (no setenv from a library constructor); `import cattus_amd` sets the
    default unless CATTUS_NO_ENV_DEFAULTS=1, and never overrides an explicit value."""
    import os
    import subprocess
    import sys

    from cattus_amd import evaluator

    lib = evaluator.load_library()._name
    getenv = ("libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
              "print((libc.getenv(b'HIP_FORCE_DEV_KERNARG') or b'unset').decode())")
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_FORCE_DEV_KERNARG", "CATTUS_NO_ENV_DEFAULTS")}

    def run(code, **extra):
        return subprocess.run([sys.executable, "-c", "import ctypes, sys; " + code, lib], env=dict(env, **extra), capture_output=True, text=True,
                              check=True, cwd=str(evaluator._PKG.parent)).stdout.strip()

    assert run("ctypes.CDLL(sys.argv[1]); " + getenv) == "unset"
    assert run("import cattus_amd; " + getenv) == "1"
    assert run("import cattus_amd; " + getenv, CATTUS_NO_ENV_DEFAULTS="1") == "unset"
    assert run("import cattus_amd; " + getenv, HIP_FORCE_DEV_KERNARG="0") == "0"


# ------------------------------------------------------------------------------------------ ttt


@pytest.mark.parametrize(
    "s,winner",
    [
        ("xxxoo____o", 1),
        ("oo_xxx___o", 1),
        ("oo____xxxo", 1),
        ("oxxo__ox_x", -1),
        ("xox_o_xo_x", -1),
        ("xxo__o_xox", -1),
        ("xxoooxxxoo", 0),
    ],
)
def test_ttt_terminal_positions(s, winner):  # ttt/core.rs:290-318
    assert P("tictactoe", s).status() == winner


@pytest.mark.parametrize("s", ["oxx_o_o__o", "o_____xx_o", "xx_xx_xo_o", "ox___x_xox", "_x_o__o_xo", "ox__o____x", "_o__o_oxxo", "__xx_x__ox"])
def test_ttt_flip(s):  # ttt/core.rs:321-338
    p = P("tictactoe", s)
    assert p.turn() != p.flipped().turn()
    assert p.flipped().flipped() == p


def _flip_rand(game, games, seed):
    """flip_rand of the reference: flip∘flip = id, legal moves commute with flip, statuses mirror."""
    rng = random.Random(seed)
    for _ in range(games):
        p = P(game)
        while p.status() == "ongoing":
            t = p.flipped()
            assert t.flipped() == p
            moves = p.legal_moves()
            back = {t.flipped_move_nn(k) for k in range(len(t.legal_moves()))}
            # nn indices are defined for Player1-to-move geometry; compare as sets of (name) via double flip
            names = {n for n, _ in moves}
            names_tt = set()
            for k, (n, _) in enumerate(t.legal_moves()):
                names_tt.add(_flip_name(game, n))
            assert names == names_tt
            assert len(back) == len(moves)
            st, stt = p.status(), t.status()
            assert (st == "ongoing") == (stt == "ongoing")
            p = p.moved(rng.randrange(len(moves)))
        assert p.flipped().status() == -p.status()


def _flip_name(game, name):
    if game == "chess":
        fr = lambda sq: sq[0] + str(9 - int(sq[1]))  # noqa: E731
        return fr(name[0:2]) + fr(name[2:4]) + name[4:]
    if game == "tictactoe":
        return name
    r, c = name.strip("()").split(", ")
    return f"({c}, {r})"


def test_ttt_flip_rand():
    _flip_rand("tictactoe", 100, 1)


# ------------------------------------------------------------------------------------------ hex


def _hex_rows(rows, turn):
    return "".join(rows) + turn


def test_hex_short_diagonal_wins():  # hex/core.rs:386-423
    red = _hex_rows(["e" * i + "r" + "e" * (10 - i) for i in range(11)], "b")
    blue = _hex_rows(["e" * i + "b" + "e" * (10 - i) for i in range(11)], "r")
    assert P("hex11", red).status() == 1
    assert P("hex11", blue).status() == -1


def test_hex_almost_short_diagonal_does_not_win():  # hex/core.rs:425-461
    red = _hex_rows(["e" * 11] + ["e" * i + "r" + "e" * (10 - i) for i in range(1, 11)], "b")
    blue = _hex_rows(["e" * i + "b" + "e" * (10 - i) for i in range(10)] + ["e" * 11], "r")
    assert P("hex11", red).status() == "ongoing"
    assert P("hex11", blue).status() == "ongoing"


def test_hex_long_diagonal_does_not_win():  # hex/core.rs:463-490
    red = _hex_rows(["e" * (10 - i) + "r" + "e" * i for i in range(11)], "b")
    blue = _hex_rows(["e" * (10 - i) + "b" + "e" * i for i in range(11)], "r")
    assert P("hex11", red).status() == "ongoing"
    assert P("hex11", blue).status() == "ongoing"


def test_hex_flip():  # hex/core.rs:493-511
    rows = ["eebeeeeeeer", "eeeeeeeeeee", "eeeebeeeree", "eeeeeeereee", "eeeeeereeee", "eeeeereeeee", "eeeerebeeee",
            "eeereeeeeee", "eereeereeee", "ereeeeeeeee", "reeeeebeeee"]
    p = P("hex11", _hex_rows(rows, "b"))
    assert p.turn() == 1 and p.flipped().turn() == 0
    assert p.flipped().flipped() == p


@pytest.mark.parametrize("game", ["hex4", "hex7", "hex11"])
def test_hex_flip_rand(game):
    _flip_rand(game, 30, 2)


def test_hex_planes_match_python_restatement():
    for s in positions.HEX11_TEST_POSITIONS:
        assert (P("hex11", s).planes() == positions.hex_planes(s, 11)).all()


# ---------------------------------------------------------------------------------------- chess


def _play(p, moves):
    for m in moves:
        assert p.status() == "ongoing"
        p = p.play(m)
    return p


def test_chess_simple_game_and_mate():  # chess/core.rs:617-630 (SAN line given here in LAN)
    line = "e2e4 e7e5 d2d4 e5d4 d1d4 b8c6 d4a4 a7a6 c1g5 h7h6 f1c4 a8b8 a4b3 b8a8 c4f7".split()
    assert _play(P("chess"), line).status() == 1


def test_chess_fifty_rule_count():  # chess/core.rs:633-671
    line = ["e2e4", "e7e5"] + ["e1e2", "e8e7", "e2e1", "e7e8"] * 24 + ["e1e2", "e8e7", "e2e1"]
    assert _play(P("chess"), line).status() == 0


@pytest.mark.parametrize(
    "fen",
    [
        "7r/2B3n1/K5R1/3nPP2/P1k2Pp1/4p1p1/2p4P/8 w - - 0 1",
        "8/5pB1/1P4P1/1p3q2/BK1pP2P/pQ1pP3/7k/8 w - - 0 1",
        "2k5/2b1p1P1/1p5P/P1pnp3/QPK1b2p/8/5r2/8 b - - 0 1",
        "2b5/3N4/2B4p/1KP1R2r/n5p1/3N3P/k2B3b/q7 w - - 0 1",
        "8/2P1r3/4k1p1/4n1p1/3Rq3/P3Pp2/P4PP1/5K1N b - - 0 1",
        "1N6/5pP1/P1pK3P/P3p3/2R3p1/k3pQ2/6r1/6N1 b - - 0 1",
        "1B6/1r6/Q1ppKP2/qP1P1N1k/3p4/1pb5/7p/8 w - - 0 1",
        "3r4/1b2p3/7k/1P3R2/K4nrN/1N5P/n1Pp3P/8 b - - 0 1",
    ],
)
def test_chess_flip(fen):  # chess/core.rs:674-693
    p = P("chess", fen)
    assert p.turn() != p.flipped().turn()
    assert p.flipped().flipped() == p


def test_chess_flip_rand():  # chess/core.rs:696-729
    _flip_rand("chess", 25, 3)


PERFT = [
    ("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", [20, 400, 8902, 197281]),
    ("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", [48, 2039, 97862]),
    ("8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", [14, 191, 2812, 43238]),
    ("r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", [6, 264, 9467]),
    ("rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8", [44, 1486, 62379]),
    ("r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10", [46, 2079, 89890]),
]


@pytest.mark.parametrize("fen,counts", PERFT)
def test_chess_perft(fen, counts):
    for depth, want in enumerate(counts, start=1):
        assert sp.chess_perft(fen, depth) == want


def test_chess_planes_match_python_restatement():  # chess/net/mod.rs:19-60 on the reference's test FENs
    for fen in positions.CHESS_TEST_FENS:
        p = P("chess", fen)
        assert (p.planes() == positions.chess_planes(fen)).all(), fen


def test_chess_en_passant_is_the_pawn_square_and_needs_a_capturer():
    p = _play(P("chess"), ["e2e4", "a7a6", "e4e5", "d7d5"])  # black pawn d5 next to white pawn e5
    assert int(p.planes()[16, 0]) == 1 << 35  # d5, the pawn's square (chess/core.rs:261-262)
    assert "e5d6" in [n for n, _ in p.legal_moves()]
    q = _play(P("chess"), ["e2e4", "a7a6", "e4e5", "h7h5"])  # nobody can capture: not recorded
    assert int(q.planes()[16, 0]) == 0
    assert str(p).endswith("d6")  # FEN shows the passed-over square (chess/core.rs:256-272)


def test_chess_policy_index_table_digest():
    moves = sp.chess_nn_moves()
    assert len(moves) == 1880 and len(set(moves)) == 1880
    digest = hashlib.sha256(",".join(moves).encode()).hexdigest()
    assert digest == (GOLDEN / "chess_nn_moves.sha256").read_text().strip()
    assert moves[:4] == ["a1b1", "a1c1", "a1d1", "a1e1"] and moves[-1] == "h7h8n" and moves[1792] == "a7a8q"


def test_chess_move_nn_index_after_flip_stays_in_table():
    p = _play(P("chess"), ["e2e4"])  # black to move: flipped moves must be white-geometry moves
    for k in range(len(p.legal_moves())):
        assert 0 <= p.flipped_move_nn(k) < 1880
