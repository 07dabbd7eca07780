"""Record format tests.  Restates training/tests/test_serialize_encode.py:95-151: for the reference's
own test positions the serialised record, read back the way the trainer reads it, must give exactly
the tensor planes_to_tensor produces; plus the probs/winner bytes test_serialize.rs defines."""

import numpy as np
import pytest

from cattus_amd import records
from cattus_amd import selfplay as sp
from oracle import oracle, positions

CASES = (
    [("tictactoe", s) for s in positions.TTT_TEST_POSITIONS]
    + [("hex11", s) for s in positions.HEX11_TEST_POSITIONS]
    + [("chess", s) for s in positions.CHESS_TEST_FENS]
)


@pytest.mark.parametrize("game,pos", CASES)
def test_serialize_then_unpack_equals_encode(game, pos):
    p = sp.Position(game, pos)
    assert p.turn() == 0  # every reference test position has Player1 to move
    entry = records.parse_record(game, p.test_record())
    planes_py = records.unpack_planes(entry, game)
    info = sp.game_info(game)
    tensor = oracle.planes_to_tensor(p.planes()[None], info["board"], 1)[0]  # what test_encode.rs emits
    assert planes_py.shape == tensor.shape
    assert (planes_py.astype(np.float32) == tensor).all()
    # probs idx/(n*(n-1)) in legal-move order, winner by n % 3 (test_serialize.rs:63-79)
    legal = p.legal_moves()
    n = len(legal)
    for i, (_, nn) in enumerate(legal):
        assert entry.probs[nn] == np.float32(i) / np.float32(n * (n - 1))
    assert (entry.probs >= 0).sum() == n and (entry.probs[entry.probs < 0] == -1).all()
    assert entry.winner == {0: 1.0, 1: -1.0, 2: 0.0}[n % 3]


def test_record_sizes():
    assert records.record_nbytes("chess") == 1280
    assert records.record_nbytes("hex7") == 245
    assert records.record_nbytes("hex4") == 113
    assert records.record_nbytes("tictactoe") == 61
    assert records.record_nbytes("hex11") == 6 * 8 + 121 * 4 + 1


def test_chess_record_probs_are_sorted_by_policy_index():
    # serialize/chess.rs:28-31: moves sorted by nn index before packing
    p = sp.Position("chess", positions.CHESS_TEST_FENS[2])
    raw = p.test_record()
    packed = np.frombuffer(raw, dtype="<f4", count=225, offset=18 * 8 + 235)
    legal = p.legal_moves()
    n = len(legal)
    order = sorted(range(n), key=lambda i: legal[i][1])
    want = [np.float32(i) / np.float32(n * (n - 1)) for i in order]
    assert list(packed[:n]) == want and (packed[n:] == -1).all()


def test_parse_rejects_wrong_size():
    with pytest.raises(ValueError):
        records.parse_record("chess", b"\0" * 100)
