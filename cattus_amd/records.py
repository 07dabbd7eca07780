"""``.traindata`` records: the read side of the self-play output.

Layouts (little-endian) as the reference's trainer parses them:
  chess   18 x u64 planes | 235-byte legal-move bitmap | 225 x f32 probs | i8 winner   (chess.py:22-49) = 1280 B
  hex N   6 x u64 planes (3 x u128 as lo,hi) | N*N x f32 probs (-1 = illegal) | i8 winner   (hex.py:22-43)
  ttt     3 x u64 planes | 9 x f32 probs | i8 winner                                       (tictactoe.py:21-41)
and ``unpack_planes`` = DataSet.unpack_planes (training/cattus_train/data_set.py:65-73).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .selfplay import game_info


@dataclass
class DataEntry:
    planes: np.ndarray  # uint64 [planes, plane_words]
    probs: np.ndarray  # float32 [moves], -1 for illegal moves
    winner: float


def record_nbytes(game: str) -> int:
    return game_info(game)["record_bytes"]


def parse_record(game: str, data: bytes) -> DataEntry:
    info = game_info(game)
    if len(data) != info["record_bytes"]:
        raise ValueError(f"invalid training data record: {len(data)} != {info['record_bytes']}")
    nplanes, words, moves = info["planes"], info["plane_words"], info["moves"]
    off = nplanes * words * 8
    planes = np.frombuffer(data, dtype="<u8", count=nplanes * words).reshape(nplanes, words).copy()
    if game == "chess":
        bitmap = np.frombuffer(data, dtype=np.uint8, count=235, offset=off)
        packed = np.frombuffer(data, dtype="<f4", count=225, offset=off + 235)
        probs = np.full((moves,), -1.0, dtype=np.float32)
        idx = np.where(np.unpackbits(bitmap, count=moves, bitorder="little"))[0]
        probs[idx] = packed[: len(idx)]
        off += 235 + 225 * 4
    else:
        probs = np.frombuffer(data, dtype="<f4", count=moves, offset=off).copy()
        off += moves * 4
    winner = float(np.frombuffer(data, dtype=np.int8, count=1, offset=off)[0])
    return DataEntry(planes=planes, probs=probs, winner=winner)


def unpack_planes(entry: DataEntry, game: str) -> np.ndarray:
    """uint8 tensor [planes, S, S] with bit h*S+w of each plane (data_set.py:65-73)."""
    info = game_info(game)
    s = info["board"]
    bits = np.unpackbits(entry.planes.view(np.uint8).reshape(info["planes"], -1), axis=1, bitorder="little")
    return bits[:, : s * s].reshape(info["planes"], s, s)
