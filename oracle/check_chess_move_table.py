#!/usr/bin/env python3
"""Build-container check: the generated chess policy-index table (cattus_amd/csrc/host/chess.h)
equals the table the reference lists in engine/src/chess/core.rs:453-593.

The reference text is only READ here to compare; what gets committed is the sha256 of the 1880
LAN strings joined by ',' (tests/golden/chess_nn_moves.sha256), which tests check on any box.
"""

import hashlib
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from cattus_amd import selfplay  # noqa: E402

src = Path("/root/reference/engine/src/chess/core.rs").read_text()
start = src.index("static NN_INDEX_TO_MOVE")
end = src.index(".into_iter()", start)
ref = re.findall(r'"([a-h][1-8][a-h][1-8][qrbn]?)"', src[start:end])
ours = selfplay.chess_nn_moves()
assert len(ref) == 1880, len(ref)
assert ours == ref, [(i, a, b) for i, (a, b) in enumerate(zip(ours, ref)) if a != b][:10]
digest = hashlib.sha256(",".join(ours).encode()).hexdigest()
(ROOT / "tests" / "golden" / "chess_nn_moves.sha256").write_text(digest + "\n")
print("chess policy-index table matches the reference; sha256", digest)
