"""Race detection for the host scheduler: the self-play driver (worker threads + evaluation thread +
sharded cache) built with -fsanitize=thread and run on hex4 and chess with the stand-in network.
The reference has no sanitizer coverage; its Batcher is argued correct by comments."""

import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_selfplay_scheduler_is_race_free_under_tsan(tmp_path):
    exe = tmp_path / "tsan_selfplay"
    subprocess.check_call(
        ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", str(exe),
         str(ROOT / "tests" / "native" / "tsan_selfplay.cpp"), str(ROOT / "cattus_amd" / "csrc" / "host" / "cabi.cpp")],
        cwd=str(ROOT / "tests" / "native"),
    )
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    assert p.stdout.count("rc=0") == 3
