"""In-tree native builds (hipcc cross-compiles gfx950 without a GPU present)."""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent

HIP_LIB = PKG / "libcattus_hip.so"
HIP_SOURCES = [CSRC / "kernels.hip", CSRC / "evaluator.hip"]
HIP_DEPS = HIP_SOURCES + [CSRC / "kernels.h", ROOT / "include" / "cattus_hip.h"]

# -ffp-contract=off: the f32 path promises a fixed fmaf-chain order, so the compiler
# must not fuse or split any multiply-add on its own, on the device or in the host-side BN folding.
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",
    "-fvisibility=hidden",
    # kernel arguments arrive in SGPRs with the wave instead of through a dependent scalar load: the conv
    # launches are short (17 us) and start cold, so the first-load latency matters (measured -0.5 us/launch)
    "-mllvm",
    "-amdgpu-kernarg-preload-count=16",
    "-Wall",
    "-Wno-unused-result",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP evaluator cannot be built")


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """Compile cattus_amd/libcattus_hip.so for gfx950."""
    if force or _stale(HIP_LIB, HIP_DEPS):
        cmd = [_hipcc(), *HIPCC_FLAGS, "-o", str(HIP_LIB), *map(str, HIP_SOURCES), "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


HOST_LIB = PKG / "libcattus_selfplay.so"
HOST_DIR = CSRC / "host"
HOST_SOURCES = [HOST_DIR / "cabi.cpp"]
HOST_DEPS = HOST_SOURCES + [HOST_DIR / n for n in ("games.h", "chess.h", "mcts.h", "selfplay.h")] + [
    ROOT / "include" / "cattus_selfplay.h"
]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fvisibility=hidden", "-Wall", "-Wextra",
              "-Wno-unused-parameter"]


def build_host(force: bool = False, verbose: bool = False) -> Path:
    """Compile cattus_amd/libcattus_selfplay.so (games, MCTS, self-play driver; plain C++, no GPU code)."""
    if force or _stale(HOST_LIB, HOST_DEPS):
        cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
        cmd = [cxx, *HOST_FLAGS, "-o", str(HOST_LIB), *map(str, HOST_SOURCES), "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)


def build_diag(verbose: bool = False) -> None:
    """Diagnostic builds of the same ABIs (never quote their run times):
    libcattus_hip_diag.so with in-kernel cycle stamps (scripts/stamps.py) and libcattus_selfplay_diag.so
    with the scheduler's worker-time split (select them with CATTUS_HIP_LIB / CATTUS_SELFPLAY_LIB)."""
    cmds = [
        [_hipcc(), *HIPCC_FLAGS, "-DCATTUS_STAMPS", "-o", str(PKG / "libcattus_hip_diag.so"), *map(str, HIP_SOURCES), "-lpthread", "-ldl"],
        [os.environ.get("CXX") or shutil.which("g++") or "g++", *HOST_FLAGS, "-DCATTUS_SCHED_STATS", "-o",
         str(PKG / "libcattus_selfplay_diag.so"), *map(str, HOST_SOURCES), "-lpthread"],
    ]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)


if __name__ == "__main__":
    import sys

    if "--diag" in sys.argv[1:]:
        build_diag(verbose=True)
    else:
        build_all(force=True, verbose=True)
