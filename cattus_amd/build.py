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
HIP_SOURCES = [CSRC / "kernels.hip", CSRC / "kernels_t64s.hip", CSRC / "kernels_wino.hip", CSRC / "kernels_wino4.hip", CSRC / "kernels_wino8.hip", CSRC / "evaluator.hip"]
HIP_DEPS = HIP_SOURCES + [CSRC / "kernels.h", CSRC / "device_common.h", ROOT / "include" / "cattus_hip.h"]

# -ffp-contract=off: the f32 path promises a fixed fmaf-chain order, so the compiler
# must not fuse or split any multiply-add on its own, on the device or in the host-side BN folding.
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
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


def _compile_and_link(lib: Path, extra_flags, verbose: bool, obj_dir: Path, force: bool) -> None:
    """One object per translation unit, compiled side by side (kernels.hip alone takes two minutes), then one link.
    An object is rebuilt when its source or any shared header is newer."""
    from concurrent.futures import ThreadPoolExecutor

    obj_dir.mkdir(parents=True, exist_ok=True)
    headers = [d for d in HIP_DEPS if d not in HIP_SOURCES]
    jobs = []
    for src in HIP_SOURCES:
        obj = obj_dir / (src.stem + ".o")
        if force or _stale(obj, [src, *headers]):
            jobs.append([_hipcc(), *HIPCC_FLAGS, *extra_flags, "-c", str(src), "-o", str(obj)])
    if verbose:
        for cmd in jobs:
            print(" ".join(cmd))
    with ThreadPoolExecutor(max_workers=max(1, len(jobs))) as pool:
        for rc in pool.map(subprocess.call, jobs):
            if rc != 0:
                raise subprocess.CalledProcessError(rc, "hipcc")
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", str(lib),
            *[str(obj_dir / (s.stem + ".o")) for s in HIP_SOURCES], "-lpthread", "-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)


def build_hip(force: bool = False, verbose: bool = False) -> Path:
    """Compile cattus_amd/libcattus_hip.so for gfx950."""
    if force or _stale(HIP_LIB, HIP_DEPS):
        _compile_and_link(HIP_LIB, [], verbose, PKG / "build" / "hip", force)
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


POOL_LIB = PKG / "libcattus_pool.so"
POOL_SOURCES = [CSRC / "pool.cpp"]
POOL_DEPS = POOL_SOURCES + [ROOT / "include" / "cattus_pool.h"]


def build_pool(force: bool = False, verbose: bool = False) -> Path:
    """Compile cattus_amd/libcattus_pool.so (record pooling over RCCL for non-Python hosts; host code, links librccl)."""
    if force or _stale(POOL_LIB, POOL_DEPS):
        rocm = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
        cmd = [_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-x", "hip", "--offload-arch=gfx950",
               "-o", str(POOL_LIB), *map(str, POOL_SOURCES), f"-L{rocm / 'lib'}", "-lrccl", f"-Wl,-rpath,{rocm / 'lib'}"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return POOL_LIB


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    build_host(force, verbose)
    build_pool(force, verbose)


def build_diag(verbose: bool = False) -> None:
    """Diagnostic builds of the same ABIs (never quote their run times):
    libcattus_hip_diag.so with in-kernel cycle stamps (scripts/stamps.py) and libcattus_selfplay_diag.so
    with the scheduler's worker-time split (select them with CATTUS_HIP_LIB / CATTUS_SELFPLAY_LIB)."""
    _compile_and_link(PKG / "libcattus_hip_diag.so", ["-DCATTUS_STAMPS"], verbose, PKG / "build" / "hip_diag", True)
    cmds = [
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
