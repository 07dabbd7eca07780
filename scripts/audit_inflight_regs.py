#!/usr/bin/env python3
"""Audit of the generated gfx950 code for the one thing hipcc cannot know about hand-placed asm loads: a VGPR that an asm
`global_load` is still writing must not be read, copied or overwritten by any instruction before the `s_waitcnt vmcnt(N)` that
covers it (the compiler treats the register as written when the asm statement ends; a `v_mov` of it at a control-flow merge copies
garbage whenever the load is slow -- wrong results on cold caches only; cdna_hip_programming.md section 5.7, item 1).

    python scripts/audit_inflight_regs.py [file.hip ...]     # default: cattus_amd/csrc/kernels_t64s.hip

Linear scan per kernel: asm loads (between ;;#ASMSTART / ;;#ASMEND) enter a FIFO with their destination registers, every
`s_waitcnt vmcnt(N)` (asm or compiler) retires all but the N youngest, and any other instruction naming an in-flight register is
reported.  Exit status 1 if anything is found.  No GPU needed (hipcc cross-compiles)."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "--offload-device-only", "-S", "-Wno-unused-command-line-argument"]


def regs(tok: str) -> set:
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit_kernel(name: str, lines: list) -> list:
    inflight, in_asm, found = [], False, []
    for no, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if in_asm and t.startswith("global_load_lds"):
            inflight.append(set())  # LDS-DMA: counted by vmcnt like a load, but it writes no register
            continue
        if in_asm and t.startswith("global_load_"):
            inflight.append(regs(t.split()[1].rstrip(",")))
            continue
        m = re.search(r"vmcnt\((\d+)\)", t) if t.startswith("s_waitcnt") else None
        if m:
            n = int(m.group(1))
            inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
            if n == 0:
                inflight = []
            continue
        if not inflight:
            continue
        live = set().union(*inflight)
        touched = set()
        for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", t):
            touched |= regs(tok)
        if touched & live:
            found.append(f"{name}: line {no}: `{t}` touches in-flight v{sorted(touched & live)}")
    return found


def main():
    files = [Path(a) for a in sys.argv[1:]] or [ROOT / "cattus_amd" / "csrc" / "kernels_t64s.hip"]
    bad = []
    for src in files:
        with tempfile.TemporaryDirectory() as td:
            out = Path(td) / "k.s"
            subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, f"-I{src.parent}", str(src), "-o", str(out)])
            text = out.read_text()
        kernels = re.split(r"\n(?=_Z\w+:\s+; @)", text)
        n = 0
        for k in kernels:
            m = re.match(r"(_Z\w+):", k)
            if not m or ";;#ASMSTART" not in k:
                continue
            n += 1
            bad += audit_kernel(m.group(1), k.split("\n"))
        print(f"{src.name}: {n} kernels with asm statements audited", file=sys.stderr)
    for b in bad:
        print(b)
    print(f"{len(bad)} uses of in-flight registers")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
