#!/usr/bin/env python3
"""Audit of the generated gfx950 code for the one thing hipcc cannot know about hand-placed asm loads: a VGPR that an asm
`global_load` is still writing must not be read, copied or overwritten by any instruction before the `s_waitcnt vmcnt(N)` that
covers it (the compiler treats the register as written when the asm statement ends; a `v_mov` of it at a control-flow merge copies
garbage whenever the load is slow -- wrong results on cold caches only; cdna_hip_programming.md section 5.7, item 1).

    python scripts/audit_inflight_regs.py [--no-cache] [file.hip | file.s ...]     # default: cattus_amd/csrc/kernels_t64s.hip

The scan follows CONTROL FLOW, not the text: a kernel is cut into basic blocks (at labels and behind every branch), the state --
the FIFO of loads in flight with the registers each one writes -- is carried along every edge (`s_branch` to its target,
`s_cbranch_*` to its target and to the fall-through, `s_endpgm` nowhere) and every block is re-walked for every distinct state
that reaches it, until no new state appears (a ring that is refilled in a loop cycles through finitely many states).  Inside a
block: asm loads (between ;;#ASMSTART / ;;#ASMEND) enter the FIFO with their destination registers, LDS-DMA enters it with none,
every `s_waitcnt vmcnt(N)` (asm or compiler) retires all but the N youngest, and any other instruction naming an in-flight
register is reported once.  (Round 4's scan read the blocks in textual order; when the compiler laid a loop's second half ahead of
its header it saw a refill's registers "in flight" at instructions that execute before the refill -- seven false alarms.)

A `.s` file is audited as it stands (unit tests plant a hazard in a canned listing).  Results are cached by the hash of the source,
the shared headers and this script under cattus_amd/build/audit/ (kernels.hip takes two minutes to compile).
Exit status 1 if anything is found.  No GPU needed (hipcc cross-compiles)."""
import hashlib
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         "--offload-device-only", "-S", "-Wno-unused-command-line-argument"]
MAX_STATES_PER_BLOCK = 4096  # a runaway (an unbounded FIFO: a loop that loads and never waits) is reported, not looped on


def regs(tok: str) -> frozenset:
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


class Block:
    __slots__ = ("label", "ops", "succ", "falls")

    def __init__(self, label):
        self.label, self.ops, self.succ, self.falls = label, [], [], True  # ops: (line number, kind, payload, text)


def split_blocks(lines: list) -> list:
    """Basic blocks of one kernel's listing, in textual order; `falls` says whether control can run into the next block."""
    blocks, cur, in_asm = [Block(None)], None, False
    cur = blocks[0]
    for no, raw in enumerate(lines):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", t)
        if m and not in_asm:
            if cur.ops or cur.label is not None:
                cur = Block(m.group(1))
                blocks.append(cur)
            else:
                cur.label = m.group(1)
            continue
        t = t.split(";")[0].strip() if not t.startswith(";") else ""
        if not t or t.startswith("."):
            continue
        op = t.split()[0]
        if op == "s_endpgm":
            cur.ops.append((no, "end", None, t))
            cur.falls = False
            cur = Block(None)
            blocks.append(cur)
        elif op == "s_branch":
            cur.succ.append(t.split()[1])
            cur.falls = False
            cur = Block(None)
            blocks.append(cur)
        elif op.startswith("s_cbranch"):
            cur.succ.append(t.split()[-1])
            cur = Block(None)
            blocks.append(cur)
        elif in_asm and op.startswith("global_load_lds"):
            cur.ops.append((no, "load", frozenset(), t))  # LDS-DMA: counted by vmcnt like a load, but it writes no register
        elif in_asm and op.startswith("global_load_"):
            cur.ops.append((no, "load", regs(t.split()[1].rstrip(",")), t))
        elif op == "s_waitcnt" and "vmcnt(" in t:
            cur.ops.append((no, "wait", int(re.search(r"vmcnt\((\d+)\)", t).group(1)), t))
        else:
            touched = frozenset().union(*[regs(tok) for tok in re.findall(r"v\[\d+:\d+\]|\bv\d+\b", t)] or [frozenset()])
            if touched:
                cur.ops.append((no, "use", touched, t))
    return blocks


def audit_kernel(name: str, lines: list) -> list:
    blocks = split_blocks(lines)
    by_label = {b.label: i for i, b in enumerate(blocks) if b.label is not None}
    seen = [set() for _ in blocks]
    found, work = {}, [(0, ())]
    while work:
        bi, state = work.pop()
        if state in seen[bi]:
            continue
        if len(seen[bi]) >= MAX_STATES_PER_BLOCK:
            found[(name, -1)] = f"{name}: block {blocks[bi].label}: more than {MAX_STATES_PER_BLOCK} distinct states (loads that are never waited for?)"
            continue
        seen[bi].add(state)
        fifo = list(state)
        b = blocks[bi]
        for no, kind, payload, text in b.ops:
            if kind == "load":
                fifo.append(payload)
            elif kind == "wait":
                if payload < len(fifo):
                    fifo = fifo[len(fifo) - payload:] if payload else []
            elif kind == "use" and fifo:
                live = frozenset().union(*fifo)
                if payload & live:
                    found.setdefault((name, no), f"{name}: line {no}: `{text}` touches in-flight v{sorted(payload & live)}")
        if not b.ops or b.ops[-1][1] != "end":
            out = tuple(fifo)
            for lab in b.succ:
                if lab in by_label:
                    work.append((by_label[lab], out))
            if b.falls and bi + 1 < len(blocks):
                work.append((bi + 1, out))
    return [found[k] for k in sorted(found)]


def audit_text(text: str):
    """(number of kernels with asm statements, findings) of a whole listing."""
    kernels = re.split(r"\n(?=_Z\w+:\s+; @)", text)
    n, bad = 0, []
    for k in kernels:
        m = re.match(r"(_Z\w+):", k)
        if not m or ";;#ASMSTART" not in k:
            continue
        n += 1
        body = k.split("\n")
        end = next((i for i, line in enumerate(body) if line.strip().startswith(".Lfunc_end")), len(body))
        bad += audit_kernel(m.group(1), body[1:end])
    return n, bad


def source_key(src: Path) -> str:
    h = hashlib.sha256()
    for p in [src, *sorted(src.parent.glob("*.h")), Path(__file__)]:
        h.update(p.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:24]


def audit_file(src: Path, use_cache: bool = True):
    if src.suffix == ".s":
        return audit_text(src.read_text())
    cache = ROOT / "cattus_amd" / "build" / "audit" / f"{src.stem}.{source_key(src)}.json"
    if use_cache and cache.exists():
        rec = json.loads(cache.read_text())
        return rec["kernels"], rec["findings"]
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, f"-I{src.parent}", str(src), "-o", str(out)])
        n, bad = audit_text(out.read_text())
    cache.parent.mkdir(parents=True, exist_ok=True)
    for old in cache.parent.glob(f"{src.stem}.*.json"):
        old.unlink()
    cache.write_text(json.dumps({"kernels": n, "findings": bad}))
    return n, bad


def main():
    args = [a for a in sys.argv[1:] if a != "--no-cache"]
    files = [Path(a) for a in args] or [ROOT / "cattus_amd" / "csrc" / "kernels_t64s.hip"]
    bad = []
    for src in files:
        n, b = audit_file(src, use_cache="--no-cache" not in sys.argv[1:])
        bad += b
        print(f"{src.name}: {n} kernels with asm statements audited", file=sys.stderr)
    for b in bad:
        print(b)
    print(f"{len(bad)} uses of in-flight registers")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
