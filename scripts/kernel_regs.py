#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in libcattus_hip.so's gfx950 code object (no GPU needed).

    python scripts/kernel_regs.py [substring ...]     # e.g. splitw  tower64  _Float16
"""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(so: bytes):
    """Every gfx950 code object of the library: one offload bundle per translation unit."""
    i = so.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while i >= 0:
        n = struct.unpack_from("<Q", so, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", so, off)
            off += 24
            triple = so[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz:
                yield so[i + o:i + o + sz]
        i = so.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)


def main():
    lib = ROOT / "cattus_amd" / "libcattus_hip.so"
    notes = ""
    for co in code_objects(lib.read_bytes()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes += subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    rows = []
    for k in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", k).group(1))  # noqa: E731
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        rows.append((name, get("vgpr_count"), int(k.split("\n")[0].strip()), get("sgpr_count"), get("private_segment_fixed_size"),
                     get("vgpr_spill_count"), get("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("vgpr agpr sgpr scratch spills static_lds  kernel")
    for (name, v, a, s, sc, sp, lds), d in zip(rows, names):
        d = re.sub(r"cattus::|\(.*", "", d)
        if sys.argv[1:] and not any(x in d for x in sys.argv[1:]):
            continue
        print(f"{v:4d} {a:4d} {s:4d} {sc:7d} {sp:6d} {lds:10d}  {d}")


if __name__ == "__main__":
    main()
