#!/usr/bin/env python3
"""Timing variants of conv3x3_wino4_kernel, made by text substitution on a COPY of cattus_amd/csrc/kernels_wino4.hip (the product
source carries no experiment switch).  The results of every variant are wrong; only its launch time means anything.

    python scripts/probes/w4_variant.py [--file OTHER.hip] NAME [NAME ...]     ->  cattus_amd/libcattus_hip_w4_NAME.so   (CATTUS_HIP_LIB selects it)
(NAME "asis" = no substitution: another version of the source, given with --file, beside the tree's)

Variants: notransform (no slices between the MFMAs), noreads (slices without their LDS reads), noring (the U ring is never
refilled nor waited for), nodma (no activation DMA in the loop), noepi (no exchange, no stores), plainstore (plain instead of
non-temporal stores: results stay right), stamps (-DCATTUS_STAMPS on top), and combinations joined with '+'."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
SRC = ROOT / "cattus_amd" / "csrc" / "kernels_wino4.hip"
OUT = Path("/tmp/variants")

SKEW_ANCHOR = """                asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory");
#pragma unroll
                for (int X = 0; X < 2; X++)"""
EDITS = {
    "notransform": [("                slot(ph == 1 ? SP_ : SP_ ^ 1, ph == 2 ? 0 : 1, j);\n", "")],
    "noreads": [("            if (jj == 2) read_row(pa, 0, ntb, nsp, ng);\n            else read_row(pb, 1, ntb, nsp, ng);\n", "")],
    "noring": [("            if (!first_use) load_ustage(ring[l], wnext, l);", ""),
               ("            if (first_use) {", "            if (false) {")],
    "nomfma": [("            Mfma<T>::mac(ul0, vh, acc[l][tbv][0]);\n", ""), ("            Mfma<T>::mac(ul1, vh, acc[l][tbv][1]);\n", ""),
               ("            Mfma<T>::mac(uh0, vl, acc[l][tbv][0]);\n", ""), ("            Mfma<T>::mac(uh1, vl, acc[l][tbv][1]);\n", ""),
               ("            Mfma<T>::mac(uh0, vh, acc[l][tbv][0]);\n", ""), ("            Mfma<T>::mac(uh1, vh, acc[l][tbv][1]);\n", "")],
    # behind the chunk barrier wave q waits q x N cycles: the four waves' load bursts (4 x 16 cycles of the CU's address unit each) then go
    # out one after the other instead of at the same moment (results stay right)
    "skew64": [(SKEW_ANCHOR, SKEW_ANCHOR.replace("#pragma unroll", "if (q >= 1) __builtin_amdgcn_s_sleep(1);\n                if (q >= 2) __builtin_amdgcn_s_sleep(1);\n                if (q >= 3) __builtin_amdgcn_s_sleep(1);\n#pragma unroll", 1))],
    "skew32": [(SKEW_ANCHOR, SKEW_ANCHOR.replace("#pragma unroll", "if (q >= 1) asm volatile(\"s_nop 7\\n\\ts_nop 7\\n\\ts_nop 7\\n\\ts_nop 7\");\n                if (q >= 2) asm volatile(\"s_nop 7\\n\\ts_nop 7\\n\\ts_nop 7\\n\\ts_nop 7\");\n                if (q >= 3) asm volatile(\"s_nop 7\\n\\ts_nop 7\\n\\ts_nop 7\\n\\ts_nop 7\");\n#pragma unroll", 1))],
    "skew128": [(SKEW_ANCHOR, SKEW_ANCHOR.replace("#pragma unroll", "if (q >= 1) __builtin_amdgcn_s_sleep(2);\n                if (q >= 2) __builtin_amdgcn_s_sleep(2);\n                if (q >= 3) __builtin_amdgcn_s_sleep(2);\n#pragma unroll", 1))],
    # an even stage's loads in its light gaps (0: the group's wait, 1 and 3: nothing of the transform) instead of gaps 2-5 (results stay right)
    "lightgaps": [("                const int le = (i & 1) ? e : e - 2;\n", "                const int le = (i & 1) ? e : e == 0 ? 0 : e == 1 ? 1 : e == 3 ? 2 : e == 5 ? 3 : -1;\n")],
    "lightgaps2": [("                const int le = (i & 1) ? e : e - 2;\n", "                const int le = (i & 1) ? e : e == 1 ? 0 : e == 3 ? 1 : e == 4 ? 2 : e == 5 ? 3 : -1;\n")],
    "sleep1": [("                __builtin_amdgcn_s_sleep(8);\n", "                __builtin_amdgcn_s_sleep(1);\n")],
    "sleep0": [("                __builtin_amdgcn_s_sleep(8);\n", "")],
    "nodma": [("                issue_chunk(ch_next, freed);\n", "")],
    "noepi": [("    asm volatile(\"s_barrier\" ::: \"memory\");  // every wave has left the chunk buffers", "    if (cin > 0) return;\n    asm volatile(\"s_barrier\" ::: \"memory\");  // every wave has left the chunk buffers")],
    # the one-launch tower with one of its sc1 / asm pieces replaced by the per-layer kernel's plain form (results may stay right)
    "p_plainstore": [("        if (PERSIST) asm volatile(\"global_store_dwordx4 %0, %1, off sc1\\n\\ts_nop 1\" ::\"v\"(op), \"v\"(v) : \"memory\");\n        else *reinterpret_cast<f32x4*>(op) = v;",
                      "        *reinterpret_cast<f32x4*>(op) = v;")],
    "p_plainskip": [("            if (PERSIST) asm volatile(\"global_load_dwordx4 %0, %1, off sc1\" : \"=&v\"(skip[k]) : \"v\"(sp) : \"memory\");\n            else skip[k] = *reinterpret_cast<const f32x4*>(sp);",
                     "            skip[k] = *reinterpret_cast<const f32x4*>(sp);")],
    "p_plainload": [("            if (PERSIST) asm volatile(\"global_load_dwordx4 %0, %1, %2 sc1\"", "            if (false) asm volatile(\"global_load_dwordx4 %0, %1, %2 sc1\"")],
    "plainstore": [("        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + (orow + k) * (size_t)cout + cout0 + pc * 4));",
                    "        *reinterpret_cast<f32x4*>(out + (orow + k) * (size_t)cout + cout0 + pc * 4) = v;")],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-mllvm", "-amdgpu-kernarg-preload-count=16",
         f"-I{ROOT / 'include'}", f"-I{SRC.parent}"]


def main():
    subprocess.check_call([sys.executable, "-c", "from cattus_amd import build; build.build_hip()"], cwd=ROOT)
    OUT.mkdir(parents=True, exist_ok=True)
    args = sys.argv[1:]
    base = SRC
    if args and args[0] == "--file":  # another version of the source as the base (an A/B of two versions on one box)
        base, args = Path(args[1]), args[2:]
    for name in args:
        text = base.read_text()
        extra = []
        for part in name.split("+"):
            if part == "asis":
                continue
            if part == "stamps":  # the diagnostic build's cycle stamps (scripts/stamps_w4.py) on top of the variant
                extra.append("-DCATTUS_STAMPS")
                continue
            for old, new in EDITS[part]:
                assert text.count(old) == 1, (part, old[:60], text.count(old))
                text = text.replace(old, new)
        src = OUT / f"kernels_wino4_{name.replace('+', '_')}.hip"
        src.write_text(text)
        obj = src.with_suffix(".o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-c", str(src), "-o", str(obj)])
        objs = [str(obj) if s == "kernels_wino4" else str(ROOT / "cattus_amd" / "build" / "hip" / f"{s}.o")
                for s in ("kernels", "kernels_t64s", "kernels_wino", "kernels_wino4", "kernels_wino8", "evaluator")]
        lib = ROOT / "cattus_amd" / f"libcattus_hip_w4_{name.replace('+', '_')}.so"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-o", str(lib), *objs, "-lpthread", "-ldl"])
        print("built", lib)


if __name__ == "__main__":
    main()
