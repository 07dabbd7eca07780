"""The binding a Cattus maintainer adds (integration/rust: hip.rs + cattus_hip.patch) and the plain-C consumer that
replays its call sequence (integration/c/consumer.c).  The CPU half: files present, the C consumer is strict C99
against include/cattus_hip.h, the patch applies to the reference where a checkout exists (build container only)."""

import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")


def test_c_consumer_is_strict_c99(tmp_path):
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-D_POSIX_C_SOURCE=200809L", f"-I{ROOT / 'include'}", "-c",
                           str(ROOT / "integration" / "c" / "consumer.c"), "-o", str(tmp_path / "consumer.o")])


def test_rust_binding_declares_what_the_header_exports():
    """Every extern "C" function of hip.rs is a symbol of include/cattus_hip.h, and the config struct has its seven u32/i32 fields."""
    import re

    rs = (ROOT / "integration" / "rust" / "hip.rs").read_text()
    header = (ROOT / "include" / "cattus_hip.h").read_text()
    block = rs[rs.index('extern "C" {') : rs.index("}", rs.index('extern "C" {'))]
    names = re.findall(r"fn (cattus_hip_\w+)\(", block)
    assert len(names) >= 6
    for n in names:
        assert re.search(rf"\b{n}\(", header), n
    cfg = rs[rs.index("struct CattusEvalConfig {") :]
    cfg = cfg[: cfg.index("}")]
    assert re.findall(r"(\w+): [ui]32", cfg) == ["struct_size", "device", "max_batch", "plane_words", "dtype", "flush_us", "tower_form"]
    for variant, value in (("F32", 0), ("Bf16", 1), ("F16x2", 2)):
        assert f"{variant} = {value}" in rs


@pytest.mark.skipif(not REF.exists(), reason="the reference checkout exists in the build container only")
def test_patch_applies_to_the_reference(tmp_path):
    patch = ROOT / "integration" / "rust" / "cattus_hip.patch"
    touched = [line[6:].strip() for line in patch.read_text().splitlines() if line.startswith("--- a/")]
    assert {"engine/src/net/mod.rs", "engine/src/net/model.rs", "engine/Cargo.toml", "engine/build.rs"} <= set(touched)
    for rel in touched:
        (tmp_path / rel).parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(REF / rel, tmp_path / rel)
    out = subprocess.run(["patch", "-p1", "--dry-run", "-i", str(patch)], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    # and regenerating it from the reference gives the committed file (the generator is deterministic)
    subprocess.check_call(["patch", "-p1", "-s", "-i", str(patch)], cwd=tmp_path)
    assert "PLANE_WORDS" in (tmp_path / "engine/src/game/mod.rs").read_text()
    assert 'InferenceConfig::Hip { device, dtype }' in (tmp_path / "engine/src/net/mod.rs").read_text()


def test_no_compiler_instruction_touches_a_register_with_an_asm_load_in_flight():
    """The split towers keep their weight rings in registers filled by hand-placed asm loads (hipcc does not count them).
    The generated gfx950 code of EVERY kernel file is audited for any instruction that reads, copies or overwrites such a register
    before the `s_waitcnt vmcnt` that covers its load: a `v_mov` at a control-flow merge did exactly that once, and the results
    were wrong on cold caches only.  scripts/audit_inflight_regs.py follows control flow (tests/test_audit_inflight.py) and caches its
    verdict by source hash: kernels.hip compiles for two minutes the first time after a change (`__graft_entry__.build()` and
    scripts/precommit.sh warm the cache), the other two take seconds."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    csrc = root / "cattus_amd" / "csrc"
    files = sorted(csrc.glob("kernels*.hip"))
    assert {"kernels.hip", "kernels_t64s.hip", "kernels_wino.hip"} <= {f.name for f in files}
    p = subprocess.run([sys.executable, str(root / "scripts" / "audit_inflight_regs.py"), *map(str, files)], capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    import re

    for f in files:  # every file holds kernels with hand-placed loads, and every one of them was walked
        m = re.search(rf"^{re.escape(f.name)}: (\d+) kernels with asm statements audited", p.stderr, re.M)
        assert m and int(m.group(1)) >= 2, (f.name, p.stderr)
