"""scripts/audit_inflight_regs.py on canned gfx950 listings: the scan must follow control flow (a block laid out textually ahead
of the block that executes before it is not a hazard) and must still see a planted `v_mov` of a ring register at a merge."""

import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
spec = importlib.util.spec_from_file_location("audit_inflight_regs", ROOT / "scripts" / "audit_inflight_regs.py")
audit = importlib.util.module_from_spec(spec)
sys.modules["audit_inflight_regs"] = audit
spec.loader.exec_module(audit)

HEAD = "\t.text\n_Z4demoPf:                              ; @_Z4demoPf\n; %bb.0:\n"
TAIL = ".Lfunc_end0:\n\t.size\t_Z4demoPf, .Lfunc_end0-_Z4demoPf\n"


def listing(body: str) -> str:
    return HEAD + body + TAIL


def test_block_laid_out_ahead_of_its_predecessor_is_not_a_hazard():
    """The use of v14 sits textually behind the load and ahead of the wait, but executes behind the wait (bb.0 -> .LBB0_2 -> .LBB0_1):
    the round-4 scan, which read the blocks in textual order, reported it."""
    text = listing("""
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v1, s[2:3]
	;;#ASMEND
	s_branch .LBB0_2
.LBB0_1:
	v_add_f32_e32 v5, v14, v15
	s_endpgm
.LBB0_2:
	s_waitcnt vmcnt(0)
	s_branch .LBB0_1
""")
    n, bad = audit.audit_text(text)
    assert n == 1 and bad == []


def test_planted_copy_of_a_ring_register_at_a_merge_is_reported():
    text = listing("""
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v1, s[2:3]
	;;#ASMEND
	s_cbranch_scc1 .LBB0_2
; %bb.1:
	v_add_f32_e32 v2, v3, v4
	s_branch .LBB0_3
.LBB0_2:
	v_mul_f32_e32 v2, v3, v4
.LBB0_3:
	v_mov_b32_e32 v20, v10
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v5, v10, v11
	s_endpgm
""")
    n, bad = audit.audit_text(text)
    assert n == 1 and len(bad) == 1 and "v_mov_b32_e32 v20, v10" in bad[0] and "[10]" in bad[0]


def test_ring_refilled_in_a_loop_reaches_a_fixed_point_and_counts_per_path():
    """A two-slot ring refilled in a loop whose tail is laid out ahead of its header: clean; the same loop with the wait one short
    (vmcnt(2) leaves the slot about to be read in flight) is reported, once."""
    body = """
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v1, s[2:3]
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v1, s[2:3] offset:1024
	;;#ASMEND
	s_branch .LBB0_2
.LBB0_1:
	;;#ASMSTART
	s_waitcnt vmcnt(%d)
	;;#ASMEND
	v_add_f32_e32 v5, v14, v15
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v1, s[2:3] offset:1024
	;;#ASMEND
	s_add_i32 s4, s4, -1
	s_cmp_lg_u32 s4, 0
	s_cbranch_scc1 .LBB0_2
	s_branch .LBB0_3
.LBB0_2:
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_add_f32_e32 v6, v10, v11
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v1, s[2:3]
	;;#ASMEND
	s_branch .LBB0_1
.LBB0_3:
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v7, v10, v14
	s_endpgm
"""
    n, bad = audit.audit_text(listing(body % 1))
    assert n == 1 and bad == []
    n, bad = audit.audit_text(listing(body % 2))
    assert len(bad) == 1 and "v5, v14, v15" in bad[0]
