"""The walker kernels' node-phase loops must come out of the compiler in the form in which every lane stays in
the loop until the common exit (tools/check_walker_listing.py explains the other form and what it breaks).
CPU only: hipcc cross-compiles render.hip to a listing."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_walker_listing as cwl  # noqa: E402

GOOD = """
_Z10k_traverseILb0E3FooEv:
.LBB0_1:
\ts_andn2_b64 exec, exec, s[14:15]
\ts_cbranch_execz .LBB0_9
.LBB0_2:                                ;   Parent Loop BB0_0 Depth=1
                                        ; =>  This Inner Loop Header: Depth=2
\tv_cmp_ne_u32_e64 s[4:5], -1, v20
\ts_bcnt1_i32_b64 s6, s[6:7]
\ts_bcnt1_i32_b64 s34, s[4:5]
\ts_branch .LBB0_1
.LBB0_9:
.Lfunc_end0:
"""
BAD = GOOD.replace("\ts_branch .LBB0_1\n", "\ts_andn2_b64 exec, exec, s[96:97]\n\ts_cbranch_execnz .LBB0_2\n\ts_branch .LBB0_1\n")


def test_checker_tells_the_two_forms_apart():
    assert cwl.check_listing(GOOD) == {"_Z10k_traverseILb0E3FooEv": (1, [])}
    assert cwl.check_listing(BAD) == {"_Z10k_traverseILb0E3FooEv": (1, [".LBB0_2"])}


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("unit", range(len(cwl.UNITS)))
def test_node_phase_loops_keep_all_lanes_to_the_common_exit(unit):
    source, extra = cwl.UNITS[unit]
    report = cwl.check_listing(cwl.compile_listing(source, extra))
    assert len(report) >= 8, sorted(report)                        # COUNT on/off x the ray sources, and the chain of later bounces
    for name, (loops, bad) in report.items():
        # eight octants + the mixed case, twice: one level of the tree per step, and two, front to back; a third time -- two
        # levels, boxes as they come -- where a launch may enter every box (camera, caller and bounce rays, kernels without
        # counters; their front-to-back phase is the one that leaves boxes out with a bound)
        # (k_traverse_chain walks the later bounces of a few paths: the same walker as the bounce launches it stands for)
        every_box = ("ILb0E" in name and ("QueueSrc" in name or "ArraySrc" in name)) or "k_traverse_chain" in name
        assert loops == (27 if every_box else 18), (name, loops)
        assert not bad, (name, bad)
