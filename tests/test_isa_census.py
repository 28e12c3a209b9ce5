"""bench.py's roofline.frac is derived from a disassembly of the shipped libswmi.so (tools/isa_census.py): these checks pin
what that derivation rests on -- the tool finds the kernels, the hot loop of the headline kernel has the instruction mix
DESIGN.md section 5 describes, and the fraction it yields is a utilisation (<= 1).  No GPU needed: hipcc cross-compiles."""
import os
import shutil
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.skipif(not (os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or shutil.which("llvm-objdump")),
                                reason="llvm-objdump not available")


def test_headline_kernel_census():
    import isa_census
    found = isa_census.census_for(r"^sw128_kernel<4,1,0,0>$", marker_op="v_dot4_i32_i8")
    assert list(found) == ["sw128_kernel<4,1,0,0>"]
    c = found["sw128_kernel<4,1,0,0>"]
    ops = c["main_loop"]["by_op"]
    # two anti-diagonal steps x 32 rows per lane: one v_dot4 (lookup + diagonal add) and one saturating subtract per cell
    assert ops["v_dot4_i32_i8"] == 64 and ops["v_sub_u32"] == 64
    assert 64 <= ops["v_max3_i32"] <= 96                 # one per cell + the part of the running maximum kept on the VALU
    assert c["main_loop"]["by_class"].get("vmem", 0) == 0 and c["main_loop"]["unmeasured_valu"] == 0
    assert c["main_loop_conditional"]["instructions"] == 0
    assert len(c["code_sha256"]) == 16


def test_every_schedule_and_row_has_a_census():
    import isa_census
    names = set(isa_census.readable(s) for s in isa_census.disassemble())
    for lanes in (64, 32, 16, 8, 4, 2):
        for mode in (0, 1, 2):
            assert "sw128_kernel<%d,1,0,%d>" % (lanes, mode) in names
    for k in ("sw_banded_affine_kernel<1,1>", "sw_banded_affine_kernel<1,0>", "sw_banded_affine_kernel<0,0>",
              "sg_forward_split_kernel<2,2>", "sg_forward_split_kernel<4,1>", "sg_forward_kernel<8>",
              "sg_walk_lane_kernel", "sg_expand_kernel", "sg_traceback_kernel"):
        assert k in names, k


def test_issue_bound_is_a_utilisation():
    import bench
    # the headline launch: 1,048,576 pairs = 65,536 wavefronts, 66 loop trips; at the measured 1.4975 ms the kernel sits at
    # ~0.85 of the VALU issue bound; no kernel time can push the fraction above 1 without being faster than the bound
    r = bench.issue_bound(r"^sw128_kernel<4,1,0,0>$", 66, 65536, 1.4975, marker=("v_dot4_i32_i8", 64))
    assert 0.8 < r["frac"] < 0.9 and r["frac"] <= r["frac_at_measured_instruction_rates"] <= 1.0
    assert r["census"]["valu_instructions_per_wavefront"] == 14247       # x 65,536 = 933.7 M: SQ_INSTS_VALU reads 933.9 M
    ideal_ms = r["frac"] * 1.4975
    assert bench.issue_bound(r"^sw128_kernel<4,1,0,0>$", 66, 65536, ideal_ms, marker=("v_dot4_i32_i8", 64))["frac"] == pytest.approx(1.0, abs=2e-3)
    # unrolled instantiations: the marker count says by how much (L = 64 is unrolled by four)
    r64 = bench.issue_bound(r"^sw128_kernel<64,1,0,0>$", 96, 1 << 20, 2.66, marker=("v_dot4_i32_i8", 4))
    assert r64["census"]["main_loop_trips"] == 24 and 0.85 < r64["frac"] < 1.0
    # the semi-global sweep: blocks under a scalar condition are tallied apart and weighted by how often they run
    sg = bench.issue_bound(r"^sg_forward_split_kernel<2,2>$", 32768, 2048, 31.0, marker=("v_alignbit_b32", 19), conditional_share=1 / 16)
    assert sg["census"]["main_loop_conditional_issue_cycles"] > 0 and 0.4 < sg["frac"] < 1.0
