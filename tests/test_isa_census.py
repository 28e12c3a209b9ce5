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
    # what the default schedule runs for a large batch with the harness parameters (10, -30, 15): the packed kernel in its
    # bias form (sw128_pk_kernel<mode 0, BIAS 1>)
    found = isa_census.census_for(r"^sw128_pk_kernel<0,1>$", marker_op="v_perm_b32")
    assert list(found) == ["sw128_pk_kernel<0,1>"]
    c = found["sw128_pk_kernel<0,1>"]
    ops = c["main_loop"]["by_op"]
    # two anti-diagonal steps x 32 rows, two alignments per register: per PAIR of cells one v_perm (lookup), two full-rate
    # 32-bit adds (diagonal term, H + Q), one saturating subtraction, one three-input max + half of one for the running best
    assert ops["v_perm_b32"] == 64 and ops["v_pk_sub_u16"] == 64 and ops["v_pk_maximum3_f16"] == 96
    assert 128 <= ops["v_add_u32"] <= 132 and "v_pk_add_u16" not in ops
    assert c["main_loop"]["by_class"].get("vmem", 0) == 0 and c["main_loop"]["unmeasured_valu"] == 0
    assert c["main_loop_conditional"]["instructions"] == 0
    assert len(c["code_sha256"]) == 16
    # without the bias (every folded score >= 0, e.g. (1,-1,1)): one add per pair fewer
    ops0 = isa_census.census_for(r"^sw128_pk_kernel<0,0>$", marker_op="v_perm_b32")["sw128_pk_kernel<0,0>"]["main_loop"]["by_op"]
    assert ops0["v_pk_sub_u16"] == 64 and ops0["v_pk_maximum3_f16"] == 96 and 64 <= ops0["v_add_u32"] <= 68
    # the int32 kernel (schedule flag 8, and every L other than 4)
    ci = isa_census.census_for(r"^sw128_kernel<4,1,0,0>$", marker_op="v_dot4_i32_i8")["sw128_kernel<4,1,0,0>"]["main_loop"]["by_op"]
    assert ci["v_dot4_i32_i8"] == 64 and ci["v_sub_u32"] == 64 and 64 <= ci["v_max3_i32"] <= 96


def test_every_schedule_and_row_has_a_census():
    import isa_census
    names = set(isa_census.readable(s) for s in isa_census.disassemble())
    for lanes in (64, 32, 16, 8, 4, 2):
        for mode in (0, 1, 2):
            assert "sw128_kernel<%d,1,0,%d>" % (lanes, mode) in names
    for mode in (0, 1, 2):
        for bias in (0, 1):
            assert "sw128_pk_kernel<%d,%d>" % (mode, bias) in names
    for k in ("sw_banded_affine_kernel<1,1>", "sw_banded_affine_kernel<1,0>", "sw_banded_affine_kernel<0,0>",
              "sg_forward_split_kernel<2,2>", "sg_forward_split_kernel<4,1>", "sg_forward_kernel<8>",
              "sg_walk_lane_kernel", "sg_expand_kernel", "sg_traceback_kernel"):
        assert k in names, k


def test_issue_bound_is_a_utilisation():
    import bench
    # the int32 kernel's launch: 1,048,576 pairs = 65,536 wavefronts, 66 loop trips; at the measured 1.4975 ms it sits at
    # ~0.85 of the VALU issue bound; no kernel time can push the fraction above 1 without being faster than the bound
    r = bench.issue_bound(r"^sw128_kernel<4,1,0,0>$", 66, 65536, 1.4975, marker=("v_dot4_i32_i8", 64))
    assert 0.8 < r["frac"] < 0.9 and r["frac"] <= r["frac_at_measured_instruction_rates"] <= 1.0
    assert r["census"]["valu_instructions_per_wavefront"] == 14247       # x 65,536 = 933.7 M: SQ_INSTS_VALU read 933.9 M
    ideal_ms = r["frac"] * 1.4975
    assert bench.issue_bound(r"^sw128_kernel<4,1,0,0>$", 66, 65536, ideal_ms, marker=("v_dot4_i32_i8", 64))["frac"] == pytest.approx(1.0, abs=2e-3)
    # the packed kernel: 32 alignments per wavefront -> 32,768 wavefronts; 1.17 ms measured.  A third of its instructions
    # are full-rate adds, which only reach their 2-cycle rate when two wavefronts present one at the same time
    rp = bench.issue_bound(r"^sw128_pk_kernel<0,1>$", 65, 32768, 1.17, marker=("v_perm_b32", 64))
    assert 0.8 < rp["frac"] <= 1.0
    # unrolled instantiations: the marker count says by how much (L = 64 is unrolled by four)
    r64 = bench.issue_bound(r"^sw128_kernel<64,1,0,0>$", 96, 1 << 20, 2.66, marker=("v_dot4_i32_i8", 4))
    assert r64["census"]["main_loop_trips"] == 24 and 0.85 < r64["frac"] < 1.0
    # the semi-global sweep: blocks under a scalar condition are tallied apart and weighted by how often they run
    sg = bench.issue_bound(r"^sg_forward_split_kernel<2,2>$", 32768, 2048, 31.0, marker=("v_alignbit_b32", 19), conditional_share=1 / 16)
    assert sg["census"]["main_loop_conditional_issue_cycles"] > 0 and 0.4 < sg["frac"] < 1.0
