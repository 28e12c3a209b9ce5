"""Every shipped instantiation of the packed kernel sw128_pk_kernel<MODE, VARIANT, L> against REFERENCE fixtures, by name.

MODE 0 = pairs, 1 = 2-bit packed input, 2 = one-vs-many; VARIANT = the cell body the host picks from the parameters
(`expected_variant` below restates the rule of swmi_api.cpp make_config from the header's description, independently);
L = lanes per pair of alignments (4, 8, 16).  Each test asks the library which instantiation a launch runs
(swmi_score_kernel_for_batch) and requires it to be the one in the test id, so a change of the dispatch rule cannot
silently leave an instantiation untested.  Scores: F1 / F4 (the reference's scalar with its SIMD variants agreeing,
tests/golden/make_golden.py) for pairs and 2-bit input, F5's `scores_111x32` blocks (SmithWaterman_8b111x32mark1/2/3,
source.cpp:1227-1522) and `ovm_scores` (every seq1 of F5 against seq2[0], general parameters) for one-vs-many --
the shape of the reference's own TestSimdSmithWaterman / TestSimdSmithWaterman111x32 (source.cpp:2943-2982, :3003-3030).

Also here: the exhaustive self-test of the premise the packed kernel rests on (v_pk_maximum3_f16 = packed integer max3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = {"pairs": 0, "packed2bit": 1, "one_vs_many": 2}
VARIANTS = ("q0", "bias", "vert")      # template argument 0, 1, 2


def expected_variant(sm, gap):
    """The packed cell body for these parameters (include/swmi.h, "schedules"):
    q0    every sm + gap >= 0: no bias at all
    vert  some sm + gap < 0 but every sm + 2 gap in [0, 255] and the row offsets fit: vertical-offset form
    bias  everything else: values shifted by Q = -(min sm + gap)"""
    sm = np.asarray(sm, np.int32)
    gap = int(gap)
    if sm.min() + gap >= 0:
        return "q0"
    if sm.min() + 2 * gap >= 0 and sm.max() + 2 * gap <= 255 and 128 * max(0, int(sm.max())) + 34 * gap + 256 < 0x7C00:
        return "vert"
    return "bias"


def expected_name(mode, variant, lanes):
    v = VARIANTS.index(variant)
    return "sw128_pk_kernel<%d,%d>" % (mode, v) if lanes == 4 else "sw128_pk_kernel<%d,%d,%d>" % (mode, v, lanes)


def param_sets(f, variant, key_sm="sm", key_gap="gap"):
    return [p for p in range(f[key_sm].shape[0]) if expected_variant(f[key_sm][p], f[key_gap][p]) == variant]


@pytest.mark.parametrize("lanes", [4, 8, 16])
@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("mode", ["pairs", "packed2bit"])
def test_pairs_and_2bit_input_vs_reference_fixtures(gpu, oracle, golden, mode, variant, lanes):
    ran = 0
    gpu.set_schedule(lanes, 0)
    try:
        for name in ("f1_random", "f2_structured", "f4_param_sweep"):
            f = golden(name)
            a, b = f["seq1"], f["seq2"]
            pa, pb = (oracle.pack(a), oracle.pack(b)) if mode == "packed2bit" else (None, None)
            for p in param_sets(f, variant):
                sm, gap = f["sm"][p], int(f["gap"][p])
                got_name, per_wave = gpu.score_kernel_for_batch(a.shape[0], sm, gap, MODES[mode])
                assert got_name == expected_name(MODES[mode], variant, lanes), (got_name, sm, gap)
                assert per_wave == 128 // lanes
                got = gpu.score_batch_packed(pa, pb, sm, gap) if mode == "packed2bit" else gpu.score_batch(a, b, sm, gap)
                assert np.array_equal(got, f["scores"][p]), "%s parameter set %d (gap %d)" % (name, p, gap)
                ran += 1
    finally:
        gpu.set_schedule(0, 0)
    assert ran >= 3, "the fixtures hold too few parameter sets for the %s cell" % variant


@pytest.mark.parametrize("lanes", [4, 8, 16])
@pytest.mark.parametrize("variant", VARIANTS)
def test_one_vs_many_vs_reference_fixtures(gpu, golden, variant, lanes):
    f = golden("f5_siblings")
    ran = 0
    gpu.set_schedule(lanes, 0)
    try:
        if variant == "q0":           # SmithWaterman_8b111x32mark1/2/3: 32 seq1 x one seq2, (1, -1, 1)
            sm = np.full((4, 4), -1, np.int8)
            np.fill_diagonal(sm, 1)
            sm = sm.reshape(16)
            assert gpu.score_kernel_for_batch(32, sm, 1, 2)[0] == expected_name(2, "q0", lanes)
            for blk in range(f["scores_111x32"].shape[0]):
                got = gpu.score_one_vs_many(f["seq1"][32 * blk:32 * blk + 32], f["seq2"][blk], sm, 1)
                assert np.array_equal(got, f["scores_111x32"][blk]), blk
            ran += 1
        n = f["seq1"].shape[0]
        for p in param_sets(f, variant, "ovm_sm", "ovm_gap"):
            sm, gap = f["ovm_sm"][p], int(f["ovm_gap"][p])
            assert gpu.score_kernel_for_batch(n, sm, gap, 2)[0] == expected_name(2, variant, lanes), (sm, gap)
            got = gpu.score_one_vs_many(f["seq1"], f["seq2"][0], sm, gap)
            assert np.array_equal(got, f["ovm_scores"][p]), (p, gap)
            for m in (1, 2, 3, 31, 33, 63, 65, 255, 1023):        # ragged tails: half-filled registers and lane groups
                assert np.array_equal(gpu.score_one_vs_many(f["seq1"][:m], f["seq2"][0], sm, gap), f["ovm_scores"][p][:m]), (p, m)
            ran += 1
    finally:
        gpu.set_schedule(0, 0)
    assert ran >= 1, "F5 holds no one-vs-many parameter set for the %s cell" % variant


def test_automatic_schedule_reaches_the_one_vs_many_packed_kernels(gpu, oracle):
    """Without a forced schedule a one-vs-many call of bench size runs the packed kernel (row N1's bench row) -- 120 000
    sequences, (10, -30, 15), against the oracle."""
    n = 120000
    a, b = oracle.generate(n, 2718, 3)
    sm = np.full((4, 4), -30, np.int8)
    np.fill_diagonal(sm, 10)
    sm = sm.reshape(16)
    name, _ = gpu.score_kernel_for_batch(n, sm, 15, 2)
    assert name == expected_name(2, expected_variant(sm, 15), 4)
    want = oracle.batch(a, np.repeat(b[:1], n, axis=0), sm, 15)
    assert np.array_equal(gpu.score_one_vs_many(a, b[0], sm, 15), want)


def test_pk_maximum3_f16_is_a_packed_integer_max_exhaustive(gpu):
    """v_pk_maximum3_f16 on every pair of 16-bit integers in [0, 0x7C00), six operand arrangements each, inside a kernel
    that sets MODE.FP_DENORM the way the scoring kernels do (sw_kernels.hip keep_f16_denormals)."""
    checked, bad = gpu.selftest_pk_max3()
    assert checked == 6 * 0x7C00 * 0x7C00
    assert bad == 0
