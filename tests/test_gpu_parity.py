"""Parity of the HIP path (through the C ABI) with the golden fixtures of the real reference and with the
CPU oracle.  Integer work: the bar is bit-exact.  Every schedule (lanes per alignment) and both cell bodies
must agree, the way the reference's nine SIMD variants must agree with its scalar (TestSimdSmithWaterman,
source.cpp:2943-2982)."""
import numpy as np
import pytest

from conftest import match_matrix

pytestmark = pytest.mark.gpu

# (lanes per alignment, flags): flags 0 = gap-folded cell when the matrix allows it, 1 = general cell,
# 2 = 16-bit-max cell, 4 = LDS score-lookup kernel (falls back to the v_dot4 kernel where it does not apply),
# 8 = never the packed kernel (what L = 16, 8 and 4 run when no other flag is set: 8 / 16 / 32 alignments per wavefront)
SCHEDULES = [(64, 0), (64, 1), (32, 0), (32, 3), (16, 0), (16, 1), (16, 4), (16, 8), (8, 0), (8, 1), (8, 2), (8, 3), (8, 4), (8, 8),
             (4, 0), (4, 1), (4, 2), (4, 4), (4, 8), (2, 0), (2, 1)]


@pytest.mark.parametrize("lanes,flags", SCHEDULES)
@pytest.mark.parametrize("name", ["f1_random", "f2_structured", "f3_harness", "f4_param_sweep"])
def test_golden_fixtures_all_schedules(gpu, golden, name, lanes, flags):
    f = golden(name)
    gpu.set_schedule(lanes, flags)
    try:
        for p in range(f["sm"].shape[0]):
            got = gpu.score_batch(f["seq1"], f["seq2"], f["sm"][p], int(f["gap"][p]))
            assert np.array_equal(got, f["scores"][p]), "%s parameter set %d (gap %d)" % (name, p, f["gap"][p])
    finally:
        gpu.set_schedule(0, 0)


def test_reference_fuzzer_shape_vs_oracle(gpu, oracle):
    """TestSimdSmithWaterman (source.cpp:2943-2982): fresh random pairs, sm = 10/-30, gap 15, scalar vs SIMD."""
    a, b = oracle.generate(50000, 424242, 0)
    sm = match_matrix(10, -30)
    assert np.array_equal(gpu.score_batch(a, b, sm, 15), oracle.batch(a, b, sm, 15))


@pytest.mark.parametrize("seed", range(6))
def test_random_parameters_vs_oracle(gpu, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = 6000
    a = rng.integers(0, 4, (n, 128), dtype=np.uint8)
    b = rng.integers(0, 4, (n, 128), dtype=np.uint8)
    sim = rng.random(n) < 0.5                       # half the pairs related, with substitutions and a shift
    keep = rng.random((n, 128)) < 0.85
    shifted = np.roll(a, int(rng.integers(0, 5)), axis=1)
    b[sim] = np.where(keep[sim], shifted[sim], b[sim])
    lo, hi = [(-128, 127), (-127, 127), (-10, 10), (-128, -1), (0, 127), (-3, 4)][seed]
    sm = rng.integers(lo, hi + 1, 16).astype(np.int8)
    for gap in (0, 1, int(rng.integers(2, 127)), 127):
        assert np.array_equal(gpu.score_batch(a, b, sm, gap), oracle.batch(a, b, sm, gap)), (sm, gap)


@pytest.mark.parametrize("seed", range(8))
def test_packed_kernel_for_non_negative_folded_scores(gpu, oracle, seed):
    """sw128_pk_kernel (row N2: (1,-1,1) and every other matrix with min(sm) + gap >= 0): two alignments per register,
    v_perm lookup, v_pk_maximum3_f16 as a packed integer max.  Against the oracle and against the int32 kernel (flag 8),
    through all three entry shapes, with ragged batch sizes (an odd last pair leaves half a register idle)."""
    import torch
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(40000, 70000)) | 1 if seed % 2 else int(rng.integers(40000, 70000)) & ~1
    a = rng.integers(0, 4, (n, 128), dtype=np.uint8)
    b = rng.integers(0, 4, (n, 128), dtype=np.uint8)
    sim = rng.random(n) < 0.6
    keep = rng.random((n, 128)) < 0.9
    b[sim] = np.where(keep[sim], np.roll(a, int(rng.integers(0, 4)), axis=1)[sim], b[sim])
    b[: n // 50] = a[: n // 50]                      # some identical pairs: the largest scores the parameters allow
    gap = int((0, 1, 1, 3, 17, 60, 127, 2)[seed])
    if seed == 0:
        sm = match_matrix(1, -1)                     # SmithWaterman_8bit111simd / _8b111x32 (source.cpp:1105-1522)
        gap = 1
    elif seed == 1:
        sm = np.full(16, 127, np.int8)               # the largest scores: every cell 127, gap 0 -> 16256
        gap = 0
    else:
        sm = rng.integers(-gap, 127 - gap + 1, 16).astype(np.int8)      # min(sm) + gap >= 0, max(sm) + gap <= 127
    want = oracle.batch(a, b, sm, gap)
    gpu.set_schedule(4, 0)
    try:
        got = gpu.score_batch(a, b, sm, gap)
        assert np.array_equal(got, want), (sm, gap)
        assert np.array_equal(gpu.score_batch_packed(oracle.pack(a), oracle.pack(b), sm, gap), want)
        assert np.array_equal(gpu.score_one_vs_many(a, b[0], sm, gap), oracle.batch(a, np.repeat(b[:1], n, axis=0), sm, gap))
        gpu.set_schedule(4, 8)                       # the same parameters on the int32 kernel
        assert np.array_equal(gpu.score_batch(a, b, sm, gap), want)
    finally:
        gpu.set_schedule(0, 0)


def test_domain_edges(gpu, oracle):
    rng = np.random.default_rng(5)
    a = rng.integers(0, 4, (2048, 128), dtype=np.uint8)
    b = a.copy()
    b[1024:] = rng.integers(0, 4, (1024, 128), dtype=np.uint8)
    cases = [(match_matrix(127, -127), 127), (match_matrix(127, -127), 0), (match_matrix(127, 127), 0),
             (match_matrix(-128, -128), 0), (match_matrix(127, -128), 1), (match_matrix(0, 0), 0),
             (match_matrix(1, 0), 0), (match_matrix(-1, -1), 5)]
    for sm, gap in cases:
        got = gpu.score_batch(a, b, sm, gap)
        assert np.array_equal(got, oracle.batch(a, b, sm, gap)), (sm[:2], gap)
    # identical pairs reach the maximum representable score 128 * 127 = 16256
    assert int(gpu.score_batch(a[:4], a[:4], match_matrix(127, -127), 127).max()) == 16256


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 255, 257, 1000, 4099])
def test_ragged_batch_sizes(gpu, oracle, n):
    a, b = oracle.generate(n, 77, 5)
    sm = match_matrix(2, -3)
    for lanes in (64, 8, 4, 2):
        gpu.set_schedule(lanes, 0)
        try:
            assert np.array_equal(gpu.score_batch(a, b, sm, 5), oracle.batch(a, b, sm, 5)), (n, lanes)
        finally:
            gpu.set_schedule(0, 0)


def test_empty_batch(gpu):
    out = gpu.score_batch(np.zeros((0, 128), np.uint8), np.zeros((0, 128), np.uint8), match_matrix(1, -1), 1)
    assert out.shape == (0,)


def test_bases_modulo_4_like_the_oracle(gpu, oracle):
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (512, 128), dtype=np.uint8)
    b = rng.integers(0, 256, (512, 128), dtype=np.uint8)
    sm = match_matrix(4, -5)
    want = oracle.batch(a, b, sm, 3)
    assert np.array_equal(gpu.score_batch(a, b, sm, 3), want)
    assert np.array_equal(gpu.score_batch(a, b, sm, 3), gpu.score_batch(a & 3, b & 3, sm, 3))
    try:                                                    # every schedule and cell body, and the sibling entry points
        for lanes in (64, 32, 16, 8, 4, 2):
            for flags in (0, 1, 2) + ((4,) if lanes in (16, 8, 4) else ()):
                gpu.set_schedule(lanes, flags)
                assert np.array_equal(gpu.score_batch(a, b, sm, 3), want), (lanes, flags)
    finally:
        gpu.set_schedule(0, 0)
    assert np.array_equal(gpu.score_one_vs_many(a, b[0], sm, 3), oracle.batch(a, np.broadcast_to(b[0], a.shape).copy(), sm, 3))
    a1 = rng.integers(0, 256, (16, 256), dtype=np.uint8)
    b1 = rng.integers(0, 256, (16, 256), dtype=np.uint8)
    assert np.array_equal(gpu.score_banded_affine(a1, b1, sm, 6, 2), oracle.banded_affine(a1 & 3, b1 & 3, sm, 6, 2))
