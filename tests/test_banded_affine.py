"""BASELINE.json configs[4] (1024 x 1024 affine gap, band 128): an extension the reference has no counterpart for.

PARITY UNPINNED BY THE REFERENCE: the oracle here is the build's own scalar banded Gotoh (oracle/sw_oracle.c).  What can
be cross-checked is: (1) the oracle against an independent numpy formulation of the same recurrences, (2) the
linear-gap special case (open == extend) against the reference-pinned scorer on pairs whose optimal alignment cannot
leave the band, (3) the GPU kernel against the oracle, bit-exact."""
import numpy as np
import pytest

from conftest import match_matrix

NEG = -(1 << 29)


def numpy_banded_gotoh(a, b, sm, gap_open, gap_ext):
    """Independent restatement: anti-diagonal-free, row by row with explicit band mask, int64 arrays."""
    n = len(a)
    sm = np.asarray(sm, np.int64).reshape(4, 4)
    H = np.zeros((n + 1, n + 1), np.int64)
    E = np.full((n + 1, n + 1), NEG, np.int64)
    F = np.full((n + 1, n + 1), NEG, np.int64)
    best = 0
    for i in range(1, n + 1):
        lo, hi = max(1, i - 64), min(n, i + 63)
        for j in range(lo, hi + 1):
            E[i, j] = max(E[i, j - 1] - gap_ext, H[i, j - 1] - gap_open)
            F[i, j] = max(F[i - 1, j] - gap_ext, H[i - 1, j] - gap_open)
            H[i, j] = max(0, H[i - 1, j - 1] + sm[a[i - 1] & 3, b[j - 1] & 3], E[i, j], F[i, j])
            best = max(best, H[i, j])
    return int(best)


def _related(rng, n, length, sub=0.08, indel=0.02):
    a = rng.integers(0, 4, (n, length), dtype=np.uint8)
    b = np.zeros_like(a)
    for k in range(n):
        out, i = [], 0
        while len(out) < length:
            r = rng.random()
            if r < indel:
                out.append(rng.integers(0, 4))
            elif r < 2 * indel:
                i += 1
            else:
                out.append(a[k, i % length] if rng.random() > sub else rng.integers(0, 4))
                i += 1
        b[k] = out[:length]
    return a, b


def test_oracle_matches_independent_numpy_formulation(oracle):
    rng = np.random.default_rng(21)
    a, b = _related(rng, 6, 192)
    b[3:] = rng.integers(0, 4, (3, 192), dtype=np.uint8)
    for sm, go, ge in ((match_matrix(2, -3), 5, 1), (match_matrix(5, -4), 10, 0), (match_matrix(1, -1), 0, 0),
                       (rng.integers(-20, 21, 16).astype(np.int8), 7, 7)):
        got = oracle.banded_affine(a, b, sm, go, ge)
        want = [numpy_banded_gotoh(a[k], b[k], sm, go, ge) for k in range(a.shape[0])]
        assert list(got) == want


def test_linear_gap_special_case_agrees_with_the_pinned_scorer(oracle, golden):
    """open == extend is the reference's linear gap.  With a mismatch no better than -gap-ish the optimal local
    alignment of a pair that differs by substitutions only is gap-free, stays on the main diagonal, hence in the band:
    banded-affine(open = extend = gap) must then equal SmithWaterman(sm, gap) (source.cpp:35-60, pinned by fixtures)."""
    rng = np.random.default_rng(8)
    a = rng.integers(0, 4, (64, 128), dtype=np.uint8)
    b = a.copy()
    m = rng.random((64, 128)) < 0.1
    b[m] = (b[m] + rng.integers(1, 4, m.sum())) % 4
    sm = match_matrix(2, -3)
    lin = oracle.batch(a, b, sm, 127)             # gaps priced out of reach: the optimum is gap-free for both scorers
    assert np.array_equal(oracle.banded_affine(a, b, sm, 127, 127), lin)
    f = golden("f2_structured")                   # identical pairs of the reference fixtures: 128 * match
    ident = np.array([np.array_equal(x, y) for x, y in zip(f["seq1"], f["seq2"])])
    assert ident.sum() >= 16
    assert np.array_equal(oracle.banded_affine(f["seq1"][ident], f["seq2"][ident], f["sm"][0], 15, 15),
                          f["scores"][0][ident])


@pytest.mark.gpu
@pytest.mark.parametrize("length", [64, 128, 1024, 1792])
def test_gpu_banded_affine_matches_the_oracle(gpu, oracle, length):
    rng = np.random.default_rng(100 + length)
    n = 96 if length >= 1024 else 300
    a, b = _related(rng, n, length)
    b[: n // 4] = rng.integers(0, 4, (n // 4, length), dtype=np.uint8)        # unrelated pairs
    a[n // 4: n // 4 + 4] = b[n // 4: n // 4 + 4]                              # identical pairs
    shift = np.roll(a[-8:], 70, axis=1)                                       # offset beyond the band
    b[-8:] = shift
    for sm, go, ge in ((match_matrix(2, -3), 5, 1), (match_matrix(10, -30), 15, 15), (match_matrix(1, -1), 0, 0),
                       (match_matrix(127, -127), 127, 0), (rng.integers(-128, 128, 16).astype(np.int8), 11, 3),
                       (match_matrix(2, -3), 1, 4), (match_matrix(5, -4), 0, 7)):      # open < extend: the other kernel body
        got = gpu.score_banded_affine(a, b, sm, go, ge)
        want = oracle.banded_affine(a, b, sm, go, ge)
        assert np.array_equal(got, want), (length, go, ge, int((got != want).sum()))


@pytest.mark.gpu
def test_gpu_banded_affine_batch_shapes_and_errors(gpu, oracle):
    rng = np.random.default_rng(3)
    sm = match_matrix(2, -3)
    for n in (1, 3, 4, 5, 7, 1025):
        a, b = _related(rng, n, 256)
        assert np.array_equal(gpu.score_banded_affine(a, b, sm, 4, 2), oracle.banded_affine(a, b, sm, 4, 2)), n
    assert gpu.score_banded_affine(np.zeros((0, 128), np.uint8), np.zeros((0, 128), np.uint8), sm, 1, 1).shape == (0,)
    with pytest.raises(gpu.SwmiError) as e:
        gpu.score_banded_affine(np.zeros((1, 32), np.uint8), np.zeros((1, 32), np.uint8), sm, 1, 1)
    assert e.value.code == gpu.ERR_INVALID_ARGUMENT
    with pytest.raises(gpu.SwmiError) as e:
        gpu.score_banded_affine(np.zeros((1, 128), np.uint8), np.zeros((1, 128), np.uint8), sm, -1, 1)
    assert e.value.code == gpu.ERR_DOMAIN


def test_banded_affine_kernel_choice_needs_no_device(swmi_mod):
    """swmi_banded_affine_kernel_for: the packed kernel (two alignments per wavefront, 16-bit halves) where every value stays
    a finite half-precision pattern -- len * max(s) + 2 max(0, -min s) + open + extend + 64 < 0x7C00 -- else the int32 cell."""
    k = swmi_mod.banded_affine_kernel_for
    assert k(1024, match_matrix(2, -3), 5, 1) == ("sw_banded_affine_pk_kernel<1>", 2)
    assert k(1024, match_matrix(2, -3), 1, 4) == ("sw_banded_affine_pk_kernel<0>", 2)
    assert k(1792, match_matrix(10, -30), 15, 15) == ("sw_banded_affine_pk_kernel<1>", 2)      # 17920 + 60 + 30 + 64
    assert k(1024, match_matrix(30, -30), 5, 1)[0] == "sw_banded_affine_pk_kernel<1>"          # 30720 + 60 + 6 + 64 = 30850 < 31744
    assert k(1024, match_matrix(31, -30), 5, 1) == ("sw_banded_affine_kernel<1,1>", 1)         # 31744 + ...: too large, still < 2^15
    assert k(1024, match_matrix(32, -30), 5, 1) == ("sw_banded_affine_kernel<1,0>", 1)         # 32768: the plain int32 cell
    assert k(64, match_matrix(127, -127), 127, 0) == ("sw_banded_affine_pk_kernel<1>", 2)      # 8128 + 254 + 127 + 64
    assert k(256, match_matrix(127, -127), 0, 127)[0] == "sw_banded_affine_kernel<0,1>"       # 32512 + 254 + 127 + 64 > 0x7C00, < 2^15


@pytest.mark.gpu
@pytest.mark.parametrize("sm_go_ge, want_kernel", [
    ((match_matrix(2, -3), 5, 1), "sw_banded_affine_pk_kernel<1>"),
    ((match_matrix(2, -3), 1, 4), "sw_banded_affine_pk_kernel<0>"),
    ((match_matrix(3, 1), 4, 2), "sw_banded_affine_pk_kernel<1>"),            # no negative score: bias 0
    ((match_matrix(3, 1), 0, 6), "sw_banded_affine_pk_kernel<0>"),
    ((match_matrix(1, -1), 0, 0), "sw_banded_affine_pk_kernel<1>"),           # free gaps
    ((match_matrix(0, -128), 127, 127), "sw_banded_affine_pk_kernel<1>"),     # the largest bias, nothing ever scores
    ((match_matrix(20, -128), 3, 127), "sw_banded_affine_pk_kernel<0>"),      # bias 128 above open: the floor is the bias
    ((match_matrix(20, -2), 100, 127), "sw_banded_affine_pk_kernel<0>"),      # open above the bias: the floor is open
    ((match_matrix(40, -30), 5, 1), "sw_banded_affine_kernel<1,0>"),          # outside the packed domain at len 1024
])
def test_gpu_banded_affine_every_kernel_by_name(gpu, oracle, sm_go_ge, want_kernel):
    """Each cell body of the banded-affine extension against the oracle, the launch checked to run the kernel named: related
    pairs, unrelated pairs, identical pairs, an offset beyond the band, odd batch sizes (the packed kernel's last wavefront
    then scores one pair twice).  Parity unpinned by the reference (no affine gap there)."""
    sm, go, ge = sm_go_ge
    length = 1024 if "pk" not in want_kernel or go + ge < 200 else 512
    name, per = gpu.banded_affine_kernel_for(length, sm, go, ge)
    assert name == want_kernel and per == (2 if "_pk_" in name else 1)
    rng = np.random.default_rng(abs(hash(want_kernel + str(go))) % 1000)
    for n in (1, 2, 7, 64, 131):
        a, b = _related(rng, n, length)
        if n >= 7:
            b[0] = rng.integers(0, 4, length, dtype=np.uint8)
            a[1] = b[1]
            b[2] = np.roll(a[2], 70)
            a[3] = 0; b[3] = 0                                                 # homopolymers
        got = gpu.score_banded_affine(a, b, sm, go, ge)
        want = oracle.banded_affine(a, b, sm, go, ge)
        assert np.array_equal(got, want), (name, n, int((got != want).sum()), got[:8], want[:8])
    mats = [rng.integers(-128, 128, 16).astype(np.int8) for _ in range(6)]
    for m in mats:                                                             # random matrices at a length that keeps them packed
        m = np.clip(m, -128, 60).astype(np.int8)
        o, e = (int(rng.integers(0, 128)), int(rng.integers(0, 128)))
        a, b = _related(rng, 33, 256)
        got = gpu.score_banded_affine(a, b, m, o, e)
        want = oracle.banded_affine(a, b, m, o, e)
        assert np.array_equal(got, want), (gpu.banded_affine_kernel_for(256, m, o, e), m, o, e, int((got != want).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("length, mismatch, go, ge", [(1024, -3, 5, 1), (1792, -2, 0, 0), (64, -128, 127, 127), (512, -30, 1, 100), (333, -7, 90, 3)])
def test_gpu_banded_affine_packed_kernel_at_the_edge_of_its_domain(gpu, oracle, length, mismatch, go, ge):
    """The packed kernel keeps every 16-bit half below 0x7C00 (a finite half-precision pattern): with the LARGEST match score
    its domain admits at this length, identical sequences reach len * match -- the highest value any half can take, on top
    of which sit the growing bias (up to 17 x max(0, -min s)) and the hand-over floors.  One score above, the int32 cell runs."""
    bias = max(0, -mismatch)
    limit = (0x7C00 - 64 - 18 * bias - go - ge - 1) // length
    top = min(127, limit)                                                      # (an int8 score; a short band leaves room to spare)
    assert top > 0
    sm = match_matrix(top, mismatch)
    assert gpu.banded_affine_kernel_for(length, sm, go, ge)[0].startswith("sw_banded_affine_pk_kernel")
    if limit < 127:
        assert not gpu.banded_affine_kernel_for(length, match_matrix(top + 1, mismatch), go, ge)[0].startswith("sw_banded_affine_pk_kernel")
    rng = np.random.default_rng(length + go)
    a, b = _related(rng, 41, length, sub=0.03, indel=0.005)
    a[0] = rng.integers(0, 4, length, dtype=np.uint8); b[0] = a[0]             # identical: the top score len * match
    a[1] = 2; b[1] = 2                                                         # homopolymer, identical
    b[2] = np.roll(a[2], 1)                                                    # one diagonal off
    got = gpu.score_banded_affine(a, b, sm, go, ge)
    want = oracle.banded_affine(a, b, sm, go, ge)
    assert np.array_equal(got, want), (top, int((got != want).sum()), got[:4], want[:4])
    assert got[0] == length * top and got[1] == length * top
