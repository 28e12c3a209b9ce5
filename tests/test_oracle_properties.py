"""Property tests of the CPU oracle (hypothesis): algebraic facts of the recurrence (source.cpp:49-53) that hold for
every input, used again at full size on the GPU (tests/test_gpu_fullsize.py)."""
import numpy as np
from hypothesis import given, settings, strategies as st

from conftest import Oracle

ORACLE = Oracle()
seqs = st.lists(st.integers(0, 3), min_size=128, max_size=128).map(lambda v: np.array(v, np.uint8))
matrices = st.lists(st.integers(-40, 40), min_size=16, max_size=16).map(lambda v: np.array(v, np.int8))
gaps = st.integers(0, 40)


@settings(max_examples=60, deadline=None)
@given(seqs, seqs, matrices, gaps)
def test_swap_with_transposed_matrix(a, b, sm, gap):
    smT = sm.reshape(4, 4).T.copy().reshape(16)
    assert ORACLE.score(a, b, sm, gap) == ORACLE.score(b, a, smT, gap)


@settings(max_examples=60, deadline=None)
@given(seqs, seqs, matrices, gaps, st.integers(1, 3))
def test_homogeneous_of_degree_one(a, b, sm, gap, k):
    assert ORACLE.score(a, b, (sm * k).astype(np.int8), gap * k) == k * ORACLE.score(a, b, sm, gap)


@settings(max_examples=60, deadline=None)
@given(seqs, seqs, matrices, gaps)
def test_bounds_and_gap_monotonicity(a, b, sm, gap):
    s = ORACLE.score(a, b, sm, gap)
    assert 0 <= s <= 128 * max(int(sm.max()), 0)
    assert ORACLE.score(a, b, sm, gap + 1) <= s
    rev = ORACLE.score(a[::-1].copy(), b[::-1].copy(), sm, gap)      # reversing both sequences keeps the optimum
    assert rev == s
