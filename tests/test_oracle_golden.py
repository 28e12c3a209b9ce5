"""The CPU oracle (oracle/sw_oracle.c) against every golden vector produced by the real reference.

Fixtures come from tests/golden/make_golden.py, which ran the reference's own SmithWaterman
(source.cpp:35-60) and required simd4/7/8/9 (source.cpp:462-1071) to agree inside their domains.
These tests are what pins the oracle; the GPU parity tests then compare the HIP path with both.
"""
import numpy as np
import pytest

from conftest import match_matrix


@pytest.mark.parametrize("name", ["f1_random", "f2_structured", "f3_harness", "f4_param_sweep"])
def test_oracle_matches_reference_scores(oracle, golden, name):
    f = golden(name)
    for p in range(f["sm"].shape[0]):
        got = oracle.batch(f["seq1"], f["seq2"], f["sm"][p], int(f["gap"][p]))
        assert np.array_equal(got, f["scores"][p]), "parameter set %d of %s" % (p, name)


def test_known_answers_of_the_reference_drivers(oracle, golden):
    # first pairs of the TestSimdSmithWaterman stream (source.cpp:2944-2959), libstdc++ draw
    f = golden("f3_harness")
    assert list(f["scores"][0][:8]) == [80, 80, 70, 95, 70, 80, 80, 75]     # (10,-30,15)
    assert list(f["scores"][1][:8]) == [18, 20, 14, 24, 22, 21, 18, 18]     # (1,-1,1)
    assert int(f["sum_first_100000"][0]) == 7550735 and int(f["sum_first_100000"][1]) == 1816827
    assert int(f["min_first_100000"][0]) == 50 and int(f["max_first_100000"][0]) == 195
    head1 = "".join(map(str, f["seq1"][0][:40]))
    assert head1 == "2220122332200113322233022202112202211032"          # SpeedTest pair, source.cpp:3033-3040
    assert oracle.score(f["seq1"][0], f["seq2"][0], match_matrix(10, -30), 15) == 80
    assert oracle.score(f["seq1"][0], f["seq2"][0], match_matrix(1, -1), 1) == 18


def test_identical_pair_scores_128_times_match(oracle):
    rng = np.random.default_rng(7)
    a = rng.integers(0, 4, 128, dtype=np.uint8)
    assert oracle.score(a, a, match_matrix(10, -30), 15) == 1280
    assert oracle.score(a, a, match_matrix(127, -127), 127) == 128 * 127
    assert oracle.score(a, a, match_matrix(1, -1), 1) == 128


def test_sibling_functions(oracle, golden):
    # SmithWaterman_111 (source.cpp:1073-1103) == general scorer with +1/-1/1; unpack (source.cpp:1580-1583)
    f = golden("f5_siblings")
    got = oracle.batch(f["seq1"], f["seq2"], match_matrix(1, -1), 1)
    assert np.array_equal(got, f["scores_111"])
    assert np.array_equal(oracle.unpack(f["packed"]), f["unpacked"])
    assert np.array_equal(oracle.pack(f["unpacked"]), f["packed"])
    # 32 seq1 x one seq2 (source.cpp:1227-1230): block k scores seq1[32k..32k+31] against seq2[k]
    for blk in range(f["scores_111x32"].shape[0]):
        s2 = np.repeat(f["seq2"][blk][None, :], 32, axis=0)
        assert np.array_equal(oracle.batch(f["seq1"][32 * blk:32 * blk + 32], s2, match_matrix(1, -1), 1),
                              f["scores_111x32"][blk])


def test_bases_are_taken_modulo_4(oracle):
    rng = np.random.default_rng(3)
    a = rng.integers(0, 4, (16, 128), dtype=np.uint8)
    b = rng.integers(0, 4, (16, 128), dtype=np.uint8)
    hi = rng.integers(0, 64, (16, 128), dtype=np.uint8) * 4
    sm = match_matrix(2, -3)
    assert np.array_equal(oracle.batch(a, b, sm, 5), oracle.batch(a + hi, b + hi, sm, 5))
