"""the reference's semi-global adaptive-band X-drop aligner
(SemiGlobal_AdaptiveBanded_XDrop_111_32_70, source.cpp:1836-1976, and its four SIMD variants :1978-2725).

Fixture F6 holds inputs and the (score, traceback) the REAL reference returned -- scalar and all four SIMD variants
agreeing, the reference's own TestSemiGlobal criterion (source.cpp:2774-2784).  The CPU oracle (oracle/sg_oracle.c) is
pinned to it here; the GPU path is compared with both, position by position."""
import numpy as np
import pytest


def _paths_from_fixture(f):
    step = {1: (1, 1), 2: (1, 0), 3: (0, 1)}
    out = []
    for k in range(len(f["scores"])):
        moves = f["moves"][f["move_offsets"][k]: f["move_offsets"][k + 1]]
        d = np.array([step[int(m)] for m in moves], np.int32).reshape(-1, 2)
        out.append(np.concatenate([np.zeros((1, 2), np.int32), np.cumsum(d, axis=0, dtype=np.int32)]))
    return out


def test_oracle_reproduces_the_reference_results(oracle, golden):
    f = golden("f6_semiglobal")
    paths = _paths_from_fixture(f)
    for k in range(len(f["scores"])):
        score, tb = oracle.semiglobal(f["seq1"][k], f["seq2"][k])
        assert score == int(f["scores"][k]), k
        assert len(tb) == int(f["lengths"][k]) and np.array_equal(tb, paths[k]), k
        assert tuple(tb[-1]) == tuple(f["ends"][k])


def test_calm_window_invariant_on_the_reference_recurrence(oracle, golden):
    """The GPU sweeps skip the X-drop test in windows of 8 rounds that start with every band cell alive and at least 13 above
    the threshold (sg_kernels.hip, "CALM WINDOWS").  The claim behind it -- inside such a window the rule of source.cpp:1938-1941
    drops nothing -- is checked here on the CPU restatement of the reference's recurrence, for the fixture's inputs, for
    substitution-only relatives (nearly every window calm) and for inputs built to graze the threshold; and the condition
    has teeth: with a margin of 1 instead of 13 the same inputs DO drop cells inside 'calm' windows."""
    f = golden("f6_semiglobal")
    rng = np.random.default_rng(813)
    cases = [(f["seq1"][k], f["seq2"][k]) for k in range(len(f["scores"]))]
    for p_sub in (0.0, 0.05, 0.15, 0.3):
        a = rng.integers(0, 4, 16384, dtype=np.uint8)
        b = np.where(rng.random(16384) < p_sub, rng.integers(0, 4, 16384, dtype=np.uint8), a).astype(np.uint8)
        cases.append((a, b))
    for k in range(24):                          # runs of guaranteed mismatches of 40 .. 100 around the X-drop limit of 70, and gaps
        a = rng.integers(0, 4, 16384, dtype=np.uint8)
        b = a.copy()
        for _ in range(1 + k % 6):
            lo, run = int(rng.integers(100, 16000)), int(rng.integers(40, 101))
            b[lo: lo + run] = (a[lo: lo + run] + 1) & 3
        if k % 3 == 0:
            cut, sh = int(rng.integers(100, 16000)), int(rng.integers(1, 30))
            b[cut:] = np.roll(b, sh)[cut:]
        cases.append((a, b))
    for run in (45, 50, 60, 69, 70, 71, 80):     # stretches without a match at any offset: the whole band sinks towards the threshold
        a = rng.integers(0, 4, 16384, dtype=np.uint8)
        b = a.copy()
        for lo in (3000, 8000, 12000):
            a[lo: lo + run], b[lo: lo + run] = 0, 1
        cases.append((a, b))
    calm_total = grazed = 0
    for a, b in cases:
        windows, calm, dropped = oracle.calm_windows(a, b, 8, 13)
        assert dropped == 0 and 0 < windows <= 4096 and calm <= windows
        calm_total += calm
        for window, margin in ((8, 12 + 1), (16, 24 + 1), (4, 6 + 1)):                  # margin = window + window / 2 + 1 for other windows too
            assert oracle.calm_windows(a, b, window, margin)[2] == 0
        grazed += oracle.calm_windows(a, b, 8, 1)[2]
    assert calm_total > 2000 * 4                 # the substitution-only relatives alone: > 90 % of their 4096 windows
    assert oracle.calm_windows(cases[17][0], cases[17][1], 8, 13)[1] > 3700      # (5 % substitutions: SpeedtestSemiGlobal's inputs)
    assert grazed > 0                            # "every cell alive" alone is not enough: with a margin of 1 cells ARE dropped inside windows


@pytest.fixture
def sg_kernels(swmi_mod):
    """Select the sweep mapping (4 / 2 / 1 = band over 4 / 2 lanes / in one lane, 10 * lanes + W = a scheduling target)
    through swmi_semiglobal_set_mapping; by default the batch size decides."""
    yield swmi_mod.semiglobal_set_mapping
    swmi_mod.semiglobal_set_mapping(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [4, 2, 1, 41, 42, 43, 44, 21, 22, 23, 11, 12])
def test_gpu_semiglobal_matches_reference_fixtures(gpu, golden, sg_kernels, sweep):
    sg_kernels(sweep)
    f = golden("f6_semiglobal")
    paths = _paths_from_fixture(f)
    scores, tbs, lengths = gpu.semiglobal_xdrop(f["seq1"], f["seq2"])
    assert np.array_equal(scores, f["scores"])
    assert np.array_equal(lengths.astype(np.int64), f["lengths"].astype(np.int64))
    for k in range(len(paths)):
        assert np.array_equal(tbs[k], paths[k]), "traceback of case %d differs" % k


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [4, 2, 1, 22, 41, 12])
def test_gpu_semiglobal_matches_oracle_on_fresh_inputs(gpu, oracle, sg_kernels, sweep):
    sg_kernels(sweep)
    rng = np.random.default_rng(77)
    n = 21 if sweep == 4 else 70                 # odd / not a multiple of 64: ragged last wavefront
    a = rng.integers(0, 4, (n, 16384), dtype=np.uint8)
    b = np.zeros_like(a)
    for k in range(n):                           # indel-rich relatives at different divergence (TestSemiGlobal's recipe)
        p_sub, p_ins, p_del = [(0.10, 0.10, 0.10), (0.02, 0.01, 0.01), (0.2, 0.05, 0.05), (0.0, 0.0, 0.0)][k % 4]
        out, i = [], 0
        while len(out) < 16384:
            r = rng.random()
            if i >= 16384 or r < p_ins:
                out.append(rng.integers(0, 4))
            elif r < p_ins + p_del:
                i += 1
            elif r < p_ins + p_del + p_sub:
                out.append(rng.integers(0, 4)); i += 1
            else:
                out.append(a[k, i]); i += 1
        b[k] = out
    b[5] = rng.integers(0, 4, 16384, dtype=np.uint8)            # unrelated: the X-drop rule ends the sweep early or late
    b[6] = np.concatenate([a[6, 300:], rng.integers(0, 4, 300, dtype=np.uint8)])   # offset far beyond the band
    scores, tbs, lengths = gpu.semiglobal_xdrop(a, b)
    for k in range(n):
        want_score, want_tb = oracle.semiglobal(a[k], b[k])
        assert int(scores[k]) == want_score, k
        assert int(lengths[k]) == len(want_tb) and np.array_equal(tbs[k], want_tb), k


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [4, 2, 1])
def test_gpu_semiglobal_small_cap_and_empty(gpu, oracle, sg_kernels, sweep):
    sg_kernels(sweep)
    rng = np.random.default_rng(4)
    a = rng.integers(0, 4, (2, 16384), dtype=np.uint8)
    b = a.copy()
    scores, tbs, lengths = gpu.semiglobal_xdrop(a, b, cap=100)
    assert list(scores) == [16384, 16384] and list(lengths) == [16385, 16385]
    assert np.array_equal(tbs[0], np.stack([np.arange(100), np.arange(100)], axis=1))   # the first 100 steps of the diagonal
    s0, t0, l0 = gpu.semiglobal_xdrop(np.zeros((0, 16384), np.uint8), np.zeros((0, 16384), np.uint8))
    assert s0.shape == (0,) and t0 == []


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [4, 2, 1])
def test_gpu_semiglobal_ragged_batches(gpu, golden, sg_kernels, sweep):
    """Batches of 1, 3 and 65 alignments: the lanes of a sweep wavefront (and the rows of its record flush) past the last
    alignment shadow that alignment -- they store its records on top of its own, the same bytes -- and a walk wavefront's
    tail lanes store nothing.  Results as in the fixture, whatever stands beside an alignment."""
    sg_kernels(sweep)
    f = golden("f6_semiglobal")
    paths = _paths_from_fixture(f)
    reps = np.arange(65) % len(paths)
    for n in (1, 3, 65):
        scores, tbs, lengths = gpu.semiglobal_xdrop(f["seq1"][reps[:n]], f["seq2"][reps[:n]])
        assert np.array_equal(scores, f["scores"][reps[:n]])
        for k in range(n):
            assert np.array_equal(tbs[k], paths[reps[k]]), (n, k)


@pytest.mark.gpu
def test_gpu_semiglobal_phase_timing_entry(gpu, oracle):
    """swmi_semiglobal_time_device: same results as the plain device call, two positive kernel durations."""
    import torch
    rng = np.random.default_rng(5)
    a = rng.integers(0, 4, (3, 16384), dtype=np.uint8)
    b = a.copy()
    b[:, ::37] = rng.integers(0, 4, b[:, ::37].shape, dtype=np.uint8)
    dev = torch.device("cuda", 0)
    d1, d2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    cap = gpu.SG_MAX_TRACEBACK
    scores = torch.zeros(3, dtype=torch.int32, device=dev)
    lengths = torch.zeros(3, dtype=torch.int32, device=dev)
    tb = torch.zeros((3, cap, 2), dtype=torch.int32, device=dev)
    sweep_ms, tb_ms = gpu.semiglobal_time_device(d1.data_ptr(), d2.data_ptr(), 3, scores.data_ptr(), tb.data_ptr(), cap,
                                                 lengths.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert sweep_ms > 0 and tb_ms > 0
    for k in range(3):
        want_score, want_tb = oracle.semiglobal(a[k], b[k])
        assert int(scores[k]) == want_score and int(lengths[k]) == len(want_tb)
        assert np.array_equal(tb[k, : len(want_tb)].cpu().numpy(), want_tb)


@pytest.mark.gpu
def test_gpu_semiglobal_host_entry_pipelines_chunks(gpu, oracle):
    """swmi_semiglobal_xdrop with more alignments than one chunk (8192): two buffer sets in flight, trimmed 2-D copies."""
    rng = np.random.default_rng(6)
    base_a = rng.integers(0, 4, (6, 16384), dtype=np.uint8)
    base_b = base_a.copy()
    for k in range(6):
        idx = rng.integers(0, 16384, 200 * (k + 1))
        base_b[k, idx] = rng.integers(0, 4, idx.shape, dtype=np.uint8)
    base_b[5, 9000:] = np.roll(base_b[5], 25)[9000:]                 # one with an indel
    want = [oracle.semiglobal(base_a[k], base_b[k]) for k in range(6)]
    n = 8192 + 301
    pick = rng.integers(0, 6, n)
    a, b = base_a[pick], base_b[pick]
    cap = 2048
    scores, tbs, lengths = gpu.semiglobal_xdrop(a, b, cap=cap)
    assert np.array_equal(scores, np.array([want[k][0] for k in pick], np.int32))
    assert np.array_equal(lengths, np.array([len(want[k][1]) for k in pick], np.uint32))
    for j in range(0, n, 97):
        assert np.array_equal(tbs[j], want[pick[j]][1][:cap]), j


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [4, 2, 1])
def test_gpu_semiglobal_bytes_that_are_no_base(gpu, oracle, sg_kernels, sweep):
    """Outside the reference's domain (its traceback indexes the 4x4 matrix with the raw byte, source.cpp:1961), but defined
    here: a byte >= 4 scores as a mismatch against everything, also against itself -- the meaning the reference's sweep
    gives it (:1918-1920).  Oracle and every GPU mapping agree."""
    sg_kernels(sweep)
    rng = np.random.default_rng(8)
    a = rng.integers(0, 4, (3, 16384), dtype=np.uint8)
    b = a.copy()
    for k in range(3):
        idx = rng.integers(0, 16384, 300)
        b[k, idx] = rng.integers(0, 4, 300, dtype=np.uint8)
    a[0, 1000:1010] = 4                     # an N-run in seq1 only
    b[1, 5000:5003] = 255                   # junk bytes in seq2 only
    a[2, 7000:7004] = 9; b[2, 7000:7004] = 9   # the same non-base in both: still a mismatch
    scores, tbs, lengths = gpu.semiglobal_xdrop(a, b)
    for k in range(3):
        want_score, want_tb = oracle.semiglobal(a[k], b[k])
        assert int(scores[k]) == want_score and np.array_equal(tbs[k], want_tb), k


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [4, 2, 1])
def test_gpu_semiglobal_scores_only(gpu, oracle, sg_kernels, sweep):
    """cap = 0: scores and path lengths without any positions (the traceback buffer may be absent)."""
    sg_kernels(sweep)
    rng = np.random.default_rng(12)
    a = rng.integers(0, 4, (5, 16384), dtype=np.uint8)
    b = a.copy()
    b[:, ::29] = rng.integers(0, 4, b[:, ::29].shape, dtype=np.uint8)
    b[4] = rng.integers(0, 4, 16384, dtype=np.uint8)                # unrelated: drops out early
    import ctypes
    scores = np.zeros(5, np.int32)
    lengths = np.zeros(5, np.uint32)
    rc = gpu.load().swmi_semiglobal_xdrop(a.ctypes.data, b.ctypes.data, 5, scores.ctypes.data, None, 0, lengths.ctypes.data)
    assert rc == 0, gpu.last_error()
    for k in range(5):
        want_score, want_tb = oracle.semiglobal(a[k], b[k])
        assert int(scores[k]) == want_score and int(lengths[k]) == len(want_tb), k


@pytest.mark.gpu
def test_gpu_semiglobal_mapping_choice_and_argument_check(gpu, swmi_mod):
    """swmi_semiglobal_kernels_for_batch reports what the launcher picks (band over 4 lanes for small batches,
    2 lanes in between, one lane per alignment from 49152 on a 256-CU device -- except where a batch gives the SIMDs two and a
    half or three wavefronts of 32 alignments); swmi_semiglobal_set_mapping rejects what is no mapping and leaves the setting alone."""
    if gpu.device_info()["compute_units"] == 256:
        want = {1: "sg_forward_split_kernel<4, 1>", 16384: "sg_forward_split_kernel<4, 1>", 32768: "sg_forward_split_kernel<2, 1>",
                49152: "sg_forward_lane_kernel<1>", 65536: "sg_forward_lane_kernel<1>", 81920: "sg_forward_split_kernel<2, 2>",
                98304: "sg_forward_split_kernel<2, 2>", 131072: "sg_forward_lane_kernel<2>", 196608: "sg_forward_lane_kernel<2>",
                262144: "sg_forward_lane_kernel<2>"}
        for n, name in want.items():
            assert swmi_mod.semiglobal_kernels_for_batch(n) == (name, "sg_walk_lane_kernel + sg_expand_kernel"), n
    swmi_mod.semiglobal_set_mapping(2)
    try:
        assert swmi_mod.semiglobal_kernels_for_batch(1000)[0] == "sg_forward_split_kernel<2, 2>"
        for bad in (0, 3, 5, 13, 14, 20, 24, 45, 101, 221):
            with pytest.raises(swmi_mod.SwmiError):
                swmi_mod.semiglobal_set_mapping(bad)
            assert swmi_mod.semiglobal_kernels_for_batch(1000)[0] == "sg_forward_split_kernel<2, 2>"      # unchanged
        swmi_mod.semiglobal_set_mapping(12)
        assert swmi_mod.semiglobal_kernels_for_batch(5)[0] == "sg_forward_lane_kernel<2>"
    finally:
        swmi_mod.semiglobal_set_mapping(-1)


def _fixture_moves_words(f, k):
    """Fixture F6 stores alignment k's steps in ASCENDING order as 1 diagonal / 2 up / 3 left; the library's move words hold them in
    walking order (last step first) as 3 diagonal / 2 up / 1 left, 32 per 64-bit word."""
    steps = f["moves"][f["move_offsets"][k]: f["move_offsets"][k + 1]][::-1]
    code = {1: 3, 2: 2, 3: 1}
    words = np.zeros(1040, np.uint64)
    for t, m in enumerate(steps):
        words[t >> 5] |= np.uint64(code[int(m)]) << np.uint64(2 * (t & 31))
    return words


def test_expand_moves_on_the_host_reproduces_the_reference_tracebacks(swmi_mod, golden):
    """swmi_semiglobal_expand_moves (no device): move words built from the reference's own paths (fixture F6) come back as the
    reference's traceback vectors, position by position; `cap` cuts the list, a length outside [1, 32769] is refused."""
    f = golden("f6_semiglobal")
    paths = _paths_from_fixture(f)
    for k in range(len(paths)):
        words = _fixture_moves_words(f, k)
        tb = swmi_mod.semiglobal_expand_moves(words, int(f["lengths"][k]))
        assert np.array_equal(tb, paths[k]), k
        assert np.array_equal(swmi_mod.semiglobal_expand_moves(words, int(f["lengths"][k]), cap=100), paths[k][:100])
    assert np.array_equal(swmi_mod.semiglobal_expand_moves(np.zeros(1040, np.uint64), 1), np.zeros((1, 2), np.int32))
    for bad in (0, 32770):
        with pytest.raises(swmi_mod.SwmiError):
            swmi_mod.semiglobal_expand_moves(np.zeros(1040, np.uint64), bad)


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [-1, 4, 2, 1])
def test_gpu_semiglobal_moves_entry_matches_reference_fixtures(gpu, golden, sg_kernels, sweep):
    """swmi_semiglobal_xdrop_moves: the traceback as 2-bit moves (8 KB per alignment over PCIe instead of 262 KB of positions).
    Scores, lengths, the move words themselves and their host-side expansion against fixture F6 (the reference's results)."""
    sg_kernels(sweep)
    f = golden("f6_semiglobal")
    paths = _paths_from_fixture(f)
    scores, moves, lengths = gpu.semiglobal_xdrop_moves(f["seq1"], f["seq2"])
    assert np.array_equal(scores, f["scores"])
    assert np.array_equal(lengths.astype(np.int64), f["lengths"].astype(np.int64))
    for k in range(len(paths)):
        want = _fixture_moves_words(f, k)
        used = (int(lengths[k]) - 1 + 31) // 32
        assert np.array_equal(moves[k, :used], want[:used]), k          # (words past the last step are unspecified)
        assert np.array_equal(gpu.semiglobal_expand_moves(moves[k], int(lengths[k])), paths[k]), k
        ups, lefts = int(paths[k][-1][0]), int(paths[k][-1][1])         # the best cell = (steps with bit 1, steps with bit 0)
        codes = [(int(moves[k, t >> 5]) >> (2 * (t & 31))) & 3 for t in range(int(lengths[k]) - 1)]
        assert sum(c >> 1 for c in codes) == ups and sum(c & 1 for c in codes) == lefts


@pytest.mark.gpu
def test_gpu_semiglobal_moves_device_entry_and_chunked_host_entry(gpu, oracle):
    """The device-resident form of the moves entry, and the host form over more than one chunk (32768 alignments for this entry)."""
    import torch
    rng = np.random.default_rng(16)
    base_a = rng.integers(0, 4, (5, 16384), dtype=np.uint8)
    base_b = base_a.copy()
    for k in range(5):
        idx = rng.integers(0, 16384, 300 * (k + 1))
        base_b[k, idx] = rng.integers(0, 4, idx.shape, dtype=np.uint8)
    base_b[4, 5000:] = np.roll(base_b[4], -17)[5000:]
    want = [oracle.semiglobal(base_a[k], base_b[k]) for k in range(5)]
    n = 32768 + 77
    pick = rng.integers(0, 5, n)
    a, b = base_a[pick], base_b[pick]
    scores, moves, lengths = gpu.semiglobal_xdrop_moves(a, b)
    assert np.array_equal(scores, np.array([want[k][0] for k in pick], np.int32))
    assert np.array_equal(lengths, np.array([len(want[k][1]) for k in pick], np.uint32))
    for j in list(range(0, n, 811)) + [32767, 32768, n - 1]:
        assert np.array_equal(gpu.semiglobal_expand_moves(moves[j], int(lengths[j])), want[pick[j]][1]), j
    m = 130
    dev = torch.device("cuda", 0)
    d1, d2 = torch.from_numpy(a[:m].copy()).to(dev), torch.from_numpy(b[:m].copy()).to(dev)
    d_scores = torch.empty(m, dtype=torch.int32, device=dev)
    d_len = torch.empty(m, dtype=torch.int32, device=dev)
    d_moves = torch.zeros(m * gpu.SG_MOVE_WORDS, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()
    gpu.semiglobal_xdrop_moves_device(d1.data_ptr(), d2.data_ptr(), m, d_scores.data_ptr(), d_moves.data_ptr(), d_len.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_scores.cpu().numpy(), scores[:m])
    got = d_moves.cpu().numpy().view(np.uint64).reshape(m, gpu.SG_MOVE_WORDS)
    for j in range(0, m, 13):
        used = (int(lengths[j]) - 1 + 31) // 32
        assert np.array_equal(got[j, :used], moves[j, :used]), j


@pytest.mark.gpu
def test_cpp_compat_semiglobal_overloads(gpu, golden, tmp_path):
    """include/swmi_compat.hpp's SemiGlobal_AdaptiveBanded_XDrop_mi355x / swmi::SemiGlobal_mi355x_batch from a plain C++ program
    (g++, no HIP headers): scores, lengths, end cells and a checksum of every traceback against fixture F6."""
    import os
    import shutil
    import subprocess
    from conftest import PKG, ROOT
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    f = golden("f6_semiglobal")
    paths = _paths_from_fixture(f)
    n = len(paths)
    raw = np.stack([f["seq1"], f["seq2"]], axis=1).astype(np.uint8)             # n x 2 x 16384
    data = tmp_path / "alignments.bin"
    raw.tofile(str(data))
    exe = str(tmp_path / "compat_semiglobal")
    lib = os.path.join(PKG, "lib")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "native", "compat_semiglobal.cpp"), "-o", exe, "-L", lib, "-lswmi", "-lpthread",
                            "-Wl,-rpath," + lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    run = subprocess.run([exe, str(data)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert run.returncode == 0, run.stderr
    rows = [tuple(map(int, line.split())) for line in run.stdout.strip().splitlines()]
    assert len(rows) == n
    for k, (score, length, ei, ej, checksum) in enumerate(rows):
        want = 0
        for i, j in paths[k]:
            want = (want * 1000003 + int(i) * 32771 + int(j)) % (1 << 64)
        assert (score, length, ei, ej, checksum) == (int(f["scores"][k]), len(paths[k]), int(paths[k][-1][0]), int(paths[k][-1][1]), want), k


@pytest.mark.gpu
def test_gpu_semiglobal_mapping_survives_a_reinit(gpu):
    """A mapping the program chose through swmi_semiglobal_set_mapping stays across swmi_shutdown / swmi_init (like the scorer's
    schedule); only SWMI_SG_SWEEP in the environment sets it at init (init used to reset it to automatic)."""
    try:
        gpu.semiglobal_set_mapping(22)
        before = gpu.semiglobal_kernels_for_batch(1000)[0]
        gpu.shutdown()
        gpu.init(0)
        assert gpu.semiglobal_kernels_for_batch(1000)[0] == before and "2, 2" in before.replace("<2,2>", "<2, 2>")
    finally:
        gpu.semiglobal_set_mapping(-1)
        gpu.set_schedule(0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [41, 21, 11])
def test_gpu_semiglobal_calm_windows_change_nothing(gpu, oracle, sg_kernels, sweep):
    """The sweeps skip the X-drop test in windows of 8 rounds in which no band cell can reach the threshold (sg_kernels.hip,
    "CALM WINDOWS"); swmi_semiglobal_set_exact(1) sends every window down the exact path.  Same scores, lengths and moves
    either way -- on healthy alignments (nearly every window calm), on alignments that die by the X-drop rule and on a batch
    that mixes them in one wavefront -- and the library's own count says which path ran."""
    import torch
    sg_kernels(sweep)
    rng = np.random.default_rng(4100 + sweep)
    n = 64
    a = rng.integers(0, 4, (n, 16384), dtype=np.uint8)
    b = a.copy()
    subs = rng.random((n, 16384)) < 0.05
    b[subs] = (b[subs] + rng.integers(1, 4, int(subs.sum()), dtype=np.uint8)) & 3       # SpeedtestSemiGlobal's inputs
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()

    def run(x, y, exact):
        m = len(x)
        d1, d2 = torch.from_numpy(x.copy()).to(dev), torch.from_numpy(y.copy()).to(dev)
        d_scores = torch.empty(m, dtype=torch.int32, device=dev)
        d_len = torch.empty(m, dtype=torch.int32, device=dev)
        d_moves = torch.zeros(m * gpu.SG_MOVE_WORDS, dtype=torch.int64, device=dev)
        gpu.semiglobal_set_exact(exact)
        try:
            gpu.semiglobal_xdrop_moves_device(d1.data_ptr(), d2.data_ptr(), m, d_scores.data_ptr(), d_moves.data_ptr(), d_len.data_ptr(), st.cuda_stream)
            windows, calm = gpu.semiglobal_window_stats(st.cuda_stream)
        finally:
            gpu.semiglobal_set_exact(False)
        lengths = d_len.cpu().numpy()
        moves = d_moves.cpu().numpy().reshape(m, gpu.SG_MOVE_WORDS)
        used = [moves[k, : (int(lengths[k]) - 1 + 31) // 32].copy() for k in range(m)]   # words past the last step are unspecified
        for k in range(m):
            tail = (int(lengths[k]) - 1) % 32
            if tail and len(used[k]):
                used[k][-1] &= (1 << (2 * tail)) - 1
        return d_scores.cpu().numpy(), lengths, used, windows, calm

    s0, l0, m0, w0, c0 = run(a, b, False)
    s1, l1, m1, w1, c1 = run(a, b, True)
    assert w0 == w1 > 0 and c1 == 0 and c0 > 0.95 * w0            # healthy alignments: the first windows of every alignment are exact, the rest calm
    assert np.array_equal(s0, s1) and np.array_equal(l0, l1) and all(np.array_equal(x, y) for x, y in zip(m0, m1))
    assert int(l0.min()) > 16384                                  # (full-length paths)
    # alignments that run into the X-drop rule at different places, next to healthy ones in the same wavefront
    a2, bad = a.copy(), b.copy()
    for k in range(0, n, 3):                                      # (unrelated random tails would not do: at +1 / -1 / -1 a banded
        cut = int(rng.integers(200, 16000))                       #  alignment of random DNA still drifts upwards)
        a2[k, cut:], bad[k, cut:] = 0, 1
    for k in range(1, n, 7):                                      # a stretch of mismatches the alignment may or may not survive
        lo = int(rng.integers(1000, 15000))
        a2[k, lo: lo + 20 + 5 * (k % 9)], bad[k, lo: lo + 20 + 5 * (k % 9)] = 2, 3
    s2, l2, m2, w2, c2 = run(a2, bad, False)
    s3, l3, m3, w3, c3 = run(a2, bad, True)
    assert c3 == 0 and 0 < c2 < w2
    assert np.array_equal(s2, s3) and np.array_equal(l2, l3) and all(np.array_equal(x, y) for x, y in zip(m2, m3))
    assert int(l2.min()) < 16384 < int(l2.max())
    # ... and both are the oracle's (pinned to the reference by fixture F6): alignments that die, that scrape past, healthy ones
    for k in list(range(0, 24)) + [n - 1]:
        want_score, want_tb = oracle.semiglobal(a2[k], bad[k])
        assert int(s2[k]) == want_score and int(l2[k]) == len(want_tb), k
        words = np.zeros(gpu.SG_MOVE_WORDS, np.uint64)
        words[: len(m2[k])] = m2[k].view(np.uint64)
        assert np.array_equal(gpu.semiglobal_expand_moves(words, int(l2[k])), want_tb), k


@pytest.mark.gpu
@pytest.mark.parametrize("sweep", [41, 11])
def test_gpu_semiglobal_walks_outside_the_band_centre(gpu, oracle, sg_kernels, sweep):
    """The traceback fetches the predecessor records of band cells 8 .. 23 only and, when a walk of its wavefront leaves those
    cells in a window, decodes the window a second time with the other half (sg_kernels.hip, walk_window).  Relatives with
    5 % substitutions stay in the centre but for the first windows; sequences that start shifted against each other, or are
    rich in insertions and deletions, do not -- and come out as the oracle has them."""
    import torch
    sg_kernels(sweep)
    rng = np.random.default_rng(5100 + sweep)
    n = 64
    a = rng.integers(0, 4, (n, 16384), dtype=np.uint8)
    calm = a.copy()
    subs = rng.random((n, 16384)) < 0.05
    calm[subs] = (calm[subs] + rng.integers(1, 4, int(subs.sum()), dtype=np.uint8)) & 3
    wild = calm.copy()
    for k in range(n):
        if k % 2:                                # the second sequence starts 3 .. 30 bases into the first one, or the other way round
            sh = 3 + (k * 7) % 28
            wild[k] = np.concatenate([calm[k, sh:], rng.integers(0, 4, sh, dtype=np.uint8)]) if k % 4 == 1 else \
                np.concatenate([rng.integers(0, 4, sh, dtype=np.uint8), calm[k, : 16384 - sh]])
        else:                                    # bursts of deletions: the path jumps across the band
            out, i = [], 0
            while len(out) < 16384:
                if i < 16384 and rng.random() < 0.004:
                    i += int(rng.integers(4, 14))
                out.append(a[k, i] if i < 16384 else rng.integers(0, 4))
                i += 1
            wild[k] = out
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()

    def run(x, y):
        m = len(x)
        d1, d2 = torch.from_numpy(x.copy()).to(dev), torch.from_numpy(y.copy()).to(dev)
        d_scores = torch.empty(m, dtype=torch.int32, device=dev)
        d_len = torch.empty(m, dtype=torch.int32, device=dev)
        d_tb = torch.zeros((m, gpu.SG_MAX_TRACEBACK, 2), dtype=torch.int32, device=dev)
        gpu.semiglobal_xdrop_device(d1.data_ptr(), d2.data_ptr(), m, d_scores.data_ptr(), d_tb.data_ptr(), gpu.SG_MAX_TRACEBACK, d_len.data_ptr(), st.cuda_stream)
        stats = gpu.semiglobal_window_stats(st.cuda_stream, walk=True)
        return d_scores.cpu().numpy(), d_len.cpu().numpy(), d_tb.cpu().numpy(), stats
    s0, l0, t0, (_, _, walked0, again0) = run(a, calm)
    assert walked0 > 0 and again0 <= 0.02 * walked0
    s1, l1, t1, (_, _, walked1, again1) = run(a, wild)
    assert again1 > 0.05 * walked1                # the second decoding is exercised ...
    for k in range(n):                            # ... and right
        want_score, want_tb = oracle.semiglobal(a[k], wild[k])
        assert int(s1[k]) == want_score and int(l1[k]) == len(want_tb), k
        assert np.array_equal(t1[k, : len(want_tb)], want_tb), k
    for k in range(0, n, 9):
        want_score, want_tb = oracle.semiglobal(a[k], calm[k])
        assert int(s0[k]) == want_score and np.array_equal(t0[k, : int(l0[k])], want_tb), k


@pytest.mark.gpu
def test_gpu_semiglobal_switches_from_the_environment(gpu):
    """SWMI_SG_SWEEP and SWMI_SG_EXACT give the initial mapping and the exact-only switch of a process (read once, at swmi_init):
    a child process with both set reports the forced build and runs no calm window; one without them does."""
    import os
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smith-waterman-simd_amd")
    code = (
        "import sys; sys.path.insert(0, %r); import numpy as np, torch, swmi\n"
        "swmi.init(0)\n"
        "rng = np.random.default_rng(3); a = rng.integers(0, 4, (4, 16384), dtype=np.uint8)\n"
        "dev = torch.device('cuda', 0); d1 = torch.from_numpy(a).to(dev); d2 = d1.clone()\n"
        "s = torch.empty(4, dtype=torch.int32, device=dev); l = torch.empty(4, dtype=torch.int32, device=dev)\n"
        "m = torch.zeros(4 * swmi.SG_MOVE_WORDS, dtype=torch.int64, device=dev); st = torch.cuda.current_stream().cuda_stream\n"
        "swmi.semiglobal_xdrop_moves_device(d1.data_ptr(), d2.data_ptr(), 4, s.data_ptr(), m.data_ptr(), l.data_ptr(), st)\n"
        "w = swmi.semiglobal_window_stats(st)\n"
        "print(swmi.semiglobal_kernels_for_batch(4)[0]); print(w[0], w[1]); print(s.cpu().tolist())\n" % pkg)

    def run(**env):
        clean = {k: v for k, v in os.environ.items() if not k.startswith("SWMI_")}
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(clean, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        return lines[-3], [int(x) for x in lines[-2].split()], lines[-1]
    kernel, (windows, calm), scores = run()
    assert kernel == "sg_forward_split_kernel<4, 1>" and windows > 0 and calm > 0.9 * windows and scores == "[16384, 16384, 16384, 16384]"
    kernel, (windows2, calm2), scores2 = run(SWMI_SG_SWEEP="22", SWMI_SG_EXACT="1")
    assert kernel == "sg_forward_split_kernel<2, 2>" and windows2 > 0 and calm2 == 0 and scores2 == scores
    kernel, (_, calm3), _ = run(SWMI_SG_SWEEP="13", SWMI_SG_EXACT="7")      # values the setters would reject are ignored
    assert kernel == "sg_forward_split_kernel<4, 1>" and calm3 > 0
