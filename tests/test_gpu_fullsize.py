"""BASELINE.json's full sizes (1M pairs = configs[1]) on the GPU: the whole batch against the multi-threaded
CPU oracle, plus size-independent properties of the recurrence that need no oracle at all."""
import numpy as np
import pytest
import torch

from conftest import match_matrix

pytestmark = pytest.mark.gpu

N = 1 << 20


@pytest.fixture(scope="module")
def resident(gpu):
    d1 = torch.empty(N * 128, dtype=torch.uint8, device="cuda")
    d2 = torch.empty(N * 128, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    gpu.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), N, 10000, 0, st)
    torch.cuda.synchronize()
    return d1, d2


def _score(gpu, d1, d2, sm, gap, n=N):
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    gpu.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, gap, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


def test_one_million_pairs_bit_exact_vs_oracle(gpu, oracle, resident):
    d1, d2 = resident
    sm = match_matrix(10, -30)      # SpeedTest parameters, source.cpp:3041-3046
    got = _score(gpu, d1, d2, sm, 15).cpu().numpy()
    a, b = oracle.generate(N, 10000, 0)
    want = oracle.batch(a, b, sm, 15)          # OpenMP over the host cores
    assert np.array_equal(got, want)
    assert 30 <= got.min() and got.max() <= 400     # random DNA under (10,-30,15) clusters around 70-80
    assert int(got.astype(np.int64).sum()) == 79139805   # checksum of this generated batch (bench.py prints the same)


def test_all_schedules_agree_on_one_million_pairs(gpu, resident):
    d1, d2 = resident
    sm = match_matrix(1, -1)        # speedtest111x32 parameters, source.cpp:3202-3207
    ref = None
    try:
        for lanes, flags in ((4, 0), (64, 0), (32, 1), (16, 2), (4, 3), (8, 4), (8, 1), (2, 0)):
            gpu.set_schedule(lanes, flags)
            s = _score(gpu, d1, d2, sm, 1)
            if ref is None:
                ref = s
            assert torch.equal(s, ref), (lanes, flags)
    finally:
        gpu.set_schedule(0, 0)


def test_symmetry_and_scaling_properties(gpu, resident):
    d1, d2 = resident
    rng = np.random.default_rng(2)
    sm = rng.integers(-12, 13, 16).astype(np.int8)
    smT = sm.reshape(4, 4).T.copy().reshape(16)
    s_ab = _score(gpu, d1, d2, sm, 4)
    s_ba = _score(gpu, d2, d1, smT, 4)          # swapping the sequences and transposing the matrix
    assert torch.equal(s_ab, s_ba)
    s_x3 = _score(gpu, d1, d2, (sm * 3).astype(np.int8), 12)   # the recurrence is homogeneous of degree 1
    assert torch.equal(s_x3, s_ab * 3)
    assert torch.equal(_score(gpu, d1, d2, sm, 4), s_ab)       # deterministic


def test_self_alignment_and_position_independence(gpu, resident):
    d1, _ = resident
    sm = match_matrix(10, -30)
    s = _score(gpu, d1, d1, sm, 15)
    assert int(s.min()) == 1280 and int(s.max()) == 1280       # identical pair -> 128 * match
    # a pair's score does not depend on where it sits in the batch (ragged offset into the resident buffer)
    full = _score(gpu, resident[0], resident[1], sm, 15)
    off = 12345
    part = torch.empty(5000, dtype=torch.int32, device="cuda")
    gpu.score_batch_device(resident[0].data_ptr() + off * 128, resident[1].data_ptr() + off * 128, 5000, sm, 15,
                           part.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(part, full[off:off + 5000])


def test_monotone_in_gap_and_bounded(gpu, resident):
    d1, d2 = resident
    sm = match_matrix(5, -4)
    prev = None
    for gap in (0, 1, 3, 10, 127):
        s = _score(gpu, d1, d2, sm, gap)
        assert int(s.min()) >= 0 and int(s.max()) <= 128 * 5
        if prev is not None:
            assert bool((s <= prev).all())      # a larger gap penalty can never raise a score
        prev = s


def test_host_batch_pipeline_across_chunks(gpu, oracle):
    """swmi_score_batch stages host buffers in 1M-pair chunks on two alternating streams: cover > 2 chunks + ragged end."""
    n = (1 << 21) + 12345
    a, b = gpu.generate_pairs_host(n, 4242, 7)
    sm = match_matrix(2, -3)
    got = gpu.score_batch(a, b, sm, 5)
    want = oracle.batch(a, b, sm, 5)
    assert np.array_equal(got, want)


def test_sixty_four_million_pairs_checksum(gpu, resident):
    """BASELINE config 3 size (64M pairs, 16 GiB of inputs) on one GPU: every 1M-pair block of the big launch must
    reproduce the scores of that block scored on its own (position independence at full size)."""
    n = 1 << 26
    d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
    d2 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    gpu.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 10000, 0, st)
    sm = match_matrix(10, -30)
    big = _score(gpu, d1, d2, sm, 15, n)
    assert int(big[:N].to(torch.int64).sum().item()) == 79139805
    for blk in (0, 17, 63):
        part = torch.empty(N, dtype=torch.int32, device="cuda")
        gpu.score_batch_device(d1.data_ptr() + blk * N * 128, d2.data_ptr() + blk * N * 128, N, sm, 15, part.data_ptr(), st)
        torch.cuda.synchronize()
        assert torch.equal(part, big[blk * N:(blk + 1) * N])
    assert int(big.min()) >= 30 and int(big.max()) <= 600


def test_host_batch_above_one_production_score_group(gpu, oracle):
    """score_host_batch's several-group branch at the PRODUCTION group size (2^24 pairs = 64 MiB of scores): one host batch of
    2^24 + 70 001 pairs (2 x 2.15 GB of host arrays) -- every score against the same pairs scored resident in one launch, and
    the oracle on samples at the head, either side of the group boundary and the ragged tail.  The packed entry takes the same
    path with its own schedule."""
    n = (1 << 24) + 70001
    sm = match_matrix(10, -30)
    d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
    d2 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
    gpu.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 4711, 0, torch.cuda.current_stream().cuda_stream)
    want = _score(gpu, d1, d2, sm, 15, n).cpu().numpy()
    a = d1.cpu().numpy().reshape(n, 128)
    b = d2.cpu().numpy().reshape(n, 128)
    del d1, d2
    torch.cuda.empty_cache()
    g = gpu.host_granules(n)
    assert sum(g) == n and (1 << 24) in np.cumsum(g)          # two groups: a granule boundary at 2^24, then the tail's own schedule
    got = gpu.score_batch(a, b, sm, 15)
    assert np.array_equal(got, want)
    for lo in (0, (1 << 24) - 600, (1 << 24), n - 1200):
        hi = min(n, lo + 1200)
        assert np.array_equal(got[lo:hi], oracle.batch(a[lo:hi], b[lo:hi], sm, 15)), lo
    got_packed = gpu.score_batch_packed(gpu.pack(a), gpu.pack(b), sm, 15)
    assert np.array_equal(got_packed, want)
