"""Every entry point of include/swmi.h on the GPU: per-pair, queue, one-vs-many, packed, unpack,
device-pointer batch, generator, timing helper, error paths."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import match_matrix

pytestmark = pytest.mark.gpu


def test_device_is_gfx950_and_native_library_is_loaded(gpu):
    info = gpu.device_info()
    assert info["arch"].startswith("gfx950") and info["wavefront_size"] == 64 and info["compute_units"] >= 32
    maps = open("/proc/self/maps").read()
    assert "libswmi.so" in maps


def test_per_pair_signature(gpu, oracle, golden):
    f = golden("f3_harness")
    sm = match_matrix(10, -30)
    # the SpeedTest pair (source.cpp:3033-3046) scores 80, and 18 with the speedtest111x32 parameters
    assert gpu.score_pair(f["seq1"][0], f["seq2"][0], sm, 15) == 80
    assert gpu.score_pair(f["seq1"][0], f["seq2"][0], match_matrix(1, -1), 1) == 18
    for k in range(1, 40):
        assert gpu.score_pair(f["seq1"][k], f["seq2"][k], sm, 15) == int(f["scores"][0][k])


def test_queue_behind_the_per_pair_signature(gpu, oracle):
    n = 150001                      # several asynchronous shipments plus a ragged remainder
    a, b = oracle.generate(n, 31337, 0)
    sm = match_matrix(10, -30)
    q = gpu.Queue(n, sm, 15)
    lib = gpu.load()
    # submit through the raw ABI to keep the loop fast
    for k in range(n):
        t = lib.swmi_queue_submit(q._q, a[k].ctypes.data, b[k].ctypes.data)
        assert t == k
    got = q.wait()
    assert np.array_equal(got, oracle.batch(a, b, sm, 15))
    with pytest.raises(gpu.SwmiError) as e:
        q.submit(a[0], b[0])
    assert e.value.code == gpu.ERR_QUEUE_FULL
    q.reset()
    assert q.submit(a[5], b[5]) == 0
    assert list(q.wait()) == [oracle.score(a[5], b[5], sm, 15)]
    q.close()


def test_one_vs_many(gpu, oracle, golden):
    f = golden("f5_siblings")       # SmithWaterman_8b111x32mark1/2/3, source.cpp:1227-1522: 32 seq1 x one seq2
    sm = match_matrix(1, -1)
    for blk in range(f["scores_111x32"].shape[0]):
        got = gpu.score_one_vs_many(f["seq1"][32 * blk:32 * blk + 32], f["seq2"][blk], sm, 1)
        assert np.array_equal(got, f["scores_111x32"][blk])
    a, b = oracle.generate(5000, 3, 0)
    sm2 = match_matrix(5, -4)
    want = oracle.batch(a, np.repeat(b[:1], 5000, axis=0), sm2, 2)
    assert np.array_equal(gpu.score_one_vs_many(a, b[0], sm2, 2), want)
    d1 = torch.from_numpy(a).cuda()
    d2 = torch.from_numpy(b[0].copy()).cuda()
    out = torch.empty(5000, dtype=torch.int32, device="cuda")
    gpu.score_one_vs_many_device(d1.data_ptr(), 5000, d2.data_ptr(), sm2, 2, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def test_fixed_111_scorer_is_the_general_one(gpu, golden):
    f = golden("f5_siblings")       # SmithWaterman_111 / _8bit111simd, source.cpp:1073-1225
    assert np.array_equal(gpu.score_batch(f["seq1"], f["seq2"], match_matrix(1, -1), 1), f["scores_111"])


def test_unpack_and_packed_scoring(gpu, oracle, golden):
    f = golden("f5_siblings")       # unpack(), source.cpp:1580-1583
    assert np.array_equal(gpu.unpack(f["packed"]), f["unpacked"])
    a, b = oracle.generate(10007, 99, 0)
    pa, pb = oracle.pack(a), oracle.pack(b)
    sm = match_matrix(10, -30)
    want = oracle.batch(a, b, sm, 15)
    for lanes in (64, 32, 16, 8, 4, 2):
        gpu.set_schedule(lanes, 0)
        try:
            assert np.array_equal(gpu.score_batch_packed(pa, pb, sm, 15), want), lanes
        finally:
            gpu.set_schedule(0, 0)


def test_device_pointer_batch_and_generator(gpu, oracle):
    n = 20000
    d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
    d2 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
    out = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    gpu.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 10000, 123456789, st)
    sm = match_matrix(10, -30)
    gpu.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = oracle.generate(n, 10000, 123456789)
    assert np.array_equal(d1.cpu().numpy().reshape(n, 128), a)
    assert np.array_equal(d2.cpu().numpy().reshape(n, 128), b)
    assert np.array_equal(out.cpu().numpy(), oracle.batch(a, b, sm, 15))
    ms = gpu.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=3)
    assert 0 < ms < 1000


def test_error_paths_on_device(gpu):
    sm = match_matrix(1, -1)
    d = torch.empty(4096, dtype=torch.uint8, device="cuda")
    o = torch.empty(16, dtype=torch.int32, device="cuda")
    with pytest.raises(gpu.SwmiError) as e:
        gpu.score_batch_device(d.data_ptr() + 1, d.data_ptr(), 4, sm, 1, o.data_ptr())
    assert e.value.code == gpu.ERR_ALIGNMENT
    with pytest.raises(gpu.SwmiError) as e:
        gpu.score_batch(np.zeros((1, 128), np.uint8), np.zeros((1, 128), np.uint8), sm, -3)
    assert e.value.code == gpu.ERR_DOMAIN
    with pytest.raises(gpu.SwmiError) as e:
        gpu.set_schedule(5, 0)
    assert e.value.code == gpu.ERR_INVALID_ARGUMENT


def test_cpp_compat_header_and_pair_queue(gpu, golden, tmp_path):
    """include/swmi_compat.hpp from a plain C++ program (g++, no HIP headers): the reference-shaped overload and PairQueue."""
    import os
    import shutil
    import subprocess
    from conftest import PKG, ROOT
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    f = golden("f3_harness")
    n = 300
    raw = np.stack([f["seq1"][:n], f["seq2"][:n]], axis=1).astype(np.uint8)      # n x 2 x 128
    data = tmp_path / "pairs.bin"
    raw.tofile(str(data))
    exe = str(tmp_path / "compat_check")
    lib = os.path.join(PKG, "lib")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "native", "compat_check.cpp"), "-o", exe, "-L", lib, "-lswmi",
                            "-Wl,-rpath," + lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout
    run = subprocess.run([exe, str(data), "10", "-30", "15"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert run.returncode == 0, run.stderr
    rows = [tuple(map(int, line.split())) for line in run.stdout.strip().splitlines()]
    assert len(rows) == n
    assert [r[0] for r in rows] == list(f["scores"][0][:n])
    assert [r[1] for r in rows] == list(f["scores"][0][:n])
    assert [r[2] for r in rows] == list(f["scores"][0][:n])          # the whole-array overload (and, inside the program, its two-context split)


def test_automatic_schedule_follows_the_batch_size(gpu, oracle):
    """lanes = 0 (default): many lanes per alignment for small batches, L = 4 for large ones; same scores either way."""
    gpu.set_schedule(0, 0)
    assert gpu.get_schedule() == (0, 0)
    assert [gpu.schedule_for_batch(n) for n in (1, 2048, 2049, 5120, 5121, 24576, 24577, 98304, 98305, 1 << 20)] == [64, 64, 32, 32, 16, 16, 8, 8, 4, 4]
    gpu.set_schedule(8, 0)
    assert gpu.schedule_for_batch(1) == 8 and gpu.schedule_for_batch(1 << 20) == 8
    gpu.set_schedule(0, 0)
    sm = match_matrix(10, -30)
    for n in (1, 17, 2048, 2049, 5121, 24577, 98305):                  # each side of every threshold, ragged sizes included
        a, b = oracle.generate(n, 4242, 77)
        assert np.array_equal(gpu.score_batch(a, b, sm, 15), oracle.batch(a, b, sm, 15)), n


@pytest.mark.parametrize("threads", [2, 1])
def test_host_batch_over_several_score_groups(gpu, oracle, threads):
    """score_host_batch's several-group branch (swmi_api.cpp): a host batch above the score group (16M pairs in production)
    returns every group's scores in one copy at the group's host offset and drains that copy before the next group's kernels
    overwrite the device score vector.  SWMI_TEST_SCORE_GROUP makes a group 32K pairs so that five groups + a ragged tail fit
    a test; all three host entries, with two issuing threads (the default) and with one (round 3's pipeline)."""
    import os
    n = 5 * 32768 + 4321
    sm = match_matrix(10, -30)
    a, b = oracle.generate(n, 777, 5)
    want = oracle.batch(a, b, sm, 15)
    want_ovm = oracle.batch(a, np.broadcast_to(b[0], a.shape).copy(), sm, 15)
    gpu.shutdown()
    os.environ["SWMI_TEST_SCORE_GROUP"] = "32768"            # knobs are read at swmi_init
    os.environ["SWMI_HOST_THREADS"] = str(threads)
    try:
        gpu.init(0)
        for entry in (gpu.ENTRY_PAIRS, gpu.ENTRY_PACKED, gpu.ENTRY_ONE_VS_MANY):
            g = gpu.host_granules(n, entry)
            assert sum(g) == n and max(g) <= 32768 and len(g) >= 6
        for _ in range(2):                                     # the second call reuses every buffer
            assert np.array_equal(gpu.score_batch(a, b, sm, 15), want)
            assert np.array_equal(gpu.score_batch_packed(gpu.pack(a), gpu.pack(b), sm, 15), want)
            assert np.array_equal(gpu.score_one_vs_many(a, b[0], sm, 15), want_ovm)
        pinned_a, pinned_b = torch.from_numpy(a).pin_memory(), torch.from_numpy(b).pin_memory()
        out = gpu.score_batch(pinned_a.numpy(), pinned_b.numpy(), sm, 15)      # pinned input: no copy blocks the issuing threads
        assert np.array_equal(out, want)
    finally:
        del os.environ["SWMI_TEST_SCORE_GROUP"]
        del os.environ["SWMI_HOST_THREADS"]
        gpu.shutdown()
        gpu.init(0)
        gpu.set_schedule(0, 0)


def test_host_batch_entries_agree_with_one_and_two_issuing_threads(gpu, oracle):
    """The per-entry granule schedules (256 / 64 / 128 bytes per pair over the link) and the two-thread issue order give the
    scores of the oracle at a size with many granules, ragged."""
    n = (1 << 20) + 777
    sm = match_matrix(10, -30)
    a, b = oracle.generate(n, 99, 3)
    want = oracle.batch(a[:70000], b[:70000], sm, 15)
    got = gpu.score_batch(a, b, sm, 15)
    got_packed = gpu.score_batch_packed(gpu.pack(a), gpu.pack(b), sm, 15)
    assert np.array_equal(got[:70000], want) and np.array_equal(got, got_packed)
    tail = oracle.batch(a[-5000:], b[-5000:], sm, 15)
    assert np.array_equal(got[-5000:], tail)
    ovm = gpu.score_one_vs_many(a, b[0], sm, 15)
    assert np.array_equal(ovm[:20000], oracle.batch(a[:20000], np.broadcast_to(b[0], (20000, 128)).copy(), sm, 15))
    assert np.array_equal(ovm[-3000:], oracle.batch(a[-3000:], np.broadcast_to(b[0], (3000, 128)).copy(), sm, 15))
