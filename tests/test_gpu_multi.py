"""Multi-GPU behind the C ABI on a one-GPU box: the GPU is bound TWICE (two contexts, two stream sets, the same
hardware), which runs the whole shard / score / gather machinery of swmi_multi.cpp exactly as two GPUs would, and the
result must reproduce the single-context scores and the oracle.  Also: thread safety of the single-GPU entry points."""
import threading

import numpy as np
import pytest
import torch

from conftest import match_matrix

pytestmark = pytest.mark.gpu


# The single-context tests come first: the module-scoped `two` fixture below re-binds the library when the first test
# that asks for it runs, and stays in force until the end of this module.


def test_rccl_gather_with_one_rank(gpu, oracle):
    """SWMI_GATHER_ALL through RCCL itself (librccl, dlopen'd): a single context is a one-rank communicator."""
    assert gpu.num_gpus() == 1
    n = 65536 + 17
    sm = match_matrix(10, -30)
    sb = gpu.ShardedBatch(n)
    sb.generate(1, 0)
    sb.score(sm, 15, gpu.GATHER_ALL)
    sb.wait()
    a, b = oracle.generate(n, 1, 0)
    want = oracle.batch(a, b, sm, 15)
    assert np.array_equal(sb.scores(), want)
    assert np.array_equal(sb.gathered(0), want)
    assert sb.gather_backend() == "rccl"
    sb.close()


def test_rccl_ragged_gather_path_with_one_rank(gpu, oracle):
    """The grouped-ncclBroadcast path ragged shards take (swmi_multi.cpp score_once), rehearsed with the one rank a one-GPU
    box offers: SWMI_TEST_GATHER_PIECE makes every batch take it and cuts the shard into pieces, so that several
    broadcasts with different offsets run inside one group (with G = 1 and whole-shard broadcasts the path would be
    a single call; with equal shards it is never taken at all)."""
    import os
    n = 65536 + 17
    sm = match_matrix(10, -30)
    a, b = oracle.generate(n, 31, 7)
    want = oracle.batch(a, b, sm, 15)
    gpu.shutdown()
    os.environ["SWMI_TEST_GATHER_PIECE"] = "10000"           # knobs are read at swmi_init
    try:
        gpu.init(0)
        sb = gpu.ShardedBatch(n)
        sb.generate(31, 7)
        sb.score(sm, 15, gpu.GATHER_ALL)
        sb.wait()
        assert sb.gather_backend() == "rccl" and sb.gather_note() == ""
        assert np.array_equal(sb.gathered(0), want)
        r = sb.time(sm, 15, gpu.GATHER_ALL, iters=3)          # the timing helper drives the same path
        assert r["wall_ms"] > 0 and np.array_equal(sb.gathered(0), want)
        sb.close()
    finally:
        del os.environ["SWMI_TEST_GATHER_PIECE"]
        gpu.shutdown()
        gpu.init(0)
        gpu.set_schedule(0, 0)


def test_rccl_that_cannot_be_loaded_is_reported_and_peer_copies_take_over(gpu, oracle):
    """SWMI_RCCL_LIB names a file that does not exist: SWMI_GATHER_ALL must still deliver the right vector (peer copies), say
    so through swmi_sharded_gather_backend / _note and leave the reason in swmi_last_error().  A child process: the library
    decides once per process which librccl it uses, and this one has loaded the real one already."""
    import os
    import subprocess
    import sys
    from conftest import PKG
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import torch, swmi
swmi.init(0)
n = 30001
sb = swmi.ShardedBatch(n)
sb.generate(5, 0)
sm = swmi.match_matrix(10, -30)
sb.score(sm, 15, swmi.GATHER_ALL)
print("last_error:", swmi.last_error())
sb.wait()
print("backend:", sb.gather_backend())
print("note:", sb.gather_note())
g = sb.gathered(0)
np.save(sys.argv[1], g)
sb.close()
swmi.shutdown()
""" % PKG
    out_file = os.path.join(os.environ.get("TMPDIR", "/tmp"), "swmi_p2p_fallback_%d.npy" % os.getpid())
    run = subprocess.run([sys.executable, "-c", code, out_file], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=dict(os.environ, SWMI_RCCL_LIB="/nonexistent/librccl-not-here.so"))
    assert run.returncode == 0, run.stderr[-2000:]
    assert "backend: p2p" in run.stdout
    assert "librccl-not-here" in run.stdout.split("note:")[1]
    assert "peer copies" in run.stdout.split("last_error:")[1].splitlines()[0]
    a, b = oracle.generate(30001, 5, 0)
    assert np.array_equal(np.load(out_file), oracle.batch(a, b, match_matrix(10, -30), 15))
    os.remove(out_file)


def test_handles_outlive_shutdown(gpu, oracle):
    """A queue and a sharded batch created before swmi_shutdown(): every later call on them fails with
    SWMI_ERR_NOT_INITIALIZED (also after a new swmi_init), and closing them still works -- a garbage-collected binding
    cannot promise to destroy handles first."""
    sm = match_matrix(10, -30)
    a, b = oracle.generate(100, 3, 0)
    q = gpu.Queue(100, sm, 15)
    q.submit(a[0], b[0])
    sb = gpu.ShardedBatch(1000)
    sb.generate(1, 0)
    sb.score(sm, 15, gpu.GATHER_NONE)
    sb.wait()
    gpu.shutdown()
    gpu.init(0)                                   # new contexts: the old handles must not attach to them
    try:
        for call in (lambda: q.wait(), lambda: q.reset(), lambda: sb.score(sm, 15, gpu.GATHER_NONE),
                     lambda: sb.wait(), lambda: sb.scores(), lambda: sb.generate(1, 0)):
            with pytest.raises(gpu.SwmiError) as e:
                call()
            assert e.value.code == gpu.ERR_NOT_INITIALIZED
        q.close()
        sb.close()
        assert np.array_equal(gpu.score_batch(a, b, sm, 15), oracle.batch(a, b, sm, 15))     # the new contexts work
    finally:
        gpu.set_schedule(0, 0)


def test_two_threads_through_the_c_abi(gpu, oracle):
    """Concurrent callers: two threads score host batches and device batches on their own streams while a third keeps
    changing the schedule -- every result must equal the oracle (the schedule only changes HOW, never WHAT)."""
    sm = match_matrix(10, -30)
    n = 60000
    data = [oracle.generate(n, 100 + k, 0) for k in range(2)]
    want = [oracle.batch(a, b, sm, 15) for a, b in data]
    errors = []
    stop = threading.Event()

    def flipper():
        k = 0
        while not stop.is_set():
            gpu.set_schedule((0, 4, 8, 16, 64)[k % 5], (0, 1)[k % 2])
            k += 1

    def worker(k):
        try:
            a, b = data[k]
            stream = torch.cuda.Stream()
            d1, d2 = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            out = torch.empty(n, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for it in range(12):
                got = gpu.score_batch(a, b, sm, 15)
                if not np.array_equal(got, want[k]):
                    errors.append("thread %d host batch iteration %d" % (k, it))
                out.zero_()
                torch.cuda.synchronize()
                gpu.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), stream.cuda_stream)
                stream.synchronize()
                if not np.array_equal(out.cpu().numpy(), want[k]):
                    errors.append("thread %d device batch iteration %d" % (k, it))
        except Exception as e:      # noqa: BLE001
            errors.append("thread %d: %r" % (k, e))

    f = threading.Thread(target=flipper)
    ws = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    f.start()
    for w in ws:
        w.start()
    for w in ws:
        w.join()
    stop.set()
    f.join()
    gpu.set_schedule(0, 0)
    assert not errors, errors


def test_three_threads_through_the_three_host_entries(gpu, oracle):
    """Round 4's host pipeline keeps a second issuing thread inside the context (the packed entry): three caller threads drive
    the pairs, packed and one-vs-many entries at once, at sizes on both sides of the "more than one granule" switch, while a
    fourth keeps changing the schedule -- calls on one context serialise, every result equals the oracle."""
    sm = match_matrix(10, -30)
    sizes = (300000, 70000, 40000, 1 << 20)
    a, b = oracle.generate(max(sizes), 321, 0)
    want = oracle.batch(a, b, sm, 15)
    want_ovm = oracle.batch(a[:300000], np.broadcast_to(b[0], (300000, 128)).copy(), sm, 15)
    pa, pb = gpu.pack(a), gpu.pack(b)
    errors = []
    stop = threading.Event()

    def flipper():
        k = 0
        while not stop.is_set():
            gpu.set_schedule((0, 4, 8, 16)[k % 4], 0)
            k += 1

    def worker(kind):
        try:
            for it in range(6):
                n = sizes[(it + kind) % len(sizes)]
                if kind == 0:
                    got, ref = gpu.score_batch(a[:n], b[:n], sm, 15), want[:n]
                elif kind == 1:
                    got, ref = gpu.score_batch_packed(pa[:n], pb[:n], sm, 15), want[:n]
                else:
                    n = min(n, 300000)
                    got, ref = gpu.score_one_vs_many(a[:n], b[0], sm, 15), want_ovm[:n]
                if not np.array_equal(got, ref):
                    errors.append("entry %d iteration %d n %d: %d mismatches" % (kind, it, n, int((got != ref).sum())))
        except Exception as e:      # noqa: BLE001
            errors.append("entry %d: %r" % (kind, e))

    f = threading.Thread(target=flipper)
    ws = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    f.start()
    for w in ws:
        w.start()
    for w in ws:
        w.join()
    stop.set()
    f.join()
    gpu.set_schedule(0, 0)
    assert not errors, errors


def test_semiglobal_on_two_streams_in_flight(gpu, oracle, golden):
    """Per-(GPU, stream) workspaces: two semi-global calls on different streams, issued back to back from two threads,
    neither waiting for the other."""
    f = golden("f6_semiglobal")
    a, b = f["seq1"], f["seq2"]
    n = a.shape[0]
    cap = 32769
    results = [None, None]

    def worker(k):
        order = np.arange(n) if k == 0 else np.arange(n)[::-1].copy()
        stream = torch.cuda.Stream()
        d1, d2 = torch.from_numpy(a[order].copy()).cuda(), torch.from_numpy(b[order].copy()).cuda()
        sc = torch.empty(n, dtype=torch.int32, device="cuda")
        ln = torch.empty(n, dtype=torch.int32, device="cuda")
        tb = torch.empty((n, cap, 2), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):
            gpu.semiglobal_xdrop_device(d1.data_ptr(), d2.data_ptr(), n, sc.data_ptr(), tb.data_ptr(), cap, ln.data_ptr(),
                                        stream.cuda_stream)
        stream.synchronize()
        results[k] = (order, sc.cpu().numpy(), ln.cpu().numpy(), tb.cpu().numpy())

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    from test_semiglobal import _paths_from_fixture
    paths = _paths_from_fixture(f)
    for order, sc, ln, tb in results:
        for pos, k in enumerate(order):
            assert sc[pos] == f["scores"][k] and ln[pos] == f["lengths"][k]
            assert np.array_equal(tb[pos, : ln[pos]], paths[k])


@pytest.fixture(scope="module")
def two(gpu):
    """The library re-bound as two contexts on device 0; restored to the session's single context afterwards."""
    gpu.shutdown()
    assert gpu.init_devices([0, 0]) == 2
    assert gpu.num_gpus() == 2
    yield gpu
    gpu.use_gpu(0)
    gpu.shutdown()
    gpu.init(0)
    gpu.set_schedule(0, 0)


def _single_context_scores(gpu, a, b, sm, gap):
    gpu.use_gpu(0)
    return gpu.score_batch(a, b, sm, gap)


@pytest.mark.parametrize("n", [1, 2, 3, 4097, 300001, (1 << 21) + 5])
def test_host_batch_over_two_contexts_matches_single_context_and_oracle(two, oracle, n):
    a, b = oracle.generate(n, 777, 5)
    sm = match_matrix(10, -30)
    got = two.score_batch_multi(a, b, sm, 15)
    assert np.array_equal(got, _single_context_scores(two, a, b, sm, 15))
    m = min(n, 20000)
    assert np.array_equal(got[:m], oracle.batch(a[:m], b[:m], sm, 15))
    assert np.array_equal(got[-m:], oracle.batch(a[-m:], b[-m:], sm, 15))


def test_packed_host_batch_over_two_contexts(two, oracle):
    a, b = oracle.generate(70001, 12, 0)
    sm = match_matrix(1, -1)
    got = two.score_batch_multi(oracle.pack(a), oracle.pack(b), sm, 1, packed=True)
    assert np.array_equal(got, oracle.batch(a, b, sm, 1))


@pytest.mark.parametrize("n", [1, 5, 100000, 100001])
@pytest.mark.parametrize("gather", [0, 1, 2])
def test_resident_shards_all_gather_modes(two, oracle, n, gather):
    sm = match_matrix(10, -30)
    sb = two.ShardedBatch(n)
    sb.generate(4242, 100)                    # each GPU generates its own shard from the global pair index
    sb.score(sm, 15, gather)
    sb.wait()
    a, b = oracle.generate(n, 4242, 100)
    want = oracle.batch(a, b, sm, 15)
    assert np.array_equal(sb.scores(), want)
    for index in range(2 if gather == two.GATHER_ALL else 1 if gather == two.GATHER_ROOT else 0):
        assert np.array_equal(sb.gathered(index), want), "gathered vector on GPU index %d" % index
    sb.close()


def test_a_gpu_bound_twice_gathers_over_peer_copies_and_says_why(two, oracle):
    n = 50000
    sb = two.ShardedBatch(n)
    sb.generate(8, 0)
    assert sb.gather_backend() == "undecided"
    sb.score(match_matrix(10, -30), 15, two.GATHER_ALL)
    assert "peer copies" in two.last_error() and "bound twice" in two.last_error()
    sb.wait()
    assert sb.gather_backend() == "p2p" and "bound twice" in sb.gather_note()
    a, b = oracle.generate(n, 8, 0)
    want = oracle.batch(a, b, match_matrix(10, -30), 15)
    assert np.array_equal(sb.gathered(0), want) and np.array_equal(sb.gathered(1), want)
    sb.close()


def test_resident_shards_upload_and_timing_helper(two, oracle):
    n = 200003
    a, b = oracle.generate(n, 9, 0)
    sm = match_matrix(2, -3)
    sb = two.ShardedBatch(n)
    sb.upload(a, b)
    r = sb.time(sm, 5, two.GATHER_ROOT, iters=4)
    assert len(r["kernel_ms"]) == 2 and all(k > 0 for k in r["kernel_ms"]) and r["wall_ms"] > 0
    assert all(g >= 0 for g in r["gather_ms"])
    assert np.array_equal(sb.scores(), oracle.batch(a, b, sm, 5))
    sb.close()
