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
