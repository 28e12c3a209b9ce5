"""Host-buffer entry point: where do the bytes come from?  (include/swmi.h swmi_score_batch)

swmi_score_batch issues its H2D copies straight from the caller's memory.  This measures, for 1M and 4M distinct pairs:
  pageable        caller's arrays are ordinary (pageable) memory -- what a reference-style caller has
  pinned          caller's arrays are already page-locked (hipHostMalloc / torch pin_memory): the link's own limit
  stage 1 thread  what library-owned pinned staging would cost: one host memcpy pageable -> pinned, then `pinned`
  stage N threads the same memcpy split over N threads (N = the container's CPU quota)
Usage (GPU box): python tools/host_staging_experiment.py
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import swmi  # noqa: E402
from bench import effective_cores  # noqa: E402

swmi.init(0)
sm = swmi.match_matrix(10, -30)
threads = max(1, int(effective_cores() or 1))
lib = swmi.load()


def score(p1, p2, n, out):
    rc = lib.swmi_score_batch(p1, p2, n, sm.ctypes.data, 15, out.ctypes.data)
    assert rc == 0, swmi.last_error()


def best_of(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


for n in (1 << 20, 1 << 22):
    a, b = swmi.generate_pairs_host(n, 10000, 0)
    out = np.zeros(n, np.int32)
    pa = torch.empty(n * 128, dtype=torch.uint8).pin_memory()
    pb = torch.empty(n * 128, dtype=torch.uint8).pin_memory()
    na, nb = pa.numpy(), pb.numpy()
    na[:] = a.reshape(-1)
    nb[:] = b.reshape(-1)
    ref = swmi.score_batch(a, b, sm, 15)

    t_page = best_of(lambda: score(a.ctypes.data, b.ctypes.data, n, out))
    assert np.array_equal(out, ref)
    t_pin = best_of(lambda: score(pa.data_ptr(), pb.data_ptr(), n, out))
    assert np.array_equal(out, ref)

    def stage_one():
        na[:] = a.reshape(-1)
        nb[:] = b.reshape(-1)
        score(pa.data_ptr(), pb.data_ptr(), n, out)
    t_stage1 = best_of(stage_one)

    fa, fb = a.reshape(-1), b.reshape(-1)
    cuts = [(len(fa) * k // threads, len(fa) * (k + 1) // threads) for k in range(threads)]
    pool = ThreadPoolExecutor(threads)

    def copy_slice(c):
        na[c[0]:c[1]] = fa[c[0]:c[1]]
        nb[c[0]:c[1]] = fb[c[0]:c[1]]

    def stage_many():
        list(pool.map(copy_slice, cuts))
        score(pa.data_ptr(), pb.data_ptr(), n, out)
    t_stagen = best_of(stage_many)
    pool.shutdown()
    gb = n * 260 / 1e9
    for name, t in (("pageable", t_page), ("pinned", t_pin), ("stage 1 thread", t_stage1), ("stage %d threads" % threads, t_stagen)):
        print("n %8d  %-18s %8.2f ms  %7.1f M alignments/s  %6.1f GB/s over PCIe" % (n, name, t * 1e3, n / t / 1e6, gb / t), flush=True)
