"""Quick A/B of the packed kernel against the int32 kernel on the GPU box: same scores, time per launch of 1M pairs.
Usage: python tools/quick_pk_timing.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi
swmi.init(0)
n = 1 << 20
REPS = int(os.environ.get("QUICK_REPS", "30"))
ONLY = os.environ.get("QUICK_ONLY")
d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda"); d2 = torch.empty_like(d1)
out = torch.empty(n, dtype=torch.int32, device="cuda"); ref = torch.empty_like(out)
st = torch.cuda.current_stream().cuda_stream
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 10000, 0, st)
m = torch.rand(n * 128, device="cuda") < 0.85
half = (n // 2) * 128
d2[:half] = torch.where(m[:half], d1[:half], d2[:half])        # half of the batch: related pairs (long alignments)
import numpy as np
rng = np.random.default_rng(5)
cases = [("10/-30/15", swmi.match_matrix(10, -30), 15), ("1/-1/1", swmi.match_matrix(1, -1), 1),
         ("random", rng.integers(-128, 128, 16).astype(np.int8), 77), ("random small", rng.integers(-12, 13, 16).astype(np.int8), 3),
         ("127/-128/127", swmi.match_matrix(127, -128), 127), ("127/-128/0", swmi.match_matrix(127, -128), 0)]
for name, sm, gap in cases:
    if ONLY and name not in ONLY.split(','): continue
    res = {}
    for flags in (swmi.NO_PACKED, 0):
        swmi.set_schedule(4, flags)
        o = ref if flags else out
        for _ in range(3): swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, gap, o.data_ptr(), st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(REPS): swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, gap, o.data_ptr(), st)
        torch.cuda.synchronize(); res[flags] = (time.perf_counter() - t0) / REPS
    bad = int((out != ref).sum())
    print("%-14s %-22s int32 %.4f ms   packed %.4f ms = %6.1f M/s   mismatches %d   max score %d" % (
        name, swmi.score_kernel_for_batch(n, sm, gap)[0], res[swmi.NO_PACKED] * 1e3, res[0] * 1e3, n / res[0] / 1e6, bad, int(ref.max())), flush=True)
