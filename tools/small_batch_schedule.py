"""Kernel time of the 128x128 scorer for small batches under every lanes-per-alignment schedule: which L for which n.
Usage (GPU box): python tools/small_batch_schedule.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi  # noqa: E402

swmi.init(0)
sm = swmi.match_matrix(10, -30)
nmax = 1 << 18
d1 = torch.empty(nmax * 128, dtype=torch.uint8, device="cuda")
d2 = torch.empty(nmax * 128, dtype=torch.uint8, device="cuda")
out = torch.empty(nmax, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), nmax, 10000, 0, st)
torch.cuda.synchronize()
print("%8s " % "n" + " ".join("%9s" % ("L=%d" % L) for L in (64, 32, 16, 8, 4, 2)) + "   (kernel microseconds, HIP events over 50 launches)")
for n in (1, 16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144):
    row = []
    for L in (64, 32, 16, 8, 4, 2):
        swmi.set_schedule(L, 0)
        swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=5)
        row.append(swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=50) * 1e3)
    best = min(row)
    print("%8d " % n + " ".join(("%8.1f%s" % (v, "*" if v == best else " ")) for v in row), flush=True)
