#!/usr/bin/env python3
"""Kernel time vs batch size and schedule (developer tool): shows launch/tail overheads and the per-pair cost."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smith-waterman-simd_amd"))
import swmi, torch
swmi.init(0)
st = torch.cuda.current_stream().cuda_stream
sm = swmi.match_matrix(10, -30)
nmax = 1 << 26
d1 = torch.empty(nmax * 128, dtype=torch.uint8, device="cuda")
d2 = torch.empty(nmax * 128, dtype=torch.uint8, device="cuda")
out = torch.empty(nmax, dtype=torch.int32, device="cuda")
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), nmax, 10000, 0, st)
torch.cuda.synchronize()
for L in [int(x) for x in os.environ.get("LANES", "8,4,2").split(",")]:
    swmi.set_schedule(L, 0)
    for logn in (10, 14, 17, 19, 20, 21, 22, 23, 26):
        n = 1 << logn
        iters = max(2, min(50, (1 << 26) // n))
        swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=2)
        ms = swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=iters)
        print("L=%d n=2^%d  %.4f ms  %.1f M align/s  %.2f TCUPS  (%.3f ns/pair)" % (L, logn, ms, n / ms / 1e3, n * 16384 / ms / 1e9, ms * 1e6 / n), flush=True)
