"""Kernel time of the 128x128 scorer for small batches under every lanes-per-alignment schedule: which L for which n.
Usage (GPU box): python tools/small_batch_schedule.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi  # noqa: E402

swmi.init(0)
sm = swmi.match_matrix(10, -30)
nmax = 1 << 20
d1 = torch.empty(nmax * 128, dtype=torch.uint8, device="cuda")
d2 = torch.empty(nmax * 128, dtype=torch.uint8, device="cuda")
out = torch.empty(nmax, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), nmax, 10000, 0, st)
torch.cuda.synchronize()
COLS = [(64, 0), (32, 0), (16, swmi.NO_PACKED), (8, swmi.NO_PACKED), (4, swmi.NO_PACKED), (16, 0), (8, 0), (4, 0), (2, 0)]
NAMES = ["L=64", "L=32", "L=16 i32", "L=8 i32", "L=4 i32", "L=16 pk", "L=8 pk", "L=4 pk", "L=2"]
print("%8s " % "n" + " ".join("%9s" % c for c in NAMES) + "   (kernel microseconds, HIP events over 50 launches; i32 = schedule flag 8,"
      " pk = the packed kernel; every column's scores equal the first column's)")
ref = torch.empty(nmax, dtype=torch.int32, device="cuda")
for n in (1, 16, 64, 256, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    row = []
    for k, (L, flags) in enumerate(COLS):
        swmi.set_schedule(L, flags)
        o = ref if k == 0 else out
        swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, o.data_ptr(), st, iters=5)
        row.append(swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, o.data_ptr(), st, iters=50) * 1e3)
        if k:
            assert torch.equal(out[:n], ref[:n]), (n, L, flags)
    best = min(row)
    print("%8d " % n + " ".join(("%8.1f%s" % (v, "*" if v == best else " ")) for v in row), flush=True)
