"""Kernel time of the headline scorer against the batch size: how much of a 1 M-pair launch is ramp, tail and the reload between
the two batches of wavefronts a SIMD runs (4 resident, 8 per SIMD at 1 M pairs)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "smith-waterman-simd_amd"))
import swmi
swmi.init(0)
sm = swmi.match_matrix(10, -30)
dev = torch.device("cuda", 0)
nmax = 1 << 24
d1 = torch.empty(nmax * 128, dtype=torch.uint8, device=dev); d2 = torch.empty_like(d1)
out = torch.empty(nmax, dtype=torch.int32, device=dev)
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), nmax, 10000, 0)
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for n in (131072, 262144, 393216, 524288, 655360, 786432, 1048576, 1572864, 2097152, 4194304, 16777216):
    swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, 5)
    ms = min(swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, 40) for _ in range(3))
    print("n %9d: %8.4f ms per launch, %7.2f us per 1 M pairs, wavefronts per SIMD %d" % (n, ms, ms * 1e3 * 1048576 / n, n // 32 // 1024), flush=True)
