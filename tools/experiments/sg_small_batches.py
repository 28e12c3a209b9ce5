"""The split sweep (41, 21) and what the library picks for the smallest batches: ms per call, device inputs.
(Run on commit 662c169.. with `for sweep in (0, 41, 21, -1)` it produced profiles/r03_sg_small_batches.txt: the half-wavefront
sweep, mapping 0, lost at every size and was removed.)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "smith-waterman-simd_amd"))
import swmi
swmi.init(0)
L, cap = 16384, swmi.SG_MAX_TRACEBACK
dev = torch.device("cuda", 0)
for n in (1, 2, 16, 64, 256, 1024, 2048):
    g = torch.Generator(device=dev); g.manual_seed(1)
    d1 = torch.randint(0, 4, (n, L), dtype=torch.uint8, device=dev, generator=g)
    rnd = torch.randint(0, 4, (n, L), dtype=torch.uint8, device=dev, generator=g)
    d2 = torch.where(torch.rand((n, L), device=dev, generator=g) < 0.95, d1, rnd).contiguous()
    scores = torch.empty(n, dtype=torch.int32, device=dev); lengths = torch.empty(n, dtype=torch.int32, device=dev)
    tb = torch.empty((n, cap, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for sweep in (41, 21, -1):
        swmi.semiglobal_set_mapping(sweep)
        swmi.semiglobal_time_device(d1.data_ptr(), d2.data_ptr(), n, scores.data_ptr(), tb.data_ptr(), cap, lengths.data_ptr(), st)
        a, b = swmi.semiglobal_time_device(d1.data_ptr(), d2.data_ptr(), n, scores.data_ptr(), tb.data_ptr(), cap, lengths.data_ptr(), st)
        out.append("%s %.2f + %.2f ms (checksum %d)" % (sweep, a, b, int(scores.sum().item()) + int(lengths.sum().item())))
    print("n %5d:" % n, " | ".join(out), flush=True)
