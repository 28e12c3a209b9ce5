"""Times the semi-global kernels (sweep mapping x traceback mapping) over batch sizes: which mapping for which batch.
Usage (GPU box): python tools/sg_sweep_matrix.py [sizes...]   -> one line per (n, sweep, traceback)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi  # noqa: E402

swmi.init(0)
sizes = [int(x) for x in sys.argv[1:]] or [4096, 16384, 32768, 65536, 131072, 262144]
L, cap = 16384, swmi.SG_MAX_TRACEBACK
dev = torch.device("cuda", 0)
for n in sizes:
    g = torch.Generator(device=dev); g.manual_seed(1)
    d1 = torch.randint(0, 4, (n, L), dtype=torch.uint8, device=dev, generator=g)
    rnd = torch.randint(0, 4, (n, L), dtype=torch.uint8, device=dev, generator=g)
    keep = torch.rand((n, L), device=dev, generator=g) < 0.95
    d2 = torch.where(keep, d1, rnd).contiguous()
    del rnd, keep
    if os.environ.get("SG_MATRIX_SHIFTED"):              # every second alignment starts 3 .. 30 bases into the other sequence: paths
        for sh in range(3, 31):                           # off the band's centre (the traceback's second decoding)
            rows = torch.arange(sh - 3, n, 56, device=dev)
            d2[rows] = torch.roll(d2[rows], -sh, dims=1)
    scores = torch.empty(n, dtype=torch.int32, device=dev)
    lengths = torch.empty(n, dtype=torch.int32, device=dev)
    tb = torch.empty((n, cap, 2), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ref = None
    variants = (-1, 4, 2, 1)                              # -1 = what the library picks for this batch size
    if os.environ.get("SG_MATRIX_SWEEPS"):               # e.g. "41,42,44,22,24,11": 10 * lanes + scheduling target
        variants = tuple(int(v) for v in os.environ["SG_MATRIX_SWEEPS"].split(","))
    for sweep in variants:
        swmi.semiglobal_set_mapping(sweep)
        swmi.semiglobal_time_device(d1.data_ptr(), d2.data_ptr(), n, scores.data_ptr(), tb.data_ptr(), cap, lengths.data_ptr(), st)
        a, b = swmi.semiglobal_time_device(d1.data_ptr(), d2.data_ptr(), n, scores.data_ptr(), tb.data_ptr(), cap, lengths.data_ptr(), st)
        chk = (int(scores.sum().item()), int(lengths.sum().item()), int(tb[:: max(1, n // 64), :4096].sum().item()))
        ref = ref or chk
        w = swmi.semiglobal_window_stats(st, walk=True)
        print("n %7d sweep %2d: sweep %8.2f ms traceback %8.2f ms  -> %8.1f k alignments/s %s  (calm %.3f, walked twice %.3f)" % (
            n, sweep, a, b, n / (a + b), "" if chk == ref else "MISMATCH", w[1] / max(1, w[0]), w[3] / max(1, w[2])), flush=True)
    del d1, d2, tb
    torch.cuda.empty_cache()
