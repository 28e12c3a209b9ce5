#!/usr/bin/env python3
"""Parity on the golden fixtures + 1M-pair timing for a list of (lanes, flags) schedules: SCHED="8:0,8:4,4:4"."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smith-waterman-simd_amd"))
import swmi, torch
swmi.init(0)
G = os.path.join(ROOT, "tests", "golden")
fixtures = {k: np.load(os.path.join(G, k + ".npz")) for k in ("f1_random", "f2_structured", "f4_param_sweep")}
sched = [tuple(int(v) for v in x.split(":")) for x in os.environ.get("SCHED", "8:0,8:4").split(",")]
n = 1 << 20
d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 10000, 0, st)
sm = swmi.match_matrix(10, -30)
for L, flags in sched:
    swmi.set_schedule(L, flags)
    bad = 0
    for name, f in fixtures.items():
        for p in range(f["sm"].shape[0]):
            got = swmi.score_batch(f["seq1"], f["seq2"], f["sm"][p], int(f["gap"][p]))
            nb = int((got != f["scores"][p]).sum())
            if nb and bad == 0:
                i = int(np.nonzero(got != f["scores"][p])[0][0])
                print("  MISMATCH %s param %d gap %d: idx %d got %d want %d (%d bad)" % (name, p, f["gap"][p], i, got[i], f["scores"][p][i], nb))
            bad += nb
    swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st)
    torch.cuda.synchronize()
    ms = min(swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=10) for _ in range(3))
    print("L=%2d flags=%d parity %s  %.3f ms / 1M -> %.1f M align/s %.2f TCUPS checksum %d" % (
        L, flags, "OK" if bad == 0 else "FAIL(%d)" % bad, ms, n / ms / 1e3, n * 16384 / ms / 1e9, int(out.sum().item())), flush=True)
