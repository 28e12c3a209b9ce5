#!/usr/bin/env python3
"""Quick on-GPU sanity + timing sweep over schedules (developer tool; the real tests are tests/ -m gpu)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "smith-waterman-simd_amd"))
import swmi
import torch

swmi.init(0)
print(swmi.device_info())
G = os.path.join(ROOT, "tests", "golden")
fixtures = {k: np.load(os.path.join(G, k + ".npz")) for k in ("f1_random", "f2_structured", "f4_param_sweep")}
ok = True
sweep = [int(x) for x in os.environ.get("LANES", "64,32,16,8,4,2").split(",")]
for L in sweep:
    for flags in (0, 1, 2, 3):
        swmi.set_schedule(L, flags)
        bad = 0
        for name, f in fixtures.items():
            for p in range(f["sm"].shape[0]):
                got = swmi.score_batch(f["seq1"], f["seq2"], f["sm"][p], int(f["gap"][p]))
                nb = int((got != f["scores"][p]).sum())
                if nb:
                    bad += nb
                    if bad <= 3 * nb:
                        i = int(np.nonzero(got != f["scores"][p])[0][0])
                        print("MISMATCH L=%d flags=%d %s param %d (gap %d): first at %d got %d want %d (%d bad)" % (
                            L, flags, name, p, f["gap"][p], i, got[i], f["scores"][p][i], nb))
        print("L=%2d flags=%d parity: %s" % (L, flags, "OK" if bad == 0 else "FAIL (%d)" % bad))
        ok &= bad == 0

# generator: device == host
n = 1 << 20
d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 10000, 0, st)
torch.cuda.synchronize()
h1, h2 = swmi.generate_pairs_host(4096, 10000, 0)
print("generator device==host:", bool((d1[:4096 * 128].cpu().numpy().reshape(-1, 128) == h1).all() and (d2[:4096 * 128].cpu().numpy().reshape(-1, 128) == h2).all()))

sm = swmi.match_matrix(10, -30)
for L in sweep:
    for flags in (0, 1, 2, 3):
        swmi.set_schedule(L, flags)
        swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st)
        torch.cuda.synchronize()
        ms = swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=10)
        s = int(out.sum().item())
        print("L=%2d flags=%d  %.3f ms / 1M  -> %.1f M align/s  %.2f TCUPS  checksum %d" % (L, flags, ms, n / ms / 1e3, n * 16384 / ms / 1e9, s))
print("ALL PARITY OK" if ok else "PARITY FAILURES")
