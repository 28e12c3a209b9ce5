import os, sys, time
import numpy as np, torch
sys.path.insert(0, "smith-waterman-simd_amd")
import swmi
swmi.init(0)
sm = swmi.match_matrix(10, -30)
n = 1 << 20
a, b = swmi.generate_pairs_host(n, 10000, 0)
ref = swmi.score_batch(a, b, sm, 15)
for packed in (False, True):
    pa, pb = (swmi.pack(a), swmi.pack(b)) if packed else (a, b)
    for chunk in (1 << 20, 1 << 19, 1 << 18, 1 << 17, 1 << 16, 1 << 15):
        os.environ["SWMI_HOST_CHUNK"] = str(chunk)
        f = (lambda: swmi.score_batch_packed(pa, pb, sm, 15)) if packed else (lambda: swmi.score_batch(pa, pb, sm, 15))
        out = f(); assert np.array_equal(out, ref)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        print("packed" if packed else "bytes ", "chunk %8d: %6.2f ms  %6.1f M alignments/s" % (chunk, min(ts) * 1e3, n / min(ts) / 1e6), flush=True)
