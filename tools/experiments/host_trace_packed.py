import os, sys
sys.path.insert(0, "smith-waterman-simd_amd")
os.environ["SWMI_HOST_TRACE"] = "1"
import numpy as np, swmi
swmi.init(0)
sm = swmi.match_matrix(10, -30)
a, b = swmi.generate_pairs_host(1 << 20, 10000, 0)
pa, pb = swmi.pack(a), swmi.pack(b)
for _ in range(3):
    swmi.score_batch_packed(pa, pb, sm, 15)
