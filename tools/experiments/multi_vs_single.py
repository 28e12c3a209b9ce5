import sys, time
sys.path.insert(0, "smith-waterman-simd_amd")
import numpy as np, swmi
n = 1 << 20
sm = swmi.match_matrix(10, -30)
def t(f, reps=6):
    out = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); out.append((time.perf_counter() - t0) * 1e3)
    return " ".join("%.2f" % x for x in out)
swmi.init(0)
a, b = swmi.generate_pairs_host(n, 10000, 0)
print("init(0)            score_batch      ", t(lambda: swmi.score_batch(a, b, sm, 15)))
swmi.shutdown()
swmi.init_devices([0])
print("init_devices([0])  score_batch      ", t(lambda: swmi.score_batch(a, b, sm, 15)))
print("init_devices([0])  score_batch_multi", t(lambda: swmi.score_batch_multi(a, b, sm, 15)))
print("init_devices([0])  score_batch      ", t(lambda: swmi.score_batch(a, b, sm, 15)))
a2, b2 = swmi.generate_pairs_host(n, 10000, 0)
print("fresh buffers      score_batch_multi", t(lambda: swmi.score_batch_multi(a2, b2, sm, 15)))
print("fresh buffers      score_batch      ", t(lambda: swmi.score_batch(a2, b2, sm, 15)))
swmi.shutdown()
