"""Packed host entry, 1M pairs: schedules that differ in their LAST granules (the kernels still to run when the last copy has
landed are the exposed tail: tools/experiments/host_trace_packed.py, SWMI_HOST_TRACE=1).  Usage: python tools/experiments/host_tail_schedules.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "smith-waterman-simd_amd"))
import swmi
sm = swmi.match_matrix(10, -30)
n = 1 << 20
a, b = swmi.generate_pairs_host(n, 10000, 0)
pa, pb = swmi.pack(a), swmi.pack(b)
want = None
lists = {"default": None,
         "A 32,96,128x5,96,64,48,32,16": [32, 96, 128, 128, 128, 128, 128, 96, 64, 48, 32, 16],
         "B 32,96,128x6,64,32,32": [32, 96, 128, 128, 128, 128, 128, 128, 64, 32, 32],
         "C 32,96,128x5,96,64,32,24,24,16": [32, 96, 128, 128, 128, 128, 128, 96, 64, 32, 24, 24, 16],
         "D 64,128x6,96,64,32": [64, 128, 128, 128, 128, 128, 128, 96, 64, 32],
         "E 32,96,160x5,64,32": [32, 96, 160, 160, 160, 160, 160, 64, 32]}
for rep in range(2):
    for label, ks in lists.items():
        os.environ.pop("SWMI_HOST_SCHEDULE", None)
        if ks:
            assert sum(ks) == 1024, (label, sum(ks))
            os.environ["SWMI_HOST_SCHEDULE"] = ",".join(str(k << 10) for k in ks)
        swmi.init(0)
        got = swmi.score_batch_packed(pa, pb, sm, 15)
        want = got if want is None else want
        assert np.array_equal(got, want)
        t = []
        for _ in range(9):
            t0 = time.perf_counter(); swmi.score_batch_packed(pa, pb, sm, 15); t.append(time.perf_counter() - t0)
        t.sort()
        print("%-40s min %.3f  median %.3f ms   %d granules" % (label, t[0] * 1e3, t[4] * 1e3, len(swmi.host_granules(n, swmi.ENTRY_PACKED))), flush=True)
        swmi.shutdown()
