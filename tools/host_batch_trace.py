"""Driver for the copy / kernel trace of the host-buffer entry (swmi_score_batch): a few calls at 1M and 4M pairs from pageable
memory.  Run under rocprofv3 --kernel-trace --memory-copy-trace (tools/profile_host_batch.sh); tools/summarize_host_trace.py
turns the two trace files into the overlap table of profiles/r03_host_batch_trace.txt."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import numpy as np
import swmi
swmi.init(0)
sm = swmi.match_matrix(10, -30)
for n in (1 << 20, 1 << 22):
    a, b = swmi.generate_pairs_host(n, 10000, 0)
    swmi.score_batch(a, b, sm, 15)                       # buffers allocated, clocks up
    for rep in range(3):
        t0 = time.perf_counter()
        swmi.score_batch(a, b, sm, 15)
        print("n %8d call %d: %.3f ms wall  granules %s" % (n, rep, (time.perf_counter() - t0) * 1e3, swmi.host_granules(n)), flush=True)
swmi.shutdown()
