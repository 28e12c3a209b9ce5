"""Driver for the copy / kernel trace of a host-buffer entry: a few calls at 1M and 4M pairs from pageable memory.
Entry (argv[1]): pairs = swmi_score_batch (default), packed = swmi_score_batch_packed, ovm = swmi_score_one_vs_many.
Run under rocprofv3 --kernel-trace --memory-copy-trace (tools/profile_host_batch.sh); tools/summarize_host_trace.py turns the
two trace files into the overlap table of profiles/r0N_host_batch_trace*.txt."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import numpy as np
import swmi
entry = sys.argv[1] if len(sys.argv) > 1 else "pairs"
swmi.init(0)
sm = swmi.match_matrix(10, -30)
for n in (1 << 20, 1 << 22):
    a, b = swmi.generate_pairs_host(n, 10000, 0)
    if entry == "packed":
        x, y = swmi.pack(a), swmi.pack(b)
        call, eid = (lambda: swmi.score_batch_packed(x, y, sm, 15)), swmi.ENTRY_PACKED
    elif entry == "ovm":
        x, y = a, b[0].copy()
        call, eid = (lambda: swmi.score_one_vs_many(x, y, sm, 15)), swmi.ENTRY_ONE_VS_MANY
    else:
        x, y = a, b
        call, eid = (lambda: swmi.score_batch(x, y, sm, 15)), swmi.ENTRY_PAIRS
    call()                                               # buffers allocated, clocks up
    for rep in range(3):
        t0 = time.perf_counter()
        call()
        g = swmi.host_granules(n, eid)
        print("n %8d %s call %d: %.3f ms wall  granules %s" % (n, entry, rep, (time.perf_counter() - t0) * 1e3,
              g if len(g) <= 8 else "%d x %s..%s" % (len(g), g[:2], g[-2:])), flush=True)
swmi.shutdown()
