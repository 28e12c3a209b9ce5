"""Host-buffer entry swmi_score_batch (PCIe inclusive) under different pipeline settings, on the GPU box.
SWMI_HOST_SERIAL=1 is round 2's order of issue (scores copied back behind every granule); the default is the tapered
schedule of swmi_api.cpp next_granule(); SWMI_HOST_GRANULE=<pairs> fixes the granule.
Usage: python tools/host_pipeline_experiment.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi
sm = swmi.match_matrix(10, -30)
settings = [("round 2 order (serial), 1M granules", {"SWMI_HOST_SERIAL": "1", "SWMI_HOST_GRANULE": str(1 << 20)}),
            ("deferred D2H, 1M granules", {"SWMI_HOST_GRANULE": str(1 << 20)}),
            ("deferred D2H, 512K granules", {"SWMI_HOST_GRANULE": str(1 << 19)}),
            ("deferred D2H, 256K granules", {"SWMI_HOST_GRANULE": str(1 << 18)}),
            ("deferred D2H, 128K granules", {"SWMI_HOST_GRANULE": str(1 << 17)}),
            ("deferred D2H, 64K granules", {"SWMI_HOST_GRANULE": str(1 << 16)}),
            ("tapered (default)", {})]
sizes = [1 << 20, 1 << 22]
h = {}
want = {}
for n in sizes:
    a, b = swmi.generate_pairs_host(n, 10000, 0)
    h[n] = (a, b, torch.from_numpy(a).pin_memory(), torch.from_numpy(b).pin_memory())
for label, env in settings:
    for k in ("SWMI_HOST_SERIAL", "SWMI_HOST_GRANULE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    swmi.init(0)
    for n in sizes:
        a, b, pa, pb = h[n]
        for kind, (x, y) in (("pageable", (a, b)), ("pinned", (pa.numpy(), pb.numpy()))):
            got = swmi.score_batch(x, y, sm, 15)
            if n not in want:
                want[n] = got
            assert np.array_equal(got, want[n]), (label, n, kind, int((got != want[n]).sum()))
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                swmi.score_batch(x, y, sm, 15)
                best = min(best, time.perf_counter() - t0)
            gran = swmi.host_granules(n)
            print("%-36s n = %8d %-9s %8.3f ms  %7.1f M alignments/s   granules %s" % (
                label, n, kind, best * 1e3, n / best / 1e6, gran if len(gran) <= 8 else "%d x ..%s" % (len(gran), gran[-4:])), flush=True)
    swmi.shutdown()
