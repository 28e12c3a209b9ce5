"""Round 3 A/B of the packed kernel's cell bodies on the GPU box: the paired-add forms (q0 = variant 0, vertical offset =
variant 2) against round 2's (SWMI_PK_OLD=1: variants 3 and 1), same scores as the int32 kernel, time per launch of 1M pairs.
Usage: python tools/pk_variants_timing.py   (QUICK_REPS, QUICK_LANES=4,8,16)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi
n = 1 << 20
REPS = int(os.environ.get("QUICK_REPS", "40"))
LANES = [int(x) for x in os.environ.get("QUICK_LANES", "4").split(",")]
rng = np.random.default_rng(5)
cases = [("10/-30/15", swmi.match_matrix(10, -30), 15), ("1/-1/1", swmi.match_matrix(1, -1), 1), ("2/-3/5", swmi.match_matrix(2, -3), 5),
         ("5/-4/0", swmi.match_matrix(5, -4), 0), ("127/-127/127", swmi.match_matrix(127, -127), 127),
         ("random small", rng.integers(-12, 13, 16).astype(np.int8), 7), ("random", rng.integers(-128, 128, 16).astype(np.int8), 77)]
results = {}
for old in (0, 1):
    os.environ["SWMI_PK_OLD"] = str(old)
    swmi.init(0)
    d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda"); d2 = torch.empty_like(d1)
    out = torch.empty(n, dtype=torch.int32, device="cuda"); ref = torch.empty_like(out)
    st = torch.cuda.current_stream().cuda_stream
    swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 10000, 0, st)
    m = torch.rand(n * 128, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) < 0.85
    half = (n // 2) * 128
    d2[:half] = torch.where(m[:half], d1[:half], d2[:half])        # half of the batch: related pairs (long alignments)
    for _ in range(20):                                               # settle the clocks
        swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, cases[0][1], 15, out.data_ptr(), st)
    for lanes in LANES:
        for name, sm, gap in cases:
            swmi.set_schedule(lanes, swmi.NO_PACKED)
            swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, gap, ref.data_ptr(), st)
            swmi.set_schedule(lanes, 0)
            for _ in range(3): swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, gap, out.data_ptr(), st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(REPS): swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, gap, out.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / REPS
            bad = int((out != ref).sum())
            kern = swmi.score_kernel_for_batch(n, sm, gap)[0]
            results[(lanes, name, old)] = ms
            print("%s L=%-2d %-14s %-24s %.4f ms = %7.1f M/s   mismatches vs int32 kernel %d   max score %d" % (
                "round2" if old else "round3", lanes, name, kern, ms, n / ms / 1e3, bad, int(ref.max())), flush=True)
    swmi.set_schedule(0, 0)
    del d1, d2, out, ref
    swmi.shutdown()
print()
for lanes in LANES:
    for name, _, _ in cases:
        a, b = results[(lanes, name, 0)], results[(lanes, name, 1)]
        print("L=%-2d %-14s round 3 %.4f ms   round 2 %.4f ms   ratio %.3f" % (lanes, name, a, b, a / b))
