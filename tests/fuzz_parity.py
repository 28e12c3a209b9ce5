#!/usr/bin/env python3
"""Long-running differential fuzz of the GPU scorer against the CPU oracle (and the real reference where present) --
the reference's own TestSimdSmithWaterman idea (source.cpp:2943-2982: fresh random pairs until bored), at GPU scale.

    python tests/fuzz_parity.py --seconds 240            # on the GPU box; writes a summary line per parameter set

Every round generates a fresh batch on the device (counter-based generator, new seed), scores it through the C ABI,
copies the inputs back and scores them with oracle/liboracle.so on all host cores (OpenMP); any mismatch is dumped.
A second phase does the same for the semi-global aligner against the reference's simd_mark4 (full tracebacks); a third
one for the banded affine extension against the oracle's scalar Gotoh (random lengths, matrices, open / extend)."""
import argparse, ctypes, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ lives one level below the repo root
sys.path.insert(0, os.path.join(ROOT, "smith-waterman-simd_amd"))
import swmi, torch

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--batch", type=int, default=1 << 22)
ap.add_argument("--sg-seconds", type=float, default=60)
ap.add_argument("--ba-seconds", type=float, default=30)
args = ap.parse_args()
swmi.init(0)
vp = ctypes.c_void_p
orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
rng = np.random.default_rng(int(time.time()))
n = args.batch
d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda"); d2 = torch.empty(n * 128, dtype=torch.uint8, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
t_end = time.time() + args.seconds
total = mism = rounds = 0
while time.time() < t_end:
    seed = int(rng.integers(0, 2**62)); first = int(rng.integers(0, 2**40))
    kind = rounds % 5
    if kind == 0: sm = swmi.match_matrix(10, -30); gap = 15
    elif kind == 1: sm = swmi.match_matrix(1, -1); gap = 1
    elif kind == 2: sm = rng.integers(-128, 128, 16).astype(np.int8); gap = int(rng.integers(0, 128))
    elif kind == 3: sm = rng.integers(-12, 13, 16).astype(np.int8); gap = int(rng.integers(0, 9))
    else:                    # every score + 2 gap in [0, 255], some score + gap < 0: the packed kernel's vertical-offset cell
        gap = int(rng.integers(1, 61))
        sm = rng.integers(-2 * gap, min(127, 255 - 2 * gap) + 1, 16).astype(np.int8)
        sm[int(rng.integers(0, 16))] = -2 * gap
    L = [4, 4, 8, 4, 16, 2, 4, 32, 64][rounds % 9]      # L = 4 (the packed kernel unless a flag says otherwise) meets every parameter family
    swmi.set_schedule(L, int((0, 0, 1, 8)[int(rng.integers(0, 4))]))
    swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, seed, first, st)
    if rounds % 3 == 1:      # make half of the batch related pairs (mutated copies) so that long alignments occur
        m = torch.rand(n * 128, device="cuda") < 0.85
        half = (n // 2) * 128
        d2[:half] = torch.where(m[:half], d1[:half], d2[:half])
    entry = rounds % 5                   # the sibling entry points share the kernel template: MODE 0 / 1 / 2
    if entry == 3:                       # 2-bit packed inputs (source.cpp:1581), packed on the device
        def pack(d):
            v = d.view(n, 32, 4).to(torch.int32)
            return (v[..., 0] | (v[..., 1] << 2) | (v[..., 2] << 4) | (v[..., 3] << 6)).to(torch.uint8).contiguous()
        p1, p2 = pack(d1), pack(d2)
        swmi.score_batch_device(p1.data_ptr(), p2.data_ptr(), n, sm, gap, out.data_ptr(), st, packed=True)
    elif entry == 4:                     # every seq1 against ONE seq2 (source.cpp:1227)
        d2.view(n, 128)[:] = d2[:128].clone()
        swmi.score_one_vs_many_device(d1.data_ptr(), n, d2.data_ptr(), sm, gap, out.data_ptr(), st)
    else:
        swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, gap, out.data_ptr(), st)
    torch.cuda.synchronize()
    a = d1.cpu().numpy(); b = d2.cpu().numpy(); got = out.cpu().numpy()
    want = np.zeros(n, np.int32)
    orc.sw_oracle_batch(a.ctypes.data_as(vp), b.ctypes.data_as(vp), ctypes.c_size_t(n), sm.ctypes.data_as(vp), gap, want.ctypes.data_as(vp))
    bad = int((got != want).sum())
    total += n; mism += bad; rounds += 1
    if bad:
        i = int(np.nonzero(got != want)[0][0])
        print("MISMATCH seed %d first %d L %d gap %d sm %s: pair %d got %d want %d (%d bad)" % (seed, first, L, gap, sm.tolist(), i, got[i], want[i], bad), flush=True)
    if rounds % 10 == 0:
        print("... %d rounds, %.0f M pairs, %d mismatches" % (rounds, total / 1e6, mism), flush=True)
print("SW128 fuzz: %d rounds, %d pairs, %d mismatches (all six schedules, folded and general cell, five parameter families, pairs / packed / one-vs-many entries)" % (rounds, total, mism), flush=True)

# ---- semi-global aligner against the real reference (if present) or the oracle --------------------------------------
ref_path = os.path.join(ROOT, "oracle", "_ref", "libswref.so")
ref = ctypes.CDLL(ref_path) if os.path.exists(ref_path) else None
t_end = time.time() + args.sg_seconds
sg_total = sg_bad = 0
workers = min(64, os.cpu_count() or 8)
def check(arg):
    a, b, score, tb = arg
    buf = np.zeros((32769, 2), np.int32); sc = ctypes.c_int32(); ln = ctypes.c_size_t()
    if ref is not None:
        ref.swref_semiglobal(4, a.ctypes.data_as(vp), b.ctypes.data_as(vp), ctypes.byref(sc), buf.ctypes.data_as(vp), ctypes.c_size_t(32769), ctypes.byref(ln))
    else:
        oob = ctypes.c_int()
        orc.sg_oracle_xdrop(a.ctypes.data_as(vp), b.ctypes.data_as(vp), ctypes.byref(sc), buf.ctypes.data_as(vp), ctypes.c_size_t(32769), ctypes.byref(ln), ctypes.byref(oob))
    return sc.value == score and ln.value == len(tb) and np.array_equal(buf[: ln.value], tb)
sg_iter = 0
while time.time() < t_end:
    m = 2048 - 17 * (sg_iter % 4)        # ragged batches too: the last sweep wavefront is partly filled (interleaved streams, tail lanes)
    a = rng.integers(0, 4, (m, 16384), dtype=np.uint8)
    p = rng.random((m, 1)) * 0.3
    b = np.where(rng.random((m, 16384)) < p, rng.integers(0, 4, (m, 16384), dtype=np.uint8), a).astype(np.uint8)
    for k in range(0, m, 7):             # indels: shift a tail of some sequences by a few bases
        cut = int(rng.integers(100, 16000)); sh = int(rng.integers(1, 40))
        b[k, cut:] = np.roll(b[k], sh)[cut:]
    for k in range(3, m, 5):             # runs of guaranteed mismatches around the X-drop limit of 70: some alignments die there, some
        for _ in range(int(rng.integers(1, 5))):        # scrape past -- cells at the threshold, dropped cells, dead lanes beside live ones
            lo = int(rng.integers(50, 16200)); run = int(rng.integers(30, 111))
            b[k, lo: lo + run] = (a[k, lo: lo + run] + 1 + (rng.integers(0, 3, len(a[k, lo: lo + run])) if rng.random() < 0.5 else 0)) & 3
    swmi.semiglobal_set_exact(sg_iter % 5 == 4)          # every fifth batch on the exact path only (no calm windows)
    swmi.semiglobal_set_mapping((41, 42, 43, 44, 21, 22, 23, 11, 12)[sg_iter % 9])
    sg_iter += 1
    if sg_iter % 2:                      # the entry that returns the walk's 2-bit moves, expanded on the host (round 4)
        scores, moves, lengths = swmi.semiglobal_xdrop_moves(a, b)
        tbs = [swmi.semiglobal_expand_moves(moves[k], int(lengths[k])) for k in range(m)]
    else:
        scores, tbs, lengths = swmi.semiglobal_xdrop(a, b)
    with ThreadPoolExecutor(workers) as ex:
        ok = list(ex.map(check, [(a[k], b[k], int(scores[k]), tbs[k]) for k in range(m)]))
    sg_total += m; sg_bad += m - sum(ok)
    print("... semi-global %d alignments, %d mismatches" % (sg_total, sg_bad), flush=True)
swmi.semiglobal_set_exact(False)
print("semi-global fuzz vs %s: %d alignments (score + full traceback; positions entry and moves entry + host expansion in turn), %d mismatches" % ("reference simd_mark4" if ref else "oracle", sg_total, sg_bad), flush=True)

# ---- banded affine extension vs oracle/sw_oracle.c (no reference counterpart: parity unpinned by the reference) ----
orc.sw_oracle_banded_affine.restype = ctypes.c_int
ba_total = ba_bad = 0
ba_kernels = {}
t_end = time.time() + args.ba_seconds
while time.time() < t_end:
    length = int(rng.choice([64, 65, 100, 128, 200, 333, 512, 1000, 1024, 1500, 1792]))
    m = int(rng.integers(1, 40))
    a = rng.integers(0, 4, (m, length), dtype=np.uint8)
    p = rng.random((m, 1)) * 0.4
    b = np.where(rng.random((m, length)) < p, rng.integers(0, 4, (m, length), dtype=np.uint8), a).astype(np.uint8)
    for k in range(0, m, 3):             # indels: offsets up to and beyond the band
        cut = int(rng.integers(1, length)); sh = int(rng.integers(1, 80))
        b[k, cut:] = np.roll(b[k], sh)[cut:]
    kind = int(rng.integers(0, 4))       # any int8 matrix / match-mismatch with any gaps / the usual small gaps / small scores, any gaps
    if kind == 3:                        # scores small enough for the packed kernel at every length, either sign
        sm = rng.integers(-128 if rng.random() < 0.3 else -12, 15, 16).astype(np.int8)
    else:
        sm = (rng.integers(-128, 128, 16) if kind == 0 else
              np.where(np.eye(4, dtype=bool), rng.integers(0, 128), rng.integers(-128, 1)).reshape(16)).astype(np.int8)
    go, ge = (int(rng.integers(0, 20)), int(rng.integers(0, 8))) if kind == 2 else (int(rng.integers(0, 128)), int(rng.integers(0, 128)))
    ba_kernels[swmi.banded_affine_kernel_for(length, sm, go, ge)[0]] = ba_kernels.get(swmi.banded_affine_kernel_for(length, sm, go, ge)[0], 0) + m
    got = swmi.score_banded_affine(a, b, sm, go, ge)
    want = np.array([orc.sw_oracle_banded_affine(a[k].ctypes.data_as(vp), b[k].ctypes.data_as(vp), length, sm.ctypes.data_as(vp), go, ge)
                     for k in range(m)], np.int32)
    ba_total += m; ba_bad += int((got != want).sum())
print("banded affine fuzz vs oracle: %d alignments, %d mismatches (11 lengths 64..1792, random matrices, open/extend 0..127 either order); per kernel: %s" % (
    ba_total, ba_bad, ", ".join("%s %d" % kv for kv in sorted(ba_kernels.items()))), flush=True)
