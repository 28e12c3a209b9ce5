"""Does the page size of the CALLER's arrays matter for the packed host entry?  (The HIP runtime stages pageable memory through
its own pinned buffers with a CPU copy.)  Usage (GPU box): python tools/experiments/host_thp.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "smith-waterman-simd_amd"))
import swmi
swmi.init(0)
libc = ctypes.CDLL("libc.so.6", use_errno=True)
n = 1 << 20
sm = swmi.match_matrix(10, -30)
rng = np.random.default_rng(1)

def make(huge):
    bufs = []
    for _ in range(2):
        raw = np.empty(n * 32 + (2 << 20), np.uint8)
        off = (-raw.ctypes.data) % (2 << 20)
        a = raw[off: off + n * 32]
        if huge:
            rc = libc.madvise(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 14)   # MADV_HUGEPAGE
            if rc != 0:
                print("madvise failed", ctypes.get_errno())
        a[:] = rng.integers(0, 256, n * 32, dtype=np.uint8)
        bufs.append((raw, a))
    return bufs

for huge in (False, True, False, True):
    (r1, a), (r2, b) = make(huge)
    out = np.empty(n, np.int32)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        swmi._check(swmi.load().swmi_score_batch_packed(a.ctypes.data, b.ctypes.data, n, sm.ctypes.data, 15, out.ctypes.data))
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("huge pages %-5s: median %.3f ms, min %.3f ms (%.0f M alignments/s at the median)" % (huge, ts[4], ts[0], n / ts[4] / 1e3), flush=True)
try:
    print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
except OSError as e:
    print(e)
