"""Throughput of the host-buffer semi-global entry point (PCIe inclusive): python tools/sg_host_rate.py [n]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi
swmi.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(1)
a = rng.integers(0, 4, (n, 16384), dtype=np.uint8)
b = np.where(rng.random((n, 16384)) < 0.95, a, rng.integers(0, 4, (n, 16384), dtype=np.uint8)).astype(np.uint8)
import ctypes
lib = swmi.load()
cap = swmi.SG_MAX_TRACEBACK
scores = np.zeros(n, np.int32); lengths = np.zeros(n, np.uint32); tb = np.zeros((n, cap, 2), np.int32)
for rep in range(2):
    t0 = time.perf_counter()
    rc = lib.swmi_semiglobal_xdrop(a.ctypes.data, b.ctypes.data, n, scores.ctypes.data, tb.ctypes.data, cap, lengths.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0
    print("host entry: %d alignments in %.1f ms = %.1f k alignments/s (mean length %.0f)" % (n, dt * 1e3, n / dt / 1e3, lengths.mean()), flush=True)
