"""One number for DESIGN.md section 5: the L = 64 schedule ("one wavefront per alignment", north_star's literal mapping) with
the diagonal dependency carried by __shfl_up(h, 1) -- which hipcc lowers to ds_bpermute_b32, an LDS-crossbar round trip -- against
the shipped DPP lane shift (v_*_dpp wave_shr:1, fused into the consumer).  Build the A/B library first (in the build
container: make -C smith-waterman-simd_amd/csrc ab_shfl_up), then on the GPU box: python tools/shfl_up_ab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "smith-waterman-simd_amd", "lib")
CHILD = r"""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(%r, "smith-waterman-simd_amd"))
import swmi
swmi.init(0)
n = 1 << 20
d1 = torch.empty(n * 128, dtype=torch.uint8, device="cuda"); d2 = torch.empty_like(d1)
out = torch.empty(n, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), n, 10000, 0, st)
sm = swmi.match_matrix(10, -30)
swmi.set_schedule(64, 0)
for _ in range(10): swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st)
ms = swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, iters=30)
m = 65536
a, b = swmi.generate_pairs_host(m, 10000, 0)
orc = ctypes.CDLL(os.path.join(%r, "oracle", "liboracle.so"))
want = np.zeros(m, np.int32)
vp = ctypes.c_void_p
orc.sw_oracle_batch(a.ctypes.data_as(vp), b.ctypes.data_as(vp), ctypes.c_size_t(m), sm.ctypes.data_as(vp), 15, want.ctypes.data_as(vp))
bad = int((out[:m].cpu().numpy() != want).sum())
print("%%-44s L = 64, 1,048,576 pairs (10,-30,15): %%.4f ms per launch = %%6.1f M alignments/s; %%d of %%d scores differ from the oracle" %% (
    os.path.basename(swmi.LIB_PATH), ms, n / ms / 1e3, bad, m))
""" % (ROOT, ROOT)
for lib in ("libswmi.so", "libswmi_shfl_up.so"):
    path = os.path.join(LIBDIR, lib)
    if not os.path.exists(path):
        sys.exit("%s missing: make -C smith-waterman-simd_amd/csrc%s" % (path, " ab_shfl_up" if "shfl" in lib else ""))
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SWMI_LIB=path), check=True)
