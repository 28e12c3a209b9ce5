"""A/B of two builds of libswmi.so on the same box: headline kernel time at 1 M and 16 M pairs per launch, alternating.
Usage (GPU box): python tools/experiments/pk_ab.py <lib A> <lib B>"""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(os.getcwd(), "smith-waterman-simd_amd"))
import torch, swmi
swmi.init(0)
sm = swmi.match_matrix(10, -30)
dev = torch.device("cuda", 0)
nmax = 1 << 24
d1 = torch.empty(nmax * 128, dtype=torch.uint8, device=dev); d2 = torch.empty_like(d1)
out = torch.empty(nmax, dtype=torch.int32, device=dev)
swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), nmax, 10000, 0)
st = torch.cuda.current_stream().cuda_stream
res = []
for n in (1 << 20, 1 << 24):
    swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, 5)
    ms = min(swmi.time_batch_device(d1.data_ptr(), d2.data_ptr(), n, sm, 15, out.data_ptr(), st, 60 if n < (1 << 22) else 8) for _ in range(4))
    res.append("%d pairs %.4f ms (%.2f us per 1 M)" % (n, ms, ms * 1e3 * (1 << 20) / n))
print(os.path.basename(swmi.LIB_PATH), "|", " | ".join(res), "| checksum", int(out[: 1 << 20].sum().item()), flush=True)
'''
for rep in range(int(os.environ.get("AB_REPS", "3"))):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, SWMI_LIB=os.path.abspath(lib)), check=True)
