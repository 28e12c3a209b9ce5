"""Overlap table of one swmi_score_batch call from rocprofv3's kernel trace + memory-copy trace (csv):
for the LAST call of each batch size: every H2D copy and kernel with its start / end relative to the call's first copy, and how
much of the kernel time lies under a host-to-device copy.  Usage: summarize_host_trace.py <kernel_trace.csv> <memory_copy_trace.csv>"""
import csv, sys
def rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))
k = [r for r in rows(sys.argv[1]) if "sw128" in r["Kernel_Name"]]
m = rows(sys.argv[2])
ev = [("K", int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:28]) for r in k]
ev += [("H2D" if "HOST_TO_DEVICE" in r["Direction"].upper() or "H2D" in r["Direction"].upper() else "D2H", int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
        "%d B" % int(r.get("Bytes", r.get("Size", 0)) or 0)) for r in m]
ev.sort(key=lambda e: e[1])
# split into calls: a call ends with the device-to-host copy of its scores
calls, cur = [], []
for e in ev:
    cur.append(e)
    if e[0] == "D2H":
        calls.append(cur); cur = []
def describe(call):
    t0 = call[0][1]
    h2d = [(s, e) for kind, s, e, _ in call if kind == "H2D"]
    kern = [(s, e) for kind, s, e, _ in call if kind == "K"]
    under = 0
    for ks, ke in kern:
        for hs, he in h2d:
            under += max(0, min(ke, he) - max(ks, hs))
    tot_k = sum(e - s for s, e in kern)
    print("call: %d H2D copy commands %.3f ms busy, %d kernels %.3f ms, first copy to end of the score copy %.3f ms; "
          "kernel time under an H2D copy: %.3f ms (%.0f %%); exposed kernel time %.3f ms" % (
        len(h2d), sum(e - s for s, e in h2d) / 1e6, len(kern), tot_k / 1e6, (max(x[2] for x in call) - t0) / 1e6, under / 1e6,
        100.0 * under / tot_k if tot_k else 0, (tot_k - under) / 1e6))
    for kind, s, e, what in call:
        print("   %-4s %9.3f .. %9.3f ms  %s" % (kind, (s - t0) / 1e6, (e - t0) / 1e6, what if kind == "K" else ""))
by_kernels = {}
for c in calls:
    nk = sum(1 for x in c if x[0] == "K")
    if nk:
        by_kernels[nk] = c                     # the last call of each shape (4 granules = 1M pairs, 7 = 4M pairs)
for nk in sorted(by_kernels):
    describe(by_kernels[nk])
