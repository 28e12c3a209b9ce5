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
# split into calls: a gap of more than 1 ms between consecutive events starts a new call
calls, cur = [], []
for e in ev:
    if cur and e[1] - max(x[2] for x in cur) > 1_000_000:
        calls.append(cur); cur = []
    cur.append(e)
if cur: calls.append(cur)
def describe(call):
    t0 = call[0][1]
    h2d = [(s, e) for kind, s, e, _ in call if kind == "H2D"]
    kern = [(s, e) for kind, s, e, _ in call if kind == "K"]
    under = 0
    for ks, ke in kern:
        for hs, he in h2d:
            under += max(0, min(ke, he) - max(ks, hs))
    tot_k = sum(e - s for s, e in kern)
    print("call: %d H2D copies %.3f ms, %d kernels %.3f ms, span %.3f ms; kernel time under an H2D copy: %.3f ms (%.0f %%)" % (
        len(h2d), sum(e - s for s, e in h2d) / 1e6, len(kern), tot_k / 1e6, (max(x[2] for x in call) - t0) / 1e6, under / 1e6,
        100.0 * under / tot_k if tot_k else 0))
    for kind, s, e, what in call:
        print("   %-4s %9.3f .. %9.3f ms  %s" % (kind, (s - t0) / 1e6, (e - t0) / 1e6, what))
big = [c for c in calls if sum(1 for x in c if x[0] == "H2D") >= 2]
for c in (big[2:3] + big[-1:]) if len(big) >= 4 else big[-2:]:
    describe(c)
