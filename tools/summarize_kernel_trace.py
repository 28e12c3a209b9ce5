"""Steady-state per-kernel durations from rocprofv3's kernel trace (csv): rocprofv3 --stats averages EVERY launch, cold ones
included (round 3: 1.0747 ms average over 270 launches against 1.0679 ms per step in the bench, because the first launches
after start-up ran up to 1.185 ms) -- this prints, per kernel, calls / mean / median / min / max and the mean without the
first SKIP launches, which is what compares with bench.py's `kernel_ms`.
Usage: summarize_kernel_trace.py <kernel_trace.csv> [skip=10]"""
import csv, statistics, sys
path, skip = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
by = {}
with open(path) as f:
    for r in csv.DictReader(f):
        by.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("# %s: per kernel, durations in ms; steady = mean without the first %d launches (in start order)" % (path.split("/")[-1], skip))
print("%-72s %6s %9s %9s %9s %9s %9s" % ("kernel", "calls", "mean", "median", "min", "max", "steady"))
for name, rows in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    rows.sort()
    d = [x[1] / 1e6 for x in rows]
    steady = d[skip:] if len(d) > 2 * skip else d
    print("%-72s %6d %9.4f %9.4f %9.4f %9.4f %9.4f" % (name.split("(")[0][-72:], len(d), statistics.mean(d), statistics.median(d), min(d), max(d),
                                                      statistics.mean(steady)))
