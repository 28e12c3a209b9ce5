#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (made by tools/profile_gpu.sh) into profiles/<tag>_*.{csv,json}.

HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B... the
counters are reported in kilobytes; on gfx950 FETCH_SIZE reads exactly half of the bytes of a wide coalesced streaming
read (128-B requests tallied at 64 B), so the read side is doubled; WRITE_SIZE is exact for streaming stores.
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
KERNEL = "sw128_"          # sw128_pk_kernel (what the default schedule runs for large batches) or sw128_kernel


def counter_rows(sub):
    rows = []
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel_avg(rows):
    acc = {}
    for r in rows:
        if KERNEL not in r.get("Kernel_Name", ""):
            continue
        name, val = r["Counter_Name"], float(r["Counter_Value"])
        s, n = acc.get(name, (0.0, 0))
        acc[name] = (s + val, n + 1)
    return {k: s / n for k, (s, n) in acc.items()}


summary = {"tag": tag}
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, tag + "_kernel_stats.csv"))
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Name"]:
            summary["kernel"] = r["Name"]
            summary["calls"] = int(r["Calls"])
            summary["avg_ns"] = float(r["AverageNs"])
            summary["min_ns"] = float(r["MinNs"])
for sub in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write", "pmc_derived", "pmc_derived2"):
    summary.update({k: round(v, 1) for k, v in per_kernel_avg(counter_rows(sub)).items()})
if "SQ_BUSY_CYCLES" in summary and "SQ_ACTIVE_INST_VALU" in summary:
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over SIMDs; SQ_BUSY_CYCLES per SE ... keep raw numbers, derive simple ratios
    pass
if "FETCH_SIZE" in summary or "WRITE_SIZE" in summary:
    fetch_b = summary.get("FETCH_SIZE", 0.0) * 1024.0
    write_b = summary.get("WRITE_SIZE", 0.0) * 1024.0
    summary["hbm_read_bytes_corrected"] = fetch_b * 2.0     # gfx950: FETCH_SIZE reports half of a wide coalesced read
    summary["hbm_write_bytes"] = write_b
    summary["hbm_bytes_per_launch"] = fetch_b * 2.0 + write_b
# tie the numbers to the code they were taken on: hash of the kernel's instruction text in the libswmi.so of this tree
# (tools/isa_census.py) -- bench.py quotes traffic / effective clock only while the library it times carries the same code
sys.path.insert(0, os.path.join(ROOT, "tools"))
try:
    import re
    import isa_census
    targs = re.search(r"(sw128_\w*kernel)<([^>]*)>", summary.get("kernel", ""))
    if targs:
        vals = [{"true": "1", "false": "0"}.get(v.strip(), v.strip()) for v in targs.group(2).split(",")]
        if targs.group(1) == "sw128_pk_kernel" and len(vals) == 3 and vals[2] == "4":
            vals = vals[:2]                 # <MODE, BIAS, L>: the L = 4 instantiation keeps its two-argument name (tools/isa_census.py)
        readable = "%s<%s>" % (targs.group(1), ",".join(vals))
        c = isa_census.census_for("^" + re.escape(readable) + "$", marker_op="v_perm_b32" if "pk" in readable else "v_dot4_i32_i8")
        summary["kernel_readable"] = readable
        summary["kernel_code_sha256"] = c[readable]["code_sha256"]
except Exception as e:      # noqa: BLE001
    summary["kernel_code_sha256_error"] = repr(e)
if "GRBM_GUI_ACTIVE" in summary and "avg_ns" in summary:
    # MI355X_MICROARCH.md, DVFS: GRBM_GUI_ACTIVE is summed over the 8 XCDs
    summary["effective_clock_ghz"] = round(summary["GRBM_GUI_ACTIVE"] / 8.0 / summary["avg_ns"], 3)
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
if "hbm_bytes_per_launch" in summary and "SQ_WAVES" in summary:
    # bench.py reads this for roofline.traffic (per launch of the default 1 048 576-pair step)
    json.dump({"pairs_per_launch": 1 << 20, "hbm_bytes_per_launch": round(summary["hbm_bytes_per_launch"]),
               "kernel": summary.get("kernel_readable"), "kernel_code_sha256": summary.get("kernel_code_sha256"),
               "effective_clock_ghz": summary.get("effective_clock_ghz"),
               "sq_insts_valu_per_launch": summary.get("SQ_INSTS_VALU"),
               "source": "profiles/%s_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)" % tag},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1, sort_keys=True))
