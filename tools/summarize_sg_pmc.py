#!/usr/bin/env python3
"""Summarise gpurun_out/prof_sg/ (tools/profile_sg_pmc.sh) into profiles/<tag>_semiglobal_pmc.json (tag = argv[1], default r02): per-kernel averages of
the PMC counters of the semi-global kernels.  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies
128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section) -- for the scattered 64-byte line reads of the walk kernel
the doubling is an upper bound."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_sg")
acc = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "swmi" not in name:
                continue
            short = name.split("::")[-1].split("(")[0]
            key = (short, r["Counter_Name"])
            s, n = acc.get(key, (0.0, 0))
            acc[key] = (s + float(r["Counter_Value"]), n + 1)
out = {}
for (k, c), (s, n) in sorted(acc.items()):
    out.setdefault(k, {})[c] = s / n
for k, d in out.items():
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_x2"] = d["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
# tie every kernel's numbers to the code they were taken on (tools/isa_census.py hash of the instruction text)
sys.path.insert(0, os.path.join(ROOT, "tools"))
hashes = {}
try:
    import re
    import isa_census
    for k in list(out):
        m = re.match(r"(\w+)(<([^>]*)>)?", k)
        readable = m.group(1) + ("<%s>" % ",".join({"true": "1", "false": "0"}.get(v.strip(), v.strip()) for v in m.group(3).split(",")) if m.group(3) else "")
        c = isa_census.census_for("^" + re.escape(readable) + "$")
        if readable in c:
            hashes[k] = c[readable]["code_sha256"]
except Exception as e:      # noqa: BLE001
    hashes["error"] = repr(e)
out["kernel_code_sha256"] = hashes
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_semiglobal_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
