"""Host-buffer entries (PCIe inclusive) under different pipeline settings, on the GPU box: swmi_score_batch (256 B per pair
over the link), swmi_score_batch_packed (64 B) and swmi_score_one_vs_many (128 B).
SWMI_HOST_SERIAL=1 is round 2's order of issue (scores copied back behind every granule); the default is the per-entry tapered
schedule of swmi_api.cpp next_granule(); SWMI_HOST_GRANULE=<pairs> fixes the granule, SWMI_HOST_TAPER=<percent> the
granule-to-granule ratio, SWMI_HOST_MIN_GRANULE=<pairs> the smallest granule.
Usage: python tools/host_pipeline_experiment.py [pairs|packed|ovm|all]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smith-waterman-simd_amd"))
import swmi
which = sys.argv[1] if len(sys.argv) > 1 else "all"
sm = swmi.match_matrix(10, -30)
KNOBS = ("SWMI_HOST_SERIAL", "SWMI_HOST_GRANULE", "SWMI_HOST_TAPER", "SWMI_HOST_MIN_GRANULE", "SWMI_HOST_THREADS", "SWMI_HOST_SLOTS", "SWMI_HOST_SCHEDULE")


def fixed(g, threads=2, slots=3):
    return ("fixed %dK granules, %d thread(s) x %d buffer sets" % (g >> 10, threads, slots),
            {"SWMI_HOST_GRANULE": str(g), "SWMI_HOST_THREADS": str(threads), "SWMI_HOST_SLOTS": str(slots)})


def taper(pct, smallest, threads=2):
    return ("taper %d %%, smallest %dK, %d thread(s)" % (pct, smallest >> 10, threads),
            {"SWMI_HOST_TAPER": str(pct), "SWMI_HOST_MIN_GRANULE": str(smallest), "SWMI_HOST_THREADS": str(threads)})


def sched(ks, threads=2, slots=2):
    return ("schedule %s K, %d thread(s) x %d sets" % (",".join(str(k) for k in ks), threads, slots),
            {"SWMI_HOST_SCHEDULE": ",".join(str(k << 10) for k in ks), "SWMI_HOST_THREADS": str(threads), "SWMI_HOST_SLOTS": str(slots)})


R3 = {"SWMI_HOST_TAPER": "25", "SWMI_HOST_MIN_GRANULE": str(1 << 14), "SWMI_HOST_THREADS": "1", "SWMI_HOST_SLOTS": "3"}
settings = {
    "pairs": [("round 2 order (serial), 1M granules", {"SWMI_HOST_SERIAL": "1", "SWMI_HOST_GRANULE": str(1 << 20)}),
              ("round 3 pipeline (one thread)", R3), fixed(1 << 18), fixed(1 << 17), taper(25, 1 << 14), taper(50, 1 << 15),
              taper(35, 1 << 15, 1), taper(50, 1 << 15, 1), ("default", {})],
    "packed": [("round 3 pipeline (256-byte taper, one thread)", R3),
               fixed(1 << 18, 1), fixed(1 << 17, 2, 2), fixed(1 << 16, 2, 2),
               sched([64, 128, 192, 256, 192, 128, 64]), sched([32, 64, 128, 256, 256, 128, 96, 64]), sched([64, 128, 256, 256, 192, 128]),
               sched([32, 64, 96, 128, 160, 160, 128, 96, 64, 48, 32, 16]), sched([16, 32, 64, 128, 256, 256, 128, 64, 48, 32]),
               sched([32, 64, 128, 192, 192, 192, 128, 64, 32], 1, 3), sched([32, 64, 128, 192, 192, 192, 128, 64, 32], 2, 3),
               sched([32, 96, 128, 128, 128, 128, 128, 128, 96, 32]), ("default", {})],
    "ovm": [("round 3 pipeline (256-byte taper, one thread)", R3), fixed(1 << 17), taper(50, 1 << 14, 1), taper(50, 1 << 14),
            taper(60, 1 << 15, 1), taper(70, 1 << 15, 1), ("default", {})],
}
sizes = [1 << 20, 1 << 22]
data = {}
for n in sizes:
    a, b = swmi.generate_pairs_host(n, 10000, 0)
    data[n] = {"pairs": (a, b), "packed": (swmi.pack(a), swmi.pack(b)), "ovm": (a, b[0].copy())}
call = {"pairs": lambda x, y: swmi.score_batch(x, y, sm, 15), "packed": lambda x, y: swmi.score_batch_packed(x, y, sm, 15),
        "ovm": lambda x, y: swmi.score_one_vs_many(x, y, sm, 15)}
entry_id = {"pairs": swmi.ENTRY_PAIRS, "packed": swmi.ENTRY_PACKED, "ovm": swmi.ENTRY_ONE_VS_MANY}
want = {}
for entry in ("pairs", "packed", "ovm"):
    if which not in ("all", entry):
        continue
    print("== %s" % entry, flush=True)
    for label, env in settings[entry]:
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)
        swmi.init(0)
        for n in sizes:
            x, y = data[n][entry]
            got = call[entry](x, y)
            key = (entry, n) if entry == "ovm" else ("pairs", n)
            if key not in want:
                want[key] = got
            assert np.array_equal(got, want[key]), (label, n, int((got != want[key]).sum()))
            times = []
            for _ in range(7):
                t0 = time.perf_counter()
                call[entry](x, y)
                times.append(time.perf_counter() - t0)
            times.sort()
            gran = swmi.host_granules(n, entry_id[entry])
            print("%-36s n = %8d  min %7.3f  median %7.3f ms  %7.1f M alignments/s (median)  granules %s" % (
                label, n, times[0] * 1e3, times[3] * 1e3, n / times[3] / 1e6,
                gran if len(gran) <= 8 else "%d x %s..%s" % (len(gran), gran[:2], gran[-3:])), flush=True)
        swmi.shutdown()
