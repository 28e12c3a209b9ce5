#!/usr/bin/env python3
"""bench.py -- alignments/s of the fixed-shape Smith-Waterman scorer on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--lanes L]

A "step" is one pass of the hot path over one batch of synthetic pairs that are already resident in HBM when the timed
region starts.

N = 1  every step scores P = 1,048,576 pairs (BASELINE.json configs[1], "1M same-shape pairs on 1 MI355X").  After the
       headline the same invocation appends `rows` -- short runs of the other rows of SURVEY.md section 8 (L = 64 = one
       wavefront per alignment, 2-bit packed input, one-vs-many, the semi-global X-drop aligner at 65,536 alignments, the
       banded affine extension at 1024 x 1024), each with kernel_ms, its VALU issue-bound fraction and gpu_mismatches
       against the reference / oracle -- and `sustained`, >= 2 s of back-to-back launches of the headline kernel.
N > 1  one rank per GPU (the driver starts `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`;
       started plainly with --gpus N > 1 the script spawns that launcher as a child).  Rank g scores the contiguous shard
       g of N * P pairs, P = 67,108,864 per GPU by default (BASELINE.json configs[3]: "512M pairs sharded across
       8 x MI355X, per-GPU sub-batch + RCCL gather"), and the int32 scores are all-gathered over RCCL (the path's only
       exchange step) through swmi/sharding.py.  `value` = the leg with a gather after EVERY step
       (asynchronous, overlapping the next step's kernel); `legs` also reports "one final gather", "no gather" and the
       configs[1]-sized leg (1M pairs per GPU per step, gather every step), each with per-rank kernel_ms / gather_ms, and
       rank 0 checks a sample of EVERY shard of the gathered vector against the CPU oracle (`gpu_mismatches`).
       Weak scaling: per-GPU work is fixed as N grows (N = 1 runs the configs[1] size the metric is quoted on).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (sw128_kernel), timed live with HIP events on the
launch stream inside the timed region.  `roofline.frac` is a utilisation: VALU issue cycles the kernel needs -- counted
from the disassembly of the libswmi.so that is being timed (tools/isa_census.py: per-class instruction counts x the
measured per-class issue cost, profiles/r01_microbench_valu_rate*.txt) -- divided by the SIMD cycles that elapsed
(kernel time x 1024 SIMDs x 2.4 GHz).  `cpu_baseline` (N = 1 only) is the reference's own simd4
(oracle/_ref/libswref.so, compiled from the reference sources in the build container) timed on ONE host core of this box
on a bounded sample of the same generated pairs, which doubles as a bit-exactness check of the GPU scores.  The oracle /
reference build are used here only as checker and baseline, never as the measured path.
"""
import argparse
import ctypes
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "smith-waterman-simd_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tools"))

CELLS = 128 * 128
INT_OPS_PER_ALIGNMENT = CELLS * 7          # 1 add, 2 sub, 4 max per cell (source.cpp:49-53)
BYTES_PER_ALIGNMENT = 128 + 128 + 4
# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz: a SIMD issues one wave64 VALU instruction per 2 (full rate), 4 (half
# rate: v_max_i32, v_max3_i32, v_dot4, every DPP form ...) or 8 cycles (tools/microbench/valu_rate*.hip).
SIMDS = 256 * 4
CLOCK_HZ = 2.4e9
SIMD_CYCLES_PER_S = SIMDS * CLOCK_HZ       # the VALU issue peak: 2457.6 G SIMD-cycles/s
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 78.6 T full-rate 32-bit lane-ops/s (informational: achieved_algorithmic_tops)
HBM_PEAK_GBS = 8000.0
REFERENCE_PUBLISHED_ALIGN_PER_S = 1e6 / 4.4    # README.md:4 of the reference: simd4 ~4.4 s / 1M on one EPYC 7501 core
CONFIG3_PAIRS_PER_GPU = 1 << 26                # BASELINE.json configs[3]: 512M pairs over 8 GPUs


def issue_bound(kernel_regex, trips, waves, kernel_ms, marker=None, conditional_share=1.0, pick="innermost", second=None):
    """VALU issue-bound utilisation of one launch, derived from the shipped library's disassembly (tools/isa_census.py).

    `trips` = iterations of the main loop AS WRITTEN IN THE SOURCE; `marker` = (mnemonic, count per source iteration):
    hipcc unrolls some instantiations, and the number of marker instructions in the compiled loop body says by how much.
    `conditional_share` = on which share of the iterations the loop's blocks under a scalar condition run (the semi-global
    sweep flushes its records every 16th round).  `pick`: how the marker picks the loop (isa_census.census).
    `second` = {"marker": (mnemonic, per iteration), "exclude": mnemonic, "trips": n}: a SECOND hot loop of the same kernel
    (the semi-global sweeps' calm loop: the one without the X-drop test) that ran n source iterations; `trips` then counts
    the first loop's only.
    Returns the dict that goes into a `roofline` object: achieved / peak in G SIMD-issue-cycles per second and
    frac = achieved / peak <= 1 by construction (a kernel cannot issue more VALU cycles than elapsed)."""
    try:
        import isa_census

        def one(marker, exclude, pick, trips):
            found = isa_census.census_for(kernel_regex, marker_op=marker[0] if marker else None, exclude_op=exclude, pick=pick)
            if len(found) != 1:
                raise RuntimeError("%d kernels match %r" % (len(found), kernel_regex))
            name, c = next(iter(found.items()))
            if marker:
                unroll = (c["main_loop"]["by_op"].get(marker[0], 0)) / float(marker[1])
                if unroll < 1:
                    raise RuntimeError("main loop of %s holds %d %s, fewer than one source iteration's %d" % (
                        name, c["main_loop"]["by_op"].get(marker[0], 0), marker[0], marker[1]))
                trips = trips / unroll
            return name, c, trips
        name, c, trips = one(marker, None, pick, trips)
        c2 = trips2 = None
        if second:
            _, c2, trips2 = one(second["marker"], second.get("exclude"), "most", second["trips"])
    except Exception as e:                  # no llvm-objdump on this box: say so, never invent a fraction
        return {"frac": None, "census_error": repr(e)}
    kernel_s = kernel_ms * 1e-3
    cyc_ideal = isa_census.issue_cycles_per_wave(c, trips, "ideal", conditional_share)
    cyc_meas = isa_census.issue_cycles_per_wave(c, trips, "measured", conditional_share)
    valu = isa_census.valu_instructions_per_wave(c, trips, conditional_share)
    if c2 is not None:                      # the second loop sits in the first census's "outside" part once: take that out, put its trips in
        for key, rates in (("issue_cycles_ideal", "ideal"), ("issue_cycles_measured_rates", "measured")):
            body = c2["main_loop"][key] + conditional_share * c2["main_loop_conditional"][key]
            static = c2["main_loop"][key] + c2["main_loop_conditional"][key]
            if rates == "ideal":
                cyc_ideal += trips2 * body - static
            else:
                cyc_meas += trips2 * body - static
        valu += trips2 * (c2["main_loop"]["valu"] + conditional_share * c2["main_loop_conditional"]["valu"]) - (
            c2["main_loop"]["valu"] + c2["main_loop_conditional"]["valu"])
    achieved = waves * cyc_ideal / kernel_s
    out = {
        "bound": "valu", "kernel": name, "kernel_code_sha256": c["code_sha256"],
        "achieved": round(achieved / 1e9, 1), "peak": round(SIMD_CYCLES_PER_S / 1e9, 1), "unit": "G VALU issue cycles/s (sum over 1024 SIMDs)",
        "frac": round(achieved / SIMD_CYCLES_PER_S, 4),
        "frac_at_measured_instruction_rates": round(waves * cyc_meas / kernel_s / SIMD_CYCLES_PER_S, 4),
        "census": {"source": "tools/isa_census.py on the libswmi.so being timed (class costs: profiles/r01_microbench_valu_rate*.txt)",
                   "main_loop_trips": round(trips, 2), "wavefronts_per_launch": waves,
                   "valu_instructions_per_wavefront": round(valu),
                   "issue_cycles_per_wavefront": round(cyc_ideal, 1),
                   "main_loop_valu_by_op": c["main_loop"]["by_op"],
                   "main_loop_issue_cycles": c["main_loop"]["issue_cycles_ideal"],
                   "main_loop_conditional_issue_cycles": c["main_loop_conditional"]["issue_cycles_ideal"],
                   "conditional_share": conditional_share,
                   "unmeasured_valu_in_main_loop": c["main_loop"]["unmeasured_valu"]},
    }
    if c2 is not None:
        out["census"]["second_loop"] = {"trips": round(trips2, 2), "valu": c2["main_loop"]["valu"], "valu_by_op": c2["main_loop"]["by_op"],
                                        "issue_cycles": c2["main_loop"]["issue_cycles_ideal"],
                                        "conditional_issue_cycles": c2["main_loop_conditional"]["issue_cycles_ideal"]}
    return out


def stamped_profile(kernel_code_sha256):
    """PMC-derived figures (HBM traffic per launch, effective clock) from the committed profile, only if they were taken
    on exactly the kernel code that is being timed now (profiles/traffic.json carries the code hash)."""
    f = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tr = json.load(open(f))
    except Exception:
        return None, "profiles/traffic.json missing"
    if tr.get("kernel_code_sha256") != kernel_code_sha256:
        return None, "profiles/traffic.json was taken on kernel code %s, the library being timed carries %s: not quoted" % (
            tr.get("kernel_code_sha256"), kernel_code_sha256)
    return tr, tr.get("source")


def effective_cores():
    """Cores the container may actually use: the cgroup CPU quota (cpu.max), else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return round(float(quota) / float(period), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return round(q / p, 2)
    except Exception:
        pass
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30, help="untimed steps (the clock takes ~10 launches to settle after idle)")
    ap.add_argument("--len", type=int, default=1024, help="sequence length of the banded-affine mode")
    ap.add_argument("--gap-open", type=int, default=5)
    ap.add_argument("--gap-extend", type=int, default=1)
    ap.add_argument("--mode", default="pairs", choices=["pairs", "packed", "one-vs-many", "banded-affine", "semiglobal"],
                    help="pairs = the headline path (+ rows); the others run ONE secondary row as a line of its own "
                         "(what tools/profile_rows.sh profiles)")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (0 = 1,048,576 on one GPU, 67,108,864 per GPU on several)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per alignment (0 = library default)")
    ap.add_argument("--match", type=int, default=10)
    ap.add_argument("--mismatch", type=int, default=-30)
    ap.add_argument("--gap", type=int, default=15)
    ap.add_argument("--seed", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rows", action="store_true", help="N = 1: headline only (profiling passes use this)")
    ap.add_argument("--sg-plain", action="store_true", help="semi-global mode: the timed calls only -- no exact-path and two-streams "
                    "legs (the counter passes use this: their per-kernel averages then cover one kind of launch)")
    ap.add_argument("--sustained-seconds", type=float, default=2.5)
    ap.add_argument("--force-dist", action="store_true", help="create the process group and run the score gather even "
                    "with one rank (rehearses the RCCL path on a one-GPU box)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; "
                    "gloo only to rehearse the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="pairs in the CPU baseline sample (0 = auto, ~10-15 s)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def cpu_baseline(swmi, np, args, gpu_scores_head, sample):
    """Reference simd4 (or, if the prebuilt reference is absent, the C oracle) on ONE host core."""
    vp = ctypes.c_void_p
    seq1, seq2 = swmi.generate_pairs_host(sample, args.seed, 0)
    sm = swmi.match_matrix(args.match, args.mismatch)
    out = np.zeros(sample, np.int32)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libswref.so")
    info = {}
    if os.path.exists(ref_path):
        ref = ctypes.CDLL(ref_path)
        ref.swref_repeat.restype = ctypes.c_longlong
        t0 = time.perf_counter()
        ref.swref_batch(4, seq1.ctypes.data_as(vp), seq2.ctypes.data_as(vp), ctypes.c_size_t(sample),
                        sm.ctypes.data_as(vp), args.gap, out.ctypes.data_as(vp))
        dt = time.perf_counter() - t0
        # the reference's own harness shape: ONE pair, 1,000,000 calls (source.cpp:3074-3082)
        t1 = time.perf_counter()
        ref.swref_repeat(4, seq1[0].ctypes.data_as(vp), seq2[0].ctypes.data_as(vp), sm.ctypes.data_as(vp), args.gap, 1000000)
        rep_ms = (time.perf_counter() - t1) * 1e3
        sys.stderr.write("simd4 version: %.0f ms / 1M\n" % rep_ms)      # line shape of source.cpp:3081
        info = {"kind": "reference", "function": "SmithWaterman_simd4 (source.cpp:462-571), g++ -O3 -mavx2",
                "repeat_one_pair_ms_per_1M": round(rep_ms, 1)}
        # the same simd4 on ALL host cores (SURVEY 8d iii): contiguous shards, one thread each (ctypes drops the GIL)
        from concurrent.futures import ThreadPoolExecutor
        cores = effective_cores()
        threads = max(1, min(os.cpu_count() or 1, int(round(cores)) if cores else 1))
        bounds = [(sample * t // threads, sample * (t + 1) // threads) for t in range(threads)]
        out_mt = np.zeros(sample, np.int32)

        passes = 4                                         # each thread scores its shard 4 times

        def shard(b):
            lo, hi = b
            for _ in range(passes if hi > lo else 0):
                ref.swref_batch(4, seq1[lo:].ctypes.data_as(vp), seq2[lo:].ctypes.data_as(vp), ctypes.c_size_t(hi - lo),
                                sm.ctypes.data_as(vp), args.gap, out_mt[lo:].ctypes.data_as(vp))
        with ThreadPoolExecutor(threads) as ex:
            t2 = time.perf_counter()
            list(ex.map(shard, bounds))
            dt_mt = time.perf_counter() - t2
        info["all_host_cores"] = {"threads": threads, "effective_cores": cores, "logical_cpus_visible": os.cpu_count(),
                                  "value": round(sample * passes / dt_mt, 1), "unit": "alignments/s",
                                  "agrees_with_one_core": bool((out_mt == out).all()),
                                  "note": "threads = the container's CPU quota (cgroup cpu.max), not os.cpu_count()"}
    else:
        orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        t0 = time.perf_counter()
        orc.sw_oracle_batch_st(seq1.ctypes.data_as(vp), seq2.ctypes.data_as(vp), ctypes.c_size_t(sample),
                               sm.ctypes.data_as(vp), args.gap, out.ctypes.data_as(vp))
        dt = time.perf_counter() - t0
        info = {"kind": "port", "function": "oracle/sw_oracle.c scalar restatement of source.cpp:35-60"}
    mism = int((out != gpu_scores_head[:sample]).sum())
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    info.update({"value": round(sample / dt, 1), "unit": "alignments/s", "cores": 1,
                 "sample": "%d distinct generated pairs (seed %d, pairs 0..%d), same parameters" % (sample, args.seed, sample - 1),
                 "seconds": round(dt, 2), "host_cpu": model, "host_cores_available": effective_cores(),
                 "gpu_scores_checked": sample, "gpu_mismatches": mism})
    return info


def oracle_scores(np, seq1, seq2, sm, gap):
    """CPU checker for a sample: the reference's simd4 when oracle/_ref is present, else the C oracle (OpenMP)."""
    vp = ctypes.c_void_p
    n = seq1.shape[0]
    out = np.zeros(n, np.int32)
    a, b = np.ascontiguousarray(seq1), np.ascontiguousarray(seq2)
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libswref.so")
    in_simd4_domain = int(np.min(sm)) >= -127 and gap >= 0
    if os.path.exists(ref_path) and in_simd4_domain:
        ctypes.CDLL(ref_path).swref_batch(4, a.ctypes.data_as(vp), b.ctypes.data_as(vp), ctypes.c_size_t(n),
                                          sm.ctypes.data_as(vp), int(gap), out.ctypes.data_as(vp))
        return out, "reference simd4 (oracle/_ref)"
    ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")).sw_oracle_batch(
        a.ctypes.data_as(vp), b.ctypes.data_as(vp), ctypes.c_size_t(n), sm.ctypes.data_as(vp), int(gap), out.ctypes.data_as(vp))
    return out, "oracle/sw_oracle.c"


def time_launches(torch, stream, launch, steps, warmup):
    """Average per-launch duration (ms) of `steps` back-to-back launches, HIP events on the launch stream."""
    for _ in range(warmup):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(steps):
        launch()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / steps


def bench_banded(args, swmi, np, torch, local_rank, steps=None, warmup=None):
    """Secondary row: 1024 x 1024 affine gap, 128-diagonal band, one wavefront per alignment (single GPU)."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    length = args.len
    P = args.pairs if args.pairs and args.mode == "banded-affine" else 1 << 16
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    npairs128 = P * length // 128               # reuse the 128-mer generator: P len-mers = P*len/128 consecutive 128-mers
    d1 = torch.empty(P * length, dtype=torch.uint8, device=dev)
    d2 = torch.empty(P * length, dtype=torch.uint8, device=dev)
    swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), npairs128, args.seed, 0, stream.cuda_stream)
    # make seq2 a mutated copy of seq1 (~8 % substitutions) so that alignments are long: random-vs-random would idle in H = 0
    mut = torch.rand(P * length, device=dev) < 0.08
    d2 = torch.where(mut, d2, d1).contiguous()
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    sm = swmi.match_matrix(2, -3)

    def launch():
        swmi.score_banded_affine_device(d1.data_ptr(), d2.data_ptr(), P, length, sm, args.gap_open, args.gap_extend,
                                        scores.data_ptr(), stream.cuda_stream)
    for _ in range(warmup):
        launch()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream); launch(); b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / steps
    band_cells = sum(min(length, i + 63) - max(1, i - 64) + 1 for i in range(1, length + 1))
    value = P * steps / elapsed
    # main loop = `length` iterations of two anti-diagonal steps; one wavefront per alignment (sw_banded_affine_kernel, int32
    # cell) or per TWO alignments (sw_banded_affine_pk_kernel, 16-bit halves): the library says which it launched
    kname, per_wave = swmi.banded_affine_kernel_for(length, sm, args.gap_open, args.gap_extend)
    packed_cell = "_pk_" in kname
    roof = issue_bound("^" + re.escape(kname) + "$", length, (P + per_wave - 1) // per_wave, kernel_ms,
                       marker=("v_perm_b32", 2) if packed_cell else ("v_dot4_i32_i8", 2))
    # 12 algorithmic int ops per cell: 4 sub, 2+3 max, 1 add, 1 running max, 1 lookup (informational)
    roof.update({"kernel_ms": round(kernel_ms, 4), "traffic": None,
                 "achieved_algorithmic_tops": round(P * band_cells * 12 / (kernel_ms * 1e-3) / 1e12, 3),
                 "gcups_kernel": round(P * band_cells / (kernel_ms * 1e-3) / 1e9, 1)})
    line = {"metric": "alignments/sec (and GCUPS), banded affine extension", "value": round(value, 1), "unit": "alignments/s",
            "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed * 1e3 / steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16" if packed_cell else "int32", "data": "synthetic",
            "gcups": round(value * band_cells / 1e9, 1),
            "config": {"workload": "BASELINE.json configs[4] (extension, parity unpinned by the reference): %d pairs of %d-mers, "
                                   "128-diagonal band, sm 2/-3, gap open %d extend %d, inputs resident in HBM" % (
                                       P, length, args.gap_open, args.gap_extend), "band_cells_per_alignment": band_cells},
            "roofline": roof,
            "checksum": int(scores.to(torch.int64).sum().item())}
    if not args.no_cpu_baseline:
        sample = 256
        a = d1[: sample * length].cpu().numpy().reshape(sample, length)
        b = d2[: sample * length].cpu().numpy().reshape(sample, length)
        orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        orc.sw_oracle_banded_affine.restype = ctypes.c_int
        vp = ctypes.c_void_p
        t0 = time.perf_counter()
        want = np.array([orc.sw_oracle_banded_affine(a[k].ctypes.data_as(vp), b[k].ctypes.data_as(vp), length,
                                                     sm.ctypes.data_as(vp), args.gap_open, args.gap_extend) for k in range(sample)], np.int32)
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"kind": "port", "function": "oracle/sw_oracle.c sw_oracle_banded_affine (scalar, full tables); parity unpinned by the reference",
                                "value": round(sample / dt, 1), "unit": "alignments/s", "cores": 1,
                                "sample": "first %d pairs of the batch" % sample,
                                "gpu_mismatches": int((want != scores[:sample].cpu().numpy()).sum())}
    return line


def sg_traffic(P, kernel, code_sha):
    """HBM bytes per launch of the sweep kernel from the committed PMC passes (65536 alignments), or None; quoted only
    when the profile was taken on the same kernel code."""
    for name in ("r04_semiglobal_pmc.json", "r03_semiglobal_pmc.json", "r02_semiglobal_pmc.json", "r01_semiglobal_pmc.json"):
        f = os.path.join(ROOT, "profiles", name)
        if P != 65536 or not os.path.exists(f):
            continue
        try:
            doc = json.load(open(f))
            k = doc.get(kernel, {})
            if doc.get("kernel_code_sha256", {}).get(kernel) != code_sha:
                continue
            return int(k["hbm_read_bytes_x2"] + k["hbm_write_bytes"]), "profiles/" + name
        except Exception:
            continue
    return None, None


def arith_dtype(kernel_name):
    """The arithmetic type a kernel computes in (the line's `dtype`): the packed scorer and the semi-global sweeps work on
    unsigned 16-bit halves of 32-bit registers (scores leave as int32), the other kernels on int32."""
    k = (kernel_name or "").replace(" ", "")
    return "u16" if k.startswith(("sw128_pk_kernel", "sg_forward_")) else "int32"


def sg_sweep_shape(kernel_name):
    """(alignments per wavefront, census marker) of a semi-global sweep kernel as swmi_semiglobal_kernels_for_batch names it:
    64 / 32 / 16 alignments per wavefront with the band in 1 / 2 / 4 lanes; the sweeps' exact loop holds eight rounds and the
    X-drop test's one v_pk_ashrrev_i16 per register (two cells) marks a round (the calm loop has none: bench_semiglobal)."""
    name = kernel_name.replace(" ", "")
    if name.startswith("sg_forward_lane_kernel<"):
        return 64, ("v_pk_ashrrev_i16", 16)
    if name.startswith("sg_forward_split_kernel<2,"):
        return 32, ("v_pk_ashrrev_i16", 8)
    if name.startswith("sg_forward_split_kernel<4,"):
        return 16, ("v_pk_ashrrev_i16", 4)
    raise ValueError("unknown semi-global sweep kernel %r" % kernel_name)


def bench_semiglobal(args, swmi, np, torch, local_rank, steps=None, warmup=None):
    """Secondary row: the reference's semi-global adaptive-band X-drop aligner incl. traceback (single GPU).
    Inputs follow SpeedtestSemiGlobal (source.cpp:2805-2813): a random 16384-mer and a copy with 5 % substitutions."""
    L = 16384
    P = args.pairs if args.pairs and args.mode == "semiglobal" else 65536
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    g = torch.Generator(device=dev); g.manual_seed(args.seed)
    d1 = torch.randint(0, 4, (P, L), dtype=torch.uint8, device=dev, generator=g)
    rnd = torch.randint(0, 4, (P, L), dtype=torch.uint8, device=dev, generator=g)
    keep = torch.rand((P, L), device=dev, generator=g) < 0.95
    d2 = torch.where(keep, d1, rnd).contiguous()
    del rnd, keep
    cap = 32769
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    lengths = torch.empty(P, dtype=torch.int32, device=dev)
    tb = torch.empty((P, cap, 2), dtype=torch.int32, device=dev)

    def launch():
        swmi.semiglobal_xdrop_device(d1.data_ptr(), d2.data_ptr(), P, scores.data_ptr(), tb.data_ptr(), cap, lengths.data_ptr(),
                                     stream.cuda_stream)
    steps = min(args.steps, 10) if steps is None else steps
    warm = min(args.warmup, 2) if warmup is None else warmup
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        launch()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    value = P * steps / elapsed
    line = {"metric": "alignments/sec, semi-global adaptive-band X-drop with traceback (SURVEY 8f N4)", "value": round(value, 1),
            "unit": "alignments/s", "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": round(elapsed * 1e3 / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "SemiGlobal_AdaptiveBanded_XDrop_111_32_70 (source.cpp:1836): %d pairs of 16384-mers, 5 %% "
                                   "substitutions (SpeedtestSemiGlobal inputs), band 32, X-drop 70, score + full traceback, "
                                   "inputs resident in HBM" % P},
            "mean_score": float(scores.float().mean().item()), "mean_traceback_len": float(lengths.float().mean().item())}
    # per-kernel durations from HIP events on the launch stream (swmi_semiglobal_time_device), averaged over 3 calls
    phases = [swmi.semiglobal_time_device(d1.data_ptr(), d2.data_ptr(), P, scores.data_ptr(), tb.data_ptr(), cap,
                                          lengths.data_ptr(), stream.cuda_stream) for _ in range(3)]
    sweep_ms = sum(p[0] for p in phases) / 3
    tb_ms = sum(p[1] for p in phases) / 3
    # algorithmic work of the sweep (source.cpp:1914-1941): per band cell 1 lookup+add, 2 sub, 3 max into the cell, 1 max
    # into the round maximum, compare+select of the X-drop = 9 int ops; 32 cells per round; these inputs never drop out,
    # so every alignment runs the full 32768 rounds
    rounds, ops_cell = 32768, 9
    sweep_ops = P * rounds * 32 * ops_cell
    alg_bytes = P * (2 * L + 8) + int(lengths.to(torch.int64).sum().item()) * 8
    sweep_kernel, tb_kernel = swmi.semiglobal_kernels_for_batch(P)          # the library says which mapping it ran
    name = sweep_kernel.replace(" ", "")
    per_wave, marker = sg_sweep_shape(name)
    # Two loops of eight unrolled rounds each: the exact one (X-drop test: the marker) and the calm one (sg_kernels.hip: windows in
    # which no cell can reach the threshold); the library counts how many windows of the last call were calm.  The record
    # flush and the stream top-up each sit behind ONE of a loop's eight rounds under a test of the window's number: they run on
    # every second trip.
    windows, calm_windows, walk_windows, walk_again = swmi.semiglobal_window_stats(stream.cuda_stream, walk=True)
    calm_share = calm_windows / windows if windows else 0.0
    roof = issue_bound("^" + name + "$", rounds * (1.0 - calm_share), (P + per_wave - 1) // per_wave, sweep_ms, marker=marker,
                       conditional_share=0.5, pick="most",
                       second={"marker": ("v_pk_maximum3_f16", marker[1] * 3 // 2), "exclude": marker[0], "trips": rounds * calm_share})
    roof["calm_window_share"] = round(calm_share, 4)
    # the traceback fetches the records of band cells 8 .. 23 only; a window in which a walk leaves them is decoded a second time
    roof["walk_windows_decoded_twice_share"] = round(walk_again / walk_windows, 4) if walk_windows else None
    traffic, traffic_src = sg_traffic(P, sweep_kernel, roof.get("kernel_code_sha256"))
    roof.update({
        "kernel_ms": round(sweep_ms, 3), "traceback_kernels": tb_kernel, "traceback_kernel_ms": round(tb_ms, 3),
        "achieved_algorithmic_tops": round(sweep_ops / (sweep_ms * 1e-3) / 1e12, 3),
        "algorithmic_ops_per_launch": sweep_ops, "gcups_kernel": round(P * rounds * 32 / (sweep_ms * 1e-3) / 1e9, 1),
        "traffic": traffic, "traffic_source": traffic_src,
        "kernel_ms_covers": "sweep phase between HIP events: the stream-packing pre-pass (~0.6 ms at 65536) + the sweep kernel",
        "hbm": {"algorithmic_bytes_per_launch": alg_bytes, "achieved": round(alg_bytes / ((sweep_ms + tb_ms) * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / ((sweep_ms + tb_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "sequences in + (i, j) pairs out; the predecessor records between the two kernels are implementation "
                        "traffic on top"}})
    line["roofline"] = roof
    # the same call with every round on the exact path (what a batch of alignments that hover at their X-drop thresholds runs)
    forced = os.environ.get("SWMI_SG_EXACT") == "1"
    if not getattr(args, "sg_plain", False):
        swmi.semiglobal_set_exact(True)
        try:
            ph = [swmi.semiglobal_time_device(d1.data_ptr(), d2.data_ptr(), P, scores.data_ptr(), tb.data_ptr(), cap,
                                              lengths.data_ptr(), stream.cuda_stream) for _ in range(3)]
        finally:
            swmi.semiglobal_set_exact(forced)
        ex_sweep, ex_tb = sum(p[0] for p in ph) / 3, sum(p[1] for p in ph) / 3
        line["exact_path_forced"] = {"sweep_ms": round(ex_sweep, 3), "traceback_ms": round(ex_tb, 3),
                                     "value": round(P / ((ex_sweep + ex_tb) * 1e-3), 1), "unit": "alignments/s",
                                     "note": "swmi_semiglobal_set_exact(1): no calm windows, the X-drop test in every round; same results"}
    # Two calls in flight on two streams (the library keeps a workspace per stream: tests/test_gpu_multi.py): the second
    # call's sweep shares the SIMDs with the first one's -- two wavefronts per SIMD issue 2 instructions per ~4.5 cycles where
    # one issues 1 per ~5 -- and the tracebacks run under the other call's sweep.  Same per-call batch, same results; a
    # separate figure, never the row's `value` (which is one call at a time).
    if P * 2 * (2 * L + cap * 8 + 16) < 200e9 and not getattr(args, "sg_plain", False):
        s2 = torch.cuda.Stream()
        d1b, d2b = d1.clone(), d2.clone()
        scores_b, lengths_b = torch.empty_like(scores), torch.empty_like(lengths)
        tb_b = torch.empty_like(tb)
        torch.cuda.synchronize()

        def launch_pair():
            swmi.semiglobal_xdrop_device(d1.data_ptr(), d2.data_ptr(), P, scores.data_ptr(), tb.data_ptr(), cap, lengths.data_ptr(), stream.cuda_stream)
            swmi.semiglobal_xdrop_device(d1b.data_ptr(), d2b.data_ptr(), P, scores_b.data_ptr(), tb_b.data_ptr(), cap, lengths_b.data_ptr(), s2.cuda_stream)
        launch_pair()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pairs_of_calls = max(3, steps // 2)
        for _ in range(pairs_of_calls):
            launch_pair()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        line["two_calls_in_flight"] = {"value": round(2 * P * pairs_of_calls / dt2, 1), "unit": "alignments/s",
                                       "ms_per_pair_of_calls": round(dt2 * 1e3 / pairs_of_calls, 3),
                                       "same_results_on_both_streams": bool(torch.equal(scores, scores_b) and torch.equal(lengths, lengths_b) and
                                                                             torch.equal(tb[:: max(1, P // 256), :2048], tb_b[:: max(1, P // 256), :2048])),
                                       "note": "two calls of %d alignments on two streams at once; not the row's value" % P}
        del d1b, d2b, scores_b, lengths_b, tb_b, s2
        torch.cuda.empty_cache()
    if not args.no_cpu_baseline:
        # the host-buffer entries (PCIe inclusive; never `value`) on the first 16384 alignments of the batch: the traceback as
        # the reference's (i, j) list (up to 262 KB per alignment back over the link) and as the walk's 2-bit moves (8 KB)
        m = min(P, 16384)
        h1, h2 = d1[:m].cpu().numpy(), d2[:m].cpu().numpy()
        lib = swmi.load()                               # the C entries themselves, output buffers allocated (and touched) beforehand
        h_scores, h_len = np.zeros(m, np.int32), np.zeros(m, np.uint32)
        h_tb = np.zeros((m, cap, 2), np.int32)
        h_moves = np.zeros((m, swmi.SG_MOVE_WORDS), np.uint64)
        host = {"alignments": m, "stat": "second of two calls, output buffers allocated beforehand"}
        calls = (("positions", "swmi_semiglobal_xdrop", lambda: lib.swmi_semiglobal_xdrop(
                      h1.ctypes.data, h2.ctypes.data, m, h_scores.ctypes.data, h_tb.ctypes.data, cap, h_len.ctypes.data)),
                 ("moves", "swmi_semiglobal_xdrop_moves", lambda: lib.swmi_semiglobal_xdrop_moves(
                      h1.ctypes.data, h2.ctypes.data, m, h_scores.ctypes.data, h_moves.ctypes.data, h_len.ctypes.data)))
        for label, entry, call in calls:
            h_scores[:] = -1
            if call() != 0:
                raise RuntimeError(entry + ": " + swmi.last_error())
            t1 = time.perf_counter()
            call()
            dt1 = time.perf_counter() - t1
            host[label] = {"entry": entry, "ms": round(dt1 * 1e3, 2), "value": round(m / dt1, 1), "unit": "alignments/s",
                           "scores_match_resident": bool((h_scores == scores[:m].cpu().numpy()).all())}
        host["positions"]["rows_match_resident"] = all(
            np.array_equal(h_tb[k, : int(h_len[k])], tb[k, : int(h_len[k])].cpu().numpy()) for k in (0, m // 2, m - 1))
        host["moves"]["expanded_rows_match_resident"] = all(   # move rows expanded on the host against the positions the device entry wrote
            np.array_equal(swmi.semiglobal_expand_moves(h_moves[k], int(h_len[k])), tb[k, : int(h_len[k])].cpu().numpy()) for k in (0, m // 2, m - 1))
        del h_tb, h_moves
        line["host_buffer_path"] = host
        del h1, h2
        ref_path = os.path.join(ROOT, "oracle", "_ref", "libswref.so")
        sample = 64
        a = d1[:sample].cpu().numpy(); b = d2[:sample].cpu().numpy()
        vp = ctypes.c_void_p
        tbh = np.zeros((40000, 2), np.int32)
        got_scores = scores[:sample].cpu().numpy(); got_len = lengths[:sample].cpu().numpy(); got_tb = tb[:sample].cpu().numpy()
        mism = 0
        if os.path.exists(ref_path):
            ref = ctypes.CDLL(ref_path)
            res = {}
            for variant, name in ((1, "simd"), (4, "simd_mark4")):
                t0 = time.perf_counter()
                for k in range(sample):
                    sc, ln = ctypes.c_int32(), ctypes.c_size_t()
                    ref.swref_semiglobal(variant, a[k].ctypes.data_as(vp), b[k].ctypes.data_as(vp), ctypes.byref(sc),
                                         tbh.ctypes.data_as(vp), ctypes.c_size_t(40000), ctypes.byref(ln))
                    if variant == 4:
                        ok = sc.value == got_scores[k] and ln.value == got_len[k] and np.array_equal(tbh[: ln.value], got_tb[k, : ln.value])
                        mism += 0 if ok else 1
                res[name] = sample / (time.perf_counter() - t0)
            line["cpu_baseline"] = {"kind": "reference", "function": "SemiGlobal_AdaptiveBanded_XDrop_111_32_70_simd_mark4 (source.cpp:2543), g++ -O3 -mavx2",
                                    "value": round(res["simd_mark4"], 1), "unit": "alignments/s", "cores": 1,
                                    "simd_variant_alignments_per_s": round(res["simd"], 1),
                                    "sample": "first %d pairs of the batch (score + whole traceback compared)" % sample, "gpu_mismatches": mism}
        else:
            orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
            t0 = time.perf_counter()
            for k in range(8):
                sc, ln, oob = ctypes.c_int32(), ctypes.c_size_t(), ctypes.c_int()
                orc.sg_oracle_xdrop(a[k].ctypes.data_as(vp), b[k].ctypes.data_as(vp), ctypes.byref(sc), tbh.ctypes.data_as(vp),
                                    ctypes.c_size_t(40000), ctypes.byref(ln), ctypes.byref(oob))
                ok = sc.value == got_scores[k] and ln.value == got_len[k] and np.array_equal(tbh[: ln.value], got_tb[k, : ln.value])
                mism += 0 if ok else 1
            line["cpu_baseline"] = {"kind": "port", "function": "oracle/sg_oracle.c", "value": round(8 / (time.perf_counter() - t0), 1),
                                    "unit": "alignments/s", "cores": 1, "sample": "first 8 pairs of the batch", "gpu_mismatches": mism}
    return line


def row_summary(line):
    """The part of a secondary line that goes into the headline's `rows` object."""
    r = line["roofline"]
    out = {"value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"], "steps": line["steps"],
           "kernel": r.get("kernel"), "kernel_ms": r.get("kernel_ms"), "frac": r.get("frac"),
           "frac_at_measured_instruction_rates": r.get("frac_at_measured_instruction_rates"),
           "gcups_kernel": r.get("gcups_kernel"), "workload": line["config"]["workload"]}
    for k in ("traceback_kernel_ms", "traffic"):
        if r.get(k) is not None:
            out[k] = r[k]
    cb = line.get("cpu_baseline")
    if cb:
        out["gpu_mismatches"] = cb["gpu_mismatches"]
        out["checked_against"] = "%s, %s" % (cb["function"], cb["sample"])
        out["cpu_one_core"] = cb["value"]
    if line.get("host_buffer_path"):
        out["host_buffer_path"] = line["host_buffer_path"]
    if line.get("two_calls_in_flight"):
        out["two_calls_in_flight"] = line["two_calls_in_flight"]
    if line.get("exact_path_forced"):
        out["exact_path_forced"] = line["exact_path_forced"]
    for k in ("calm_window_share", "walk_windows_decoded_twice_share"):
        if r.get(k) is not None:
            out[k] = r[k]
    return out


def sw128_roofline(swmi, P, sched_lanes, sched_flags, mode, kernel_ms, match, mismatch, gap):
    """`roofline` object of one scoring launch over P pairs under the schedule setting (sched_lanes, sched_flags) -- 0 lanes
    = automatic.  The library says which kernel instantiation such a launch runs (swmi_score_kernel_for_batch)."""
    import re
    mode_id = {"pairs": 0, "packed": 1, "one-vs-many": 2}[mode]
    cur = swmi.get_schedule()
    swmi.set_schedule(sched_lanes, sched_flags)
    try:
        kernel, per_wave = swmi.score_kernel_for_batch(P, swmi.match_matrix(match, mismatch), gap, mode_id)
    finally:
        swmi.set_schedule(*cur)
    packed_kernel = kernel.startswith("sw128_pk_kernel")          # sw128_pk_kernel<MODE,VARIANT> (L = 4) or <MODE,VARIANT,L>
    if packed_kernel:
        targs = re.search(r"<([\d,]+)>", kernel).group(1).split(",")
        lanes = int(targs[2]) if len(targs) > 2 else 4
    else:
        lanes = int(re.search(r"<(\d+)", kernel).group(1))
    # pairs of anti-diagonal steps in the main loop: the int32 kernel rounds T = 128 + L - 1 up, the packed kernel runs the odd
    # last step after the loop (the census counts it with the instructions outside the loop)
    trips = (127 + lanes) // 2 if packed_kernel else (128 + lanes) // 2
    waves = (P + per_wave - 1) // per_wave
    kernel_s = kernel_ms * 1e-3
    # what marks one source iteration in the compiled loop: one v_perm per PAIR of cells (packed kernel: 2 steps x R rows),
    # one v_dot4 per cell (int32 kernel: 2 steps x R rows)
    marker = ("v_perm_b32" if packed_kernel else "v_dot4_i32_i8", 2 * (128 // lanes))
    roof = issue_bound("^" + re.escape(kernel) + "$", trips, waves, kernel_ms, marker=marker)
    roof["lanes_per_alignment"] = lanes
    bytes_per_alignment = {"pairs": BYTES_PER_ALIGNMENT, "packed": 32 + 32 + 4, "one-vs-many": 128 + 4}[mode]
    roof.update({
        "kernel_ms": round(kernel_ms, 4), "traffic": None,
        "achieved_algorithmic_tops": round(P * INT_OPS_PER_ALIGNMENT / kernel_s / 1e12, 3),
        "algorithmic_note": "7 int ops per cell (SURVEY 8d) against the 78.6 T full-rate lane-op/s peak would read %.2f: fused "
                            "instructions (v_dot4 = lookup + add, v_max3 = two max) retire several algorithmic ops per issue "
                            "slot, so that ratio is not a utilisation; `frac` is" % (P * INT_OPS_PER_ALIGNMENT / kernel_s / 1e12 / VALU_PEAK_TOPS),
        "algorithmic_ops_per_launch": P * INT_OPS_PER_ALIGNMENT,
        "algorithmic_bytes_per_launch": P * bytes_per_alignment,
        "hbm": {"achieved": round(P * bytes_per_alignment / kernel_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(P * bytes_per_alignment / kernel_s / 1e9 / HBM_PEAK_GBS, 5)},
        "gcups_kernel": round(P * CELLS / kernel_s / 1e9, 1)})
    return roof


def single_gpu(args, swmi, np, torch, local_rank):
    """N = 1: the headline (configs[1]) + rows + sustained."""
    P = args.pairs or (1 << 20)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    if args.lanes:
        swmi.set_schedule(args.lanes, 0)
    lanes, flags = swmi.get_schedule()
    lanes = lanes or swmi.schedule_for_batch(P)
    d1 = torch.empty(P * 128, dtype=torch.uint8, device=dev)
    d2 = torch.empty(P * 128, dtype=torch.uint8, device=dev)
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), P, args.seed, 0, stream.cuda_stream)
    sm = swmi.match_matrix(args.match, args.mismatch)
    p1 = p2 = None
    if args.mode == "packed":                   # pack on the device with torch (input preparation, outside the timed region)
        def pack(d):
            v = d.view(P, 32, 4).to(torch.int32)
            return (v[..., 0] | (v[..., 1] << 2) | (v[..., 2] << 4) | (v[..., 3] << 6)).to(torch.uint8).contiguous()
        p1, p2 = pack(d1), pack(d2)

    def launch(mode=None, out=None):
        mode = mode or args.mode
        out_ptr = (out if out is not None else scores).data_ptr()
        if mode == "packed":
            swmi.score_batch_device(p1.data_ptr(), p2.data_ptr(), P, sm, args.gap, out_ptr, stream.cuda_stream, packed=True)
        elif mode == "one-vs-many":
            swmi.score_one_vs_many_device(d1.data_ptr(), P, d2.data_ptr(), sm, args.gap, out_ptr, stream.cuda_stream)
        else:
            swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), P, sm, args.gap, out_ptr, stream.cuda_stream)

    # settle the clocks first: after idle the first ~10 launches run 10 % slower (DVFS ramp); extra untimed work on top of
    # the W warmup steps the caller asked for, so that a small W still measures the steady state
    for _ in range(40 + args.warmup):
        launch()
    # HIP events bracket RUNS of `stride` launches of the timed region, one pair per run, every launch inside some run: the
    # kernel time per launch is (sum of the runs) / steps, which cannot exceed ms_per_step (the same launches plus whatever the
    # event records cost between runs).  Round 2 bracketed every 8th launch with a pair of its own, whose ~7 us of event
    # overhead landed INSIDE the bracket: kernel_ms read 0.6 % above ms_per_step.
    stride = max(1, int(os.environ.get("SWMI_BENCH_EVENT_STRIDE", "8")))
    runs = [(k, min(k + stride, args.steps)) for k in range(0, args.steps, stride)]
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in runs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (lo, hi), (e_a, e_b) in zip(runs, events):
        e_a.record(stream)
        for _ in range(lo, hi):
            launch()
        e_b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / args.steps     # HIP events on the launch stream
    value = P * args.steps / elapsed
    roof = sw128_roofline(swmi, P, args.lanes, flags, args.mode, kernel_ms, args.match, args.mismatch, args.gap)
    prof, prof_src = stamped_profile(roof.get("kernel_code_sha256"))
    if prof and prof.get("pairs_per_launch") == P and args.mode == "pairs":
        roof["traffic"] = prof.get("hbm_bytes_per_launch")
        roof["traffic_source"] = prof_src
        if prof.get("effective_clock_ghz"):
            roof["effective_clock_ghz"] = prof["effective_clock_ghz"]
            roof["effective_clock_source"] = "GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same profile (profiled passes clock a little lower than un-profiled ones)"
            if roof.get("frac") is not None:
                roof["frac_at_effective_clock"] = round(roof["frac"] * CLOCK_HZ / (prof["effective_clock_ghz"] * 1e9), 4)
    else:
        roof["traffic_note"] = prof_src
    checksum = int(scores.to(torch.int64).sum().item())
    line = {
        "metric": "alignments/sec (and GCUPS) on 1M fixed-length pairs, 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "alignments/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": round(value / REFERENCE_PUBLISHED_ALIGN_PER_S, 1),
        "baseline_note": "reference README.md:4: simd4 ~4.4 s / 1M calls on one EPYC 7501 core (227k alignments/s)",
        "dtype": arith_dtype(roof.get("kernel")), "data": "synthetic",
        "gcups": round(value * CELLS / 1e9, 1),
        "config": {"workload": "%s: %d random 128x128 pairs per GPU per step, sm %d/%d gap %d, inputs resident in HBM, int32 scores" % (
                       {"pairs": "BASELINE.json configs[1]", "packed": "SURVEY 8f N3 (2-bit packed inputs, source.cpp:1581)",
                        "one-vs-many": "SURVEY 8f N1 (every seq1 vs ONE seq2, source.cpp:1227)"}[args.mode],
                       P, args.match, args.mismatch, args.gap),
                   "pairs_per_gpu": P, "global_pairs": P, "lanes_per_alignment": lanes, "schedule_flags": flags,
                   "parallelism": "batch-sharded x1"},
        "roofline": roof, "checksum": checksum,
    }
    if args.mode != "pairs":
        a_h, b_h = swmi.generate_pairs_host(4096, args.seed, 0)
        if args.mode == "one-vs-many":
            b_h = np.repeat(b_h[:1], 4096, axis=0)
        want, who = oracle_scores(np, a_h, b_h, sm, args.gap)
        line["cpu_baseline"] = {"kind": "reference" if "reference" in who else "port", "function": who, "sample": "first 4096 pairs",
                                "gpu_mismatches": int((want != scores[:4096].cpu().numpy()).sum()), "value": None, "unit": "alignments/s", "cores": 1}
        return line
    head_scores = scores.cpu().numpy()
    if not args.no_cpu_baseline:
        sample = args.cpu_sample or min(P, 1 << 20)
        line["cpu_baseline"] = cpu_baseline(swmi, np, args, head_scores, min(sample, P))
        line["gpu_over_cpu_core"] = round(value / line["cpu_baseline"]["value"], 1)
        # end-to-end through the host-buffer entry point (H2D + kernel + D2H; SURVEY 8d "reported separately"; never `value`):
        # the batch the reference's harness shape produces (1M pairs) and four times that, pageable host memory
        h1, h2 = swmi.generate_pairs_host(4 * P if P <= (1 << 20) else P, args.seed, 0)
        hb = {"entry": "swmi_score_batch (pageable host memory, PCIe inclusive; granules: swmi_host_granules_for)",
              "stat": "median of 7 timed calls after two untimed (`ms`, `value`); `ms_min` = the fastest of the seven; the result "
                      "array is the caller's and is reused (a fresh one's pages are first touched by the copy that fills them: "
                      "+0.2 ms per 1 M scores, the Python wrapper's doing, not the entry's)"}
        host_out = np.zeros(h1.shape[0], np.int32)
        host_out[:] = 1                                   # (touched)

        def timed_host(call, m, bytes_per_pair, entry_id, want):
            call()
            call()
            times = []
            for _ in range(7):
                t3 = time.perf_counter()
                hs = call()
                times.append(time.perf_counter() - t3)
            times.sort()
            med = times[3]
            g = swmi.host_granules(int(m), entry_id)
            return {"pairs": int(m), "ms": round(med * 1e3, 3), "ms_min": round(times[0] * 1e3, 3), "value": round(m / med, 1),
                    "unit": "alignments/s", "h2d_gb_per_s": round(m * bytes_per_pair / med / 1e9, 1),
                    "granules": g if len(g) <= 12 else "%d granules: %s ... %s" % (len(g), g[:3], g[-3:]),
                    "matches_resident_scores": bool((hs[:len(want)] == want[:len(hs)]).all())}
        for label, m in (("pairs_1x", P), ("pairs_4x", h1.shape[0])):
            if label == "pairs_4x" and m == P:
                continue
            hb[label] = timed_host(lambda: swmi.score_batch(h1[:m], h2[:m], sm, args.gap, out=host_out[:m]), m, 256, swmi.ENTRY_PAIRS, head_scores)
        # the same pairs through the reference's 2-bit wire format (source.cpp:1581; 64 B per pair over the link) and the
        # one-vs-many entry (128 B per pair): each entry has its own granule schedule (swmi_api.cpp next_granule)
        p1, p2 = swmi.pack(h1), swmi.pack(h2)
        for label, m in (("pairs_1x", P), ("pairs_4x", h1.shape[0])):
            if label == "pairs_4x" and m == P:
                continue
            hb.setdefault("packed", {"entry": "swmi_score_batch_packed (2-bit inputs, 64 B per pair over the link)"})[label] = timed_host(
                lambda: swmi.score_batch_packed(p1[:m], p2[:m], sm, args.gap, out=host_out[:m]), m, 64, swmi.ENTRY_PACKED, head_scores)
        ovm_want, _ = oracle_scores(np, h1[:4096], np.repeat(h2[:1], 4096, axis=0), sm, args.gap)
        hb["one_vs_many"] = {"entry": "swmi_score_one_vs_many (128 B per pair over the link; checked against the CPU checker on 4096)",
                             "pairs_1x": timed_host(lambda: swmi.score_one_vs_many(h1[:P], h2[0], sm, args.gap, out=host_out[:P]), P, 128,
                                                    swmi.ENTRY_ONE_VS_MANY, ovm_want)}
        del p1, p2
        hb.update({"ms": hb["pairs_1x"]["ms"], "value": hb["pairs_1x"]["value"], "unit": "alignments/s",
                   "matches_resident_scores": all(v["matches_resident_scores"] for k, v in hb.items() if k.startswith("pairs_")) and
                                              all(v["matches_resident_scores"] for k, v in hb["packed"].items() if k.startswith("pairs_")) and
                                              hb["one_vs_many"]["pairs_1x"]["matches_resident_scores"]})
        line["host_buffer_path"] = hb
        del h1, h2
    if args.no_rows:
        return line

    # ---- sustained: >= 2 s of back-to-back launches of the headline kernel (so that a sampling monitor sees the GPU busy)
    n_sus = max(args.steps, int(args.sustained_seconds / (kernel_ms * 1e-3)) + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_sus):
        launch()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    line["sustained"] = {"launches": n_sus, "seconds": round(dt, 3), "value": round(P * n_sus / dt, 1), "unit": "alignments/s",
                         "ms_per_step": round(dt * 1e3 / n_sus, 4),
                         "checksum_unchanged": int(scores.to(torch.int64).sum().item()) == checksum}

    # ---- rows: the other rows of SURVEY.md section 8, short runs, same contract --------------------------------------
    rows = {}
    out2 = torch.empty(P, dtype=torch.int32, device=dev)
    a_h, b_h = swmi.generate_pairs_host(4096, args.seed, 0)

    def sw_row(name, mode, lanes_row, workload):
        swmi.set_schedule(lanes_row, 0)
        try:
            ms = time_launches(torch, stream, lambda: launch(mode, out2), 20, 5)
        finally:
            swmi.set_schedule(args.lanes, 0)
        got = out2.cpu().numpy()
        r = sw128_roofline(swmi, P, lanes_row, 0, mode, ms, args.match, args.mismatch, args.gap)
        row = {"value": round(P / (ms * 1e-3), 1), "unit": "alignments/s", "kernel": r.get("kernel"), "kernel_ms": r["kernel_ms"],
               "frac": r.get("frac"), "frac_at_measured_instruction_rates": r.get("frac_at_measured_instruction_rates"),
               "gcups_kernel": r["gcups_kernel"], "steps": 20, "workload": workload}
        if mode == "one-vs-many":
            want, who = oracle_scores(np, a_h, np.repeat(b_h[:1], 4096, axis=0), sm, args.gap)
            row["gpu_mismatches"] = int((want != got[:4096]).sum())
            row["checked_against"] = "%s, first 4096 sequences" % who
        else:                                   # same pairs as the headline, whose 1M scores the reference's simd4 has just checked
            row["gpu_mismatches"] = int((got != head_scores).sum())
            row["checked_against"] = "all %d scores of the headline batch (themselves compared with the reference's simd4 above)" % P
        rows[name] = row

    sw_row("one_wavefront_per_alignment_L64", "pairs", 64,
           "configs[1] pairs with north_star's literal mapping: one 64-lane wavefront per alignment (swmi_set_schedule(64))")
    def pack(d):
        v = d.view(P, 32, 4).to(torch.int32)
        return (v[..., 0] | (v[..., 1] << 2) | (v[..., 2] << 4) | (v[..., 3] << 6)).to(torch.uint8).contiguous()
    p1, p2 = pack(d1), pack(d2)
    sw_row("packed_2bit_input", "packed", 0, "SURVEY 8f N3: the same pairs in the reference's 2-bit wire format (source.cpp:1581), 68 B per pair")
    sw_row("one_vs_many", "one-vs-many", 0, "SURVEY 8f N1: every seq1 against ONE seq2 (source.cpp:1227)")
    del p1, p2, out2
    # N2 (fixed (1,-1,1) scorer, source.cpp:1073-1225): the general kernel with those parameters
    sm111 = swmi.match_matrix(1, -1)
    ms111 = time_launches(torch, stream, lambda: swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), P, sm111, 1, scores.data_ptr(),
                                                                         stream.cuda_stream), 20, 5)
    want111, who = oracle_scores(np, a_h, b_h, sm111, 1)
    r111 = sw128_roofline(swmi, P, args.lanes, 0, "pairs", ms111, 1, -1, 1)
    rows["fixed_111_scorer"] = {"value": round(P / (ms111 * 1e-3), 1), "unit": "alignments/s", "kernel": r111.get("kernel"),
                                "kernel_ms": r111["kernel_ms"], "frac": r111.get("frac"), "steps": 20,
                                "gpu_mismatches": int((want111 != scores[:4096].cpu().numpy()).sum()), "checked_against": "%s, first 4096 pairs" % who,
                                "workload": "SURVEY 8f N2: sm +1/-1, gap 1 (SmithWaterman_8bit111simd, source.cpp:1105-1225): every folded score "
                                            ">= 0, so the packed kernel runs its unbiased cell (8 instructions per two rows)"}
    # the int32 kernel (round 1's cell: v_dot4 + v_max3_i32 + v_sub, schedule flag 8) on the same two parameter sets: what the
    # packed kernel is measured against
    def int32_row(name, match, mismatch, gap, ms_packed, workload):
        smx = swmi.match_matrix(match, mismatch)
        swmi.set_schedule(args.lanes, 8)
        try:
            ms = time_launches(torch, stream, lambda: swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), P, smx, gap, scores.data_ptr(),
                                                                             stream.cuda_stream), 20, 5)
        finally:
            swmi.set_schedule(args.lanes, 0)
        got = scores[:4096].cpu().numpy()
        want_x, who_x = oracle_scores(np, a_h, b_h, smx, gap)
        r = sw128_roofline(swmi, P, args.lanes, 8, "pairs", ms, match, mismatch, gap)
        rows[name] = {"value": round(P / (ms * 1e-3), 1), "unit": "alignments/s", "kernel": r.get("kernel"), "kernel_ms": r["kernel_ms"],
                      "frac": r.get("frac"), "steps": 20, "gpu_mismatches": int((want_x != got).sum()),
                      "checked_against": "%s, first 4096 pairs" % who_x, "packed_kernel_speedup_over_this": round(ms / ms_packed, 3),
                      "workload": workload}
    # the third packed cell body: parameters with some score + 2 gap < 0 ((5, -4, 0)) run round 2's biased form
    sm540 = swmi.match_matrix(5, -4)
    ms540 = time_launches(torch, stream, lambda: swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), P, sm540, 0, scores.data_ptr(),
                                                                         stream.cuda_stream), 20, 5)
    want540, who540 = oracle_scores(np, a_h, b_h, sm540, 0)
    r540 = sw128_roofline(swmi, P, args.lanes, 0, "pairs", ms540, 5, -4, 0)
    rows["biased_cell_5_-4_0"] = {"value": round(P / (ms540 * 1e-3), 1), "unit": "alignments/s", "kernel": r540.get("kernel"),
                                  "kernel_ms": r540["kernel_ms"], "frac": r540.get("frac"), "steps": 20,
                                  "gpu_mismatches": int((want540 != scores[:4096].cpu().numpy()).sum()), "checked_against": "%s, first 4096 pairs" % who540,
                                  "workload": "sm +5/-4, gap 0: a score + 2 gap is negative, so neither the unbiased nor the vertical-offset "
                                              "cell applies and the packed kernel runs its biased cell (11 instructions per two rows)"}
    int32_row("int32_cell_kernel", args.match, args.mismatch, args.gap, kernel_ms,
              "the headline batch on the int32 kernel (swmi_set_schedule flag 8: one alignment per 4 lanes, v_dot4 lookup)")
    int32_row("fixed_111_scorer_int32_cell_kernel", 1, -1, 1, ms111, "sm +1/-1, gap 1 on the int32 kernel")
    del d1, d2, scores
    torch.cuda.empty_cache()
    # (the clock takes ~10 launches to settle on a new kernel: with 3 warm-up launches this row read 6 % low)
    rows["banded_affine_1024"] = row_summary(bench_banded(args, swmi, np, torch, local_rank, steps=30, warmup=12))
    torch.cuda.empty_cache()
    rows["semiglobal_xdrop_65536"] = row_summary(bench_semiglobal(args, swmi, np, torch, local_rank, steps=5, warmup=2))
    line["rows"] = rows
    return line


def multi_gpu(args, swmi, np, torch, dist, rank, world, local_rank):
    """N > 1 (or --force-dist): contiguous shards, scores all-gathered through swmi/sharding.py."""
    from swmi import sharding
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    if args.lanes:
        swmi.set_schedule(args.lanes, 0)
    sm = swmi.match_matrix(args.match, args.mismatch)
    alloc = lambda n: torch.empty(n, dtype=torch.int32, device=dev)        # noqa: E731

    def run_leg(P, mode, steps, warmup, keep=False):
        """One timed leg: `steps` passes over this rank's P resident pairs; returns the per-leg record (and tensors if keep)."""
        n_total = P * world
        lo, hi = sharding.shard_bounds(n_total, rank, world)
        d1 = torch.empty(P * 128, dtype=torch.uint8, device=dev)
        d2 = torch.empty(P * 128, dtype=torch.uint8, device=dev)
        swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), P, args.seed, lo, stream.cuda_stream)
        pipe = sharding.GatherPipeline(n_total, rank, world, alloc, None, mode, depth=4 if P <= (1 << 22) else 2)

        def launch(out):
            swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), P, sm, args.gap, out.data_ptr(), stream.cuda_stream)
        for k in range(warmup):
            pipe.step(k, launch)
        pipe.finish()
        torch.cuda.synchronize()
        stride = 1 if P > (1 << 22) else 8
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if k % stride == 0 else None
                  for k in range(steps)]
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            ev = events[k]
            pipe.step(k, launch, (lambda e=ev: e[0].record(stream)) if ev else None, (lambda e=ev: e[1].record(stream)) if ev else None)
        full = pipe.finish()
        torch.cuda.synchronize()
        dist.barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        timed = [ev for ev in events if ev is not None]
        kernel_ms = sum(a.elapsed_time(b) for a, b in timed) / len(timed)
        # exposed cost of one gather with nothing to hide behind (kernel done, stream idle): host-timed around a synchronous one
        gather_ms = 0.0
        if mode != "none":
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            sharding.gather_scores(pipe.local[pipe.last_slot], n_total, None, out=pipe.full[0])
            torch.cuda.synchronize()
            gather_ms = (time.perf_counter() - tg) * 1e3
        per_rank = torch.tensor([kernel_ms, gather_ms], dtype=torch.float64, device=dev)
        every = torch.empty(2 * world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, per_rank)
        every = every.cpu().numpy().reshape(world, 2)
        rec = {"pairs_per_gpu": P, "global_pairs": n_total, "steps": steps, "gather": {"every": "after every step (async, overlapped)",
               "final": "one, after the last step", "none": "none (scores stay sharded)"}[mode],
               "value": round(n_total * steps / elapsed, 1), "unit": "alignments/s", "ms_per_step": round(elapsed * 1e3 / steps, 4),
               "kernel_ms_per_rank": [round(float(x), 4) for x in every[:, 0]],
               "isolated_gather_ms_per_rank": [round(float(x), 3) for x in every[:, 1]],
               "gather_bytes_per_rank": 4 * P if mode != "none" else 0}
        if keep:
            return rec, full, kernel_ms, n_total
        del d1, d2, pipe, full
        torch.cuda.empty_cache()
        return rec

    P = args.pairs or CONFIG3_PAIRS_PER_GPU
    small = min(P, 1 << 20)
    for _ in range(2):                          # settle clocks and the communicator on the small shape first
        run_leg(small, "every", 20, 10)
    legs = {}
    legs["configs1_size_gather_every_step"] = run_leg(small, "every", max(args.steps, 50), args.warmup)
    legs["no_gather"] = run_leg(P, "none", args.steps, min(args.warmup, 3))
    legs["one_final_gather"] = run_leg(P, "final", args.steps, min(args.warmup, 3))
    head, full, kernel_ms, n_total = run_leg(P, "every", args.steps, min(args.warmup, 3), keep=True)
    legs["gather_every_step"] = head

    # every rank must hold the same, complete score vector after the gather
    checksum = int(full.to(torch.int64).sum().item())
    c = torch.tensor([checksum], dtype=torch.int64, device=dev)
    cmin, cmax = c.clone(), c.clone()
    dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    ranks_agree = int(cmin.item()) == int(cmax.item())
    line = None
    if rank == 0:
        # a sample of EVERY shard of the gathered vector against the CPU checker: head, middle and tail of each shard
        per = 1024
        checked = mism = 0
        who = ""
        for r in range(world):
            lo, hi = sharding.shard_bounds(n_total, r, world)
            for first in sorted({lo, max(lo, (lo + hi) // 2 - per // 2), max(lo, hi - per)}):
                m = min(per, hi - first)
                a_h, b_h = swmi.generate_pairs_host(m, args.seed, first)
                want, who = oracle_scores(np, a_h, b_h, sm, args.gap)
                mism += int((want != full[first:first + m].cpu().numpy()).sum())
                checked += m
        lanes, flags = swmi.get_schedule()
        lanes = lanes or swmi.schedule_for_batch(P)
        roof = sw128_roofline(swmi, P, args.lanes, flags, "pairs", kernel_ms, args.match, args.mismatch, args.gap)
        roof["kernel_ms_note"] = "rank 0's kernel; every rank's is in legs.*.kernel_ms_per_rank"
        line = {
            "metric": "alignments/sec (and GCUPS) on 1M fixed-length pairs, 1/2/4/8 MI355X",
            "value": head["value"], "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(head["value"] / REFERENCE_PUBLISHED_ALIGN_PER_S, 1),
            "baseline_note": "reference README.md:4: simd4 ~4.4 s / 1M calls on one EPYC 7501 core (227k alignments/s)",
            "dtype": arith_dtype(roof.get("kernel")), "data": "synthetic", "gcups": round(head["value"] * CELLS / 1e9, 1),
            "config": {"workload": "BASELINE.json configs[3] shape: %d random 128x128 pairs per GPU per step (%d in all), sm %d/%d gap %d, "
                                   "inputs resident in HBM (generated on each GPU from the global pair index), int32 scores, RCCL "
                                   "all-gather of the scores after every step, overlapped with the next step's kernel" % (
                                       P, n_total, args.match, args.mismatch, args.gap),
                       "pairs_per_gpu": P, "global_pairs": n_total, "lanes_per_alignment": lanes, "schedule_flags": flags,
                       "parallelism": "batch-sharded x%d, one process per GPU, torch.distributed %s" % (world, args.backend),
                       "n1_note": "N = 1 runs configs[1] (1M pairs per step, no gather); legs.configs1_size_gather_every_step is that "
                                  "size on every GPU WITH the per-step gather"},
            # what ONE GPU does at the SAME per-GPU batch: the no-gather leg of this very run, per GPU.  N = 1 of the default
            # command times 1M pairs per step, where a step carries ~1.8 % of launch cost that a 64M-pair step amortises
            # (profiles/r03_launch_ramp.txt), so value(N) / (N x value(1)) would read ~2 % high; value / (N x this) does not.
            "n1_equivalent": {"value": round(legs["no_gather"]["value"] / world, 1), "unit": "alignments/s per GPU",
                              "pairs_per_gpu": P,
                              "how": "legs.no_gather.value / n_gpus (max over ranks of the wall time, same pairs per GPU, no gather); "
                                     "the same thing as its own run: `bench.py --gpus 1 --pairs %d` (at 67 108 864 pairs that measured 1 002 M/s in round 3)" % P,
                              "scaling_efficiency_from_it": round(head["value"] / legs["no_gather"]["value"], 4)},
            "legs": legs, "roofline": roof, "checksum": checksum, "ranks_agree_on_gathered_scores": ranks_agree,
            "gpu_scores_checked": checked, "gpu_mismatches": mism,
            "checked_against": "%s on head / middle / tail samples of every one of the %d shards of the gathered vector" % (who, world),
        }
    dist.barrier()
    return line


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        return respawn_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    collective = world > 1 or args.force_dist
    if collective:
        # Runtime environment FIRST: the HSA / HIP runtimes read their variables when they initialise, which the first
        # torch.cuda / swmi.init call below triggers (round 3 set these afterwards, where they did nothing; the driver's
        # environment already carries HSA_ENABLE_IPC_MODE_LEGACY=0, this keeps a hand-started run equivalent)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the gather's kernels run beside a scoring kernel that always has thousands of workgroups queued: give their
        # stream the dispatcher's preference so that they take the free wavefront slots as soon as they are ready
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if world == 1:                          # rehearsal of the collective path on a one-GPU box (not a result)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")

    import numpy as np
    import torch
    import torch.distributed as dist
    import swmi

    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback to measure)")
    if args.backend == "nccl" and world > n_dev:
        raise RuntimeError("%d ranks but %d GPUs: one process per GPU" % (world, n_dev))
    local_rank %= n_dev                         # only differs from LOCAL_RANK in a gloo rehearsal on a smaller box
    torch.cuda.set_device(local_rank)
    swmi.init(local_rank)                       # raises if there is no gfx950 device: the bench never falls back
    if collective:
        # RCCL prints a version banner on STDOUT when the communicator comes up; the contract is ONE JSON line there, so
        # stdout points at stderr until the communicator exists
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.backend)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    if args.mode == "banded-affine":            # BASELINE configs[4] (extension, parity unpinned by the reference)
        line = bench_banded(args, swmi, np, torch, local_rank)
    elif args.mode == "semiglobal":
        line = bench_semiglobal(args, swmi, np, torch, local_rank)
    elif collective:
        line = multi_gpu(args, swmi, np, torch, dist, rank, world, local_rank)
    else:
        line = single_gpu(args, swmi, np, torch, local_rank)
    if rank == 0 and line is not None:
        print(json.dumps(line), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
