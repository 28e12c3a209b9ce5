#!/usr/bin/env python3
"""bench.py -- alignments/s of the fixed-shape Smith-Waterman scorer on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--lanes L]

A "step" is one pass of the hot path over one batch of synthetic pairs: every rank scores its own P pairs
(P = 1,048,576 by default = BASELINE.json configs[1], "1M same-shape pairs on 1 MI355X") that are already
resident in HBM when the timed region starts, and -- for N > 1 -- the int32 scores are all-gathered over
RCCL (the path's only exchange step).  Weak scaling: per-GPU work is fixed as N grows.
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(one rank per GPU); started plainly with --gpus N > 1 the script spawns that launcher as a child process.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (sw128_kernel), measured live with HIP
events on the launch stream inside the timed region; `cpu_baseline` (N = 1 only) is the reference's own simd4
(oracle/_ref/libswref.so, compiled from the reference sources in the build container) timed on ONE host core
of this box on a bounded sample of the same generated pairs, which doubles as a bit-exactness check of the GPU
scores.  The oracle / reference build are used here only as checker and baseline, never as the measured path.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "smith-waterman-simd_amd")
sys.path.insert(0, PKG)

CELLS = 128 * 128
INT_OPS_PER_ALIGNMENT = CELLS * 7          # 1 add, 2 sub, 4 max per cell (source.cpp:49-53)
BYTES_PER_ALIGNMENT = 128 + 128 + 4
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32 x 2.4 GHz = 78.6 T full-rate 32-bit integer lane-ops/s
# (v_add_u32 / v_sub_u32 issue at 32 lanes/clk/SIMD; v_max_i32, v_max3_i32 and v_dot4 at half of that --
#  tools/microbench/valu_rate*.hip).
VALU_PEAK_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12
HBM_PEAK_GBS = 8000.0
REFERENCE_PUBLISHED_ALIGN_PER_S = 1e6 / 4.4    # README.md:4 of the reference: simd4 ~4.4 s / 1M on one EPYC 7501 core


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30, help="untimed steps (the clock takes ~10 launches to settle after idle)")
    ap.add_argument("--len", type=int, default=1024, help="sequence length of the banded-affine mode")
    ap.add_argument("--gap-open", type=int, default=5)
    ap.add_argument("--gap-extend", type=int, default=1)
    ap.add_argument("--mode", default="pairs", choices=["pairs", "packed", "one-vs-many", "banded-affine", "semiglobal"],
                    help="pairs = the headline path; packed = 2-bit inputs (SURVEY 8f N3); one-vs-many = every seq1 against "
                         "ONE seq2 (N1) -- secondary rows, same kernel, same contract")
    ap.add_argument("--pairs", type=int, default=1 << 20, help="pairs per GPU per step")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per alignment (0 = library default)")
    ap.add_argument("--match", type=int, default=10)
    ap.add_argument("--mismatch", type=int, default=-30)
    ap.add_argument("--gap", type=int, default=15)
    ap.add_argument("--seed", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        threads = os.cpu_count() or 1
        bounds = [(sample * t // threads, sample * (t + 1) // threads) for t in range(threads)]
        out_mt = np.zeros(sample, np.int32)

        passes = 16                                        # each thread scores its shard 16 times: ~0.5 s of work per thread

        def shard(b):
            lo, hi = b
            for _ in range(passes if hi > lo else 0):
                ref.swref_batch(4, seq1[lo:].ctypes.data_as(vp), seq2[lo:].ctypes.data_as(vp), ctypes.c_size_t(hi - lo),
                                sm.ctypes.data_as(vp), args.gap, out_mt[lo:].ctypes.data_as(vp))
        with ThreadPoolExecutor(threads) as ex:
            t2 = time.perf_counter()
            list(ex.map(shard, bounds))
            dt_mt = time.perf_counter() - t2
        info["all_host_cores"] = {"threads": threads, "value": round(sample * passes / dt_mt, 1), "unit": "alignments/s",
                                  "agrees_with_one_core": bool((out_mt == out).all()),
                                  "note": "threads = os.cpu_count(); a container CPU quota (16 cores per GPU on the test "
                                          "boxes) caps what they deliver"}
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
                 "seconds": round(dt, 2), "host_cpu": model, "host_cores_available": os.cpu_count(),
                 "gpu_scores_checked": sample, "gpu_mismatches": mism})
    return info


def bench_banded(args, swmi, np, torch, local_rank):
    """Secondary row: 1024 x 1024 affine gap, 128-diagonal band, one wavefront per alignment (single GPU)."""
    length = args.len
    P = args.pairs if args.pairs != (1 << 20) else 1 << 16
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
    for _ in range(args.warmup):
        launch()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream); launch(); b.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    band_cells = sum(min(length, i + 63) - max(1, i - 64) + 1 for i in range(1, length + 1))
    value = P * args.steps / elapsed
    line = {"metric": "alignments/sec (and GCUPS), banded affine extension", "value": round(value, 1), "unit": "alignments/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "gcups": round(value * band_cells / 1e9, 1),
            "config": {"workload": "BASELINE.json configs[4] (extension, parity unpinned by the reference): %d pairs of %d-mers, "
                                   "128-diagonal band, sm 2/-3, gap open %d extend %d, inputs resident in HBM" % (
                                       P, length, args.gap_open, args.gap_extend), "band_cells_per_alignment": band_cells},
            "roofline": {"bound": "valu", "kernel": "sw_banded_affine_kernel", "kernel_ms": round(kernel_ms, 4),
                         # 12 algorithmic int ops per cell: 4 sub, 2+3 max, 1 add, 1 running max, 1 lookup
                         "achieved": round(P * band_cells * 12 / (kernel_ms * 1e-3) / 1e12, 3), "peak": round(VALU_PEAK_TOPS, 1),
                         "unit": "TOP/s (int32)", "frac": round(P * band_cells * 12 / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS, 4),
                         "traffic": None},
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
        line["cpu_baseline"] = {"kind": "port", "function": "oracle/sw_oracle.c sw_oracle_banded_affine (scalar, full tables)",
                                "value": round(sample / dt, 1), "unit": "alignments/s", "cores": 1,
                                "sample": "first %d pairs of the batch" % sample,
                                "gpu_mismatches": int((want != scores[:sample].cpu().numpy()).sum())}
    print(json.dumps(line), flush=True)
    return 0


def sg_sweep_kernel(P):
    """Name of the sweep kernel launch_semiglobal picks for P alignments (same cost model as sg_kernels.hip)."""
    if P < 6144:
        return "sg_forward_kernel<8>"
    w4, w2 = (P // 16 + 1023) // 1024, (P // 32 + 1023) // 1024
    t4 = 16.2 if w4 <= 1 else 23.5 if w4 == 2 else 31.7 if w4 == 3 else 5.7 + 8.57 * w4
    t2 = 38.0 if w2 <= 2 else 6.6 + 15.45 * w2
    return "sg_forward_split_kernel<4, %d>" % min(max(w4, 1), 4) if t4 <= t2 else "sg_forward_split_kernel<2, %d>" % (2 if w2 <= 2 else 3)


def sg_traffic(P, kernel):
    """HBM bytes per launch of the sweep kernel from the committed PMC passes (65536 alignments), or None."""
    f = os.path.join(ROOT, "profiles", "r01_semiglobal_pmc.json")
    if P != 65536 or not os.path.exists(f):
        return None
    try:
        k = json.load(open(f)).get(kernel, {})
        return int(k["hbm_read_bytes_x2"] + k["hbm_write_bytes"])
    except Exception:
        return None


def bench_semiglobal(args, swmi, np, torch, local_rank):
    """Secondary row: the reference's semi-global adaptive-band X-drop aligner incl. traceback (single GPU).
    Inputs follow SpeedtestSemiGlobal (source.cpp:2805-2813): a random 16384-mer and a copy with 5 % substitutions."""
    L = 16384
    P = args.pairs if args.pairs != (1 << 20) else 65536
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    g = torch.Generator(device=dev); g.manual_seed(args.seed)
    d1 = torch.randint(0, 4, (P, L), dtype=torch.uint8, device=dev, generator=g)
    rnd = torch.randint(0, 4, (P, L), dtype=torch.uint8, device=dev, generator=g)
    keep = torch.rand((P, L), device=dev, generator=g) < 0.95
    d2 = torch.where(keep, d1, rnd).contiguous()
    cap = 32769
    scores = torch.empty(P, dtype=torch.int32, device=dev)
    lengths = torch.empty(P, dtype=torch.int32, device=dev)
    tb = torch.empty((P, cap, 2), dtype=torch.int32, device=dev)

    def launch():
        swmi.semiglobal_xdrop_device(d1.data_ptr(), d2.data_ptr(), P, scores.data_ptr(), tb.data_ptr(), cap, lengths.data_ptr(),
                                     stream.cuda_stream)
    steps, warm = min(args.steps, 10), min(args.warmup, 2)
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
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
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
    sweep_kernel = sg_sweep_kernel(P)
    line["roofline"] = {
        "bound": "valu", "kernel": sweep_kernel,
        "kernel_ms": round(sweep_ms, 3), "achieved": round(sweep_ops / (sweep_ms * 1e-3) / 1e12, 3),
        "peak": round(VALU_PEAK_TOPS, 1), "unit": "TOP/s (int32)",
        "frac": round(sweep_ops / (sweep_ms * 1e-3) / 1e12 / VALU_PEAK_TOPS, 4),
        "algorithmic_ops_per_launch": sweep_ops, "gcups_kernel": round(P * rounds * 32 / (sweep_ms * 1e-3) / 1e9, 1),
        "traffic": sg_traffic(P, sweep_kernel),
        "traffic_source": "profiles/r01_semiglobal_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of the sweep kernel, separate passes; "
                          "the records leave in 64-byte pieces of 128-byte L2 lines, hence the read-for-ownership traffic)",
        "kernel_ms_covers": "sweep phase between HIP events: the stream-packing pre-pass (~0.6 ms at 65536) + the sweep kernel",
        "traceback_kernel_ms": round(tb_ms, 3),
        "hbm": {"algorithmic_bytes_per_launch": alg_bytes, "achieved": round(alg_bytes / ((sweep_ms + tb_ms) * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / ((sweep_ms + tb_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "note": "sequences in + (i, j) pairs out; the predecessor records between the two kernels (10 B per round, "
                        "written once and read once) are implementation traffic on top"}}
    if not args.no_cpu_baseline:
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
                                    "sample": "first %d pairs of the batch" % sample, "gpu_mismatches": mism}
    print(json.dumps(line), flush=True)
    return 0


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        return respawn_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist
    import swmi
    from swmi import sharding

    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback to measure)")
    if args.backend == "nccl" and world > n_dev:
        raise RuntimeError("%d ranks but %d GPUs: one process per GPU" % (world, n_dev))
    local_rank %= n_dev                         # only differs from LOCAL_RANK in a gloo rehearsal on a smaller box
    torch.cuda.set_device(local_rank)
    swmi.init(local_rank)                       # raises if there is no gfx950 device: the bench never falls back
    if args.lanes:
        swmi.set_schedule(args.lanes, 0)
    lanes, flags = swmi.get_schedule()
    lanes = lanes or swmi.schedule_for_batch(args.pairs)      # 0 = automatic: what it resolves to for this batch size
    if world > 1 or args.force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the gather's kernels run beside a scoring kernel that always has 16 384 workgroups queued: give their stream the
        # dispatcher's preference so that they take the free wavefront slots as soon as they are ready
        os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
        if world == 1:                          # rehearsal of the collective path on a one-GPU box (not a result)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
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
        return bench_banded(args, swmi, np, torch, local_rank)
    if args.mode == "semiglobal":
        return bench_semiglobal(args, swmi, np, torch, local_rank)
    P = args.pairs
    n_total = P * world
    collective = world > 1 or args.force_dist   # the score gather (and the barriers) run whenever a process group exists
    lo, hi = sharding.shard_bounds(n_total, rank, world)        # contiguous shard of the global pair index space
    dev = torch.device("cuda", local_rank)
    d1 = torch.empty(P * 128, dtype=torch.uint8, device=dev)
    d2 = torch.empty(P * 128, dtype=torch.uint8, device=dev)
    p1 = p2 = None
    # a ring of score buffers: the RCCL gather of step k (async, on the process group's stream) overlaps the kernels of the
    # following steps; four deep, so that a gather that gets its compute units late does not hold up the next launch
    nbuf = 4 if collective else 1
    scores = [torch.empty(P, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    gathered = [torch.empty(n_total, dtype=torch.int32, device=dev) for _ in range(nbuf)] if collective else scores
    pending = [None] * nbuf
    stream = torch.cuda.current_stream()
    swmi.generate_pairs_device(d1.data_ptr(), d2.data_ptr(), P, args.seed, lo, stream.cuda_stream)
    sm = swmi.match_matrix(args.match, args.mismatch)
    if args.mode == "packed":                   # pack on the device with torch (input preparation, outside the timed region)
        def pack(d):
            v = d.view(P, 32, 4).to(torch.int32)
            return (v[..., 0] | (v[..., 1] << 2) | (v[..., 2] << 4) | (v[..., 3] << 6)).to(torch.uint8).contiguous()
        p1, p2 = pack(d1), pack(d2)

    def launch(out_ptr):
        if args.mode == "packed":
            swmi.score_batch_device(p1.data_ptr(), p2.data_ptr(), P, sm, args.gap, out_ptr, stream.cuda_stream, packed=True)
        elif args.mode == "one-vs-many":
            swmi.score_one_vs_many_device(d1.data_ptr(), P, d2.data_ptr(), sm, args.gap, out_ptr, stream.cuda_stream)
        else:
            swmi.score_batch_device(d1.data_ptr(), d2.data_ptr(), P, sm, args.gap, out_ptr, stream.cuda_stream)

    def step(k, ev=None):
        buf = k % len(scores)
        if pending[buf] is not None:
            pending[buf].wait()                 # stream-side wait: the gather that read scores[buf] nbuf steps ago is done
        if ev is not None:
            ev[0].record(stream)
        launch(scores[buf].data_ptr())
        if ev is not None:
            ev[1].record(stream)
        if collective:
            pending[buf] = dist.all_gather_into_tensor(gathered[buf], scores[buf], async_op=True)

    def drain():
        for w in pending:
            if w is not None:
                w.wait()
        torch.cuda.synchronize()

    # settle the clocks first: after idle the first ~10 launches run 10 % slower (DVFS ramp, profiles/r01c kernel trace);
    # this is extra untimed work on top of the W warmup steps the caller asked for, so that a small W still measures
    # the steady state
    for k in range(40):
        step(k)
    drain()
    for k in range(args.warmup):
        step(k)
    drain()
    # HIP events bracket every `stride`-th launch of the timed region: an event pair per launch costs ~20 us of stream time
    # (1.3 % of a 1.5 ms step), which would show up in `value`
    stride = max(1, int(os.environ.get("SWMI_BENCH_EVENT_STRIDE", "8")))
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if k % stride == 0 else None
              for k in range(args.steps)]
    if collective:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, events[k])
    drain()
    if collective:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if collective:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed = [ev for ev in events if ev is not None]
    kernel_ms = sum(a.elapsed_time(b) for a, b in timed) / len(timed)      # HIP events on the launch stream

    # every rank must hold the same, complete score vector after the gather
    last = (args.steps - 1) % len(scores)
    checksum = int(gathered[last].to(torch.int64).sum().item())
    if collective:
        c = torch.tensor([checksum], dtype=torch.int64, device=dev)
        cmin, cmax = c.clone(), c.clone()
        dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        assert int(cmin.item()) == int(cmax.item()), "ranks disagree on the gathered scores"

    if rank == 0:
        value = n_total * args.steps / elapsed
        kernel_s = kernel_ms * 1e-3
        bytes_per_alignment = {"pairs": BYTES_PER_ALIGNMENT, "packed": 32 + 32 + 4, "one-vs-many": 128 + 4}[args.mode]
        roof = {
            "bound": "valu",
            "kernel": "sw128_kernel<L=%d>" % lanes,
            "achieved": round(P * INT_OPS_PER_ALIGNMENT / kernel_s / 1e12, 3),
            "peak": round(VALU_PEAK_TOPS, 1),
            "unit": "TOP/s (int32)",
            "frac": round(P * INT_OPS_PER_ALIGNMENT / kernel_s / 1e12 / VALU_PEAK_TOPS, 4),
            "traffic": None,
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_ops_per_launch": P * INT_OPS_PER_ALIGNMENT,
            "algorithmic_bytes_per_launch": P * bytes_per_alignment,
            "hbm": {"achieved": round(P * bytes_per_alignment / kernel_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(P * bytes_per_alignment / kernel_s / 1e9 / HBM_PEAK_GBS, 5)},
            "gcups_kernel": round(P * CELLS / kernel_s / 1e9, 1),
        }
        # the honest utilisation figure: VALU issue cycles the cell body needs (1 v_dot4 + 1 v_max3 + 1/4 v_max3 at 4 clk,
        # 1 v_sub at 2 clk = 11 per wave-cell) against the cycles the launch took at the nominal clock
        wave_cells = P * CELLS / 64.0
        ideal_s = wave_cells * 11.0 / (256 * 4 * 2.4e9)
        roof["issue_bound"] = {"valu_cycles_per_wave_cell": 11, "ideal_kernel_ms_at_2.4GHz": round(ideal_s * 1e3, 4),
                               "frac": round(ideal_s / kernel_s, 4)}
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file))
                if tr.get("pairs_per_launch") == P and args.mode == "pairs":
                    roof["traffic"] = tr.get("hbm_bytes_per_launch")
                    roof["traffic_source"] = tr.get("source")
            except Exception:
                pass
        line = {
            "metric": "alignments/sec (and GCUPS) on 1M fixed-length pairs, 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(value / REFERENCE_PUBLISHED_ALIGN_PER_S, 1),
            "baseline_note": "reference README.md:4: simd4 ~4.4 s / 1M calls on one EPYC 7501 core (227k alignments/s)",
            "dtype": "int32", "data": "synthetic",
            "gcups": round(value * CELLS / 1e9, 1),
            "config": {"workload": "%s: %d random 128x128 pairs per GPU per step, sm %d/%d gap %d, "
                                   "inputs resident in HBM, int32 scores%s" % (
                                       {"pairs": "BASELINE.json configs[1]", "packed": "SURVEY 8f N3 (2-bit packed inputs, source.cpp:1581)",
                                        "one-vs-many": "SURVEY 8f N1 (every seq1 vs ONE seq2, source.cpp:1227)"}[args.mode],
                                       P, args.match, args.mismatch, args.gap,
                                       ", RCCL all-gather of scores each step (overlapped with the next step's kernel)" if collective else ""),
                       "pairs_per_gpu": P, "global_pairs": n_total, "lanes_per_alignment": lanes, "schedule_flags": flags,
                       "parallelism": "batch-sharded x%d" % world},
            "roofline": roof, "checksum": checksum,
        }
        if world == 1 and not args.no_cpu_baseline and args.mode == "pairs":
            sample = args.cpu_sample or min(P, 1 << 20)
            line["cpu_baseline"] = cpu_baseline(swmi, np, args, scores[0].cpu().numpy(), min(sample, P))
            line["gpu_over_cpu_core"] = round(value / line["cpu_baseline"]["value"], 1)
            # end-to-end through the host-buffer entry point (H2D + kernel + D2H; SURVEY 8d "reported separately"; never `value`)
            h1, h2 = swmi.generate_pairs_host(P, args.seed, 0)
            swmi.score_batch(h1, h2, sm, args.gap)
            t3 = time.perf_counter()
            hs = swmi.score_batch(h1, h2, sm, args.gap)
            dt3 = time.perf_counter() - t3
            line["host_buffer_path"] = {"entry": "swmi_score_batch (pageable host memory, PCIe inclusive)",
                                        "ms": round(dt3 * 1e3, 3), "value": round(P / dt3, 1), "unit": "alignments/s",
                                        "matches_resident_scores": bool((hs == scores[0].cpu().numpy()).all())}
        print(json.dumps(line), flush=True)
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
