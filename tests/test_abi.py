"""The C ABI without a GPU: the library loads, exports every symbol include/swmi.h declares, and refuses to
score when there is no device (there is no CPU fallback in the product path)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import PKG, ROOT, match_matrix


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "swmi.h")).read()
    return sorted(set(re.findall(r"SWMI_API\s+[^;(]*?\b(swmi_\w+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = _declared_symbols()
    for must in ("swmi_init", "swmi_shutdown", "swmi_last_error", "swmi_score_pair", "swmi_score_batch",
                 "swmi_score_batch_device", "swmi_score_one_vs_many", "swmi_score_one_vs_many_device", "swmi_score_batch_packed", "swmi_unpack",
                 "swmi_queue_create", "swmi_queue_submit", "swmi_queue_wait", "swmi_queue_destroy",
                 "swmi_set_schedule", "swmi_generate_pairs_device", "swmi_generate_pairs_host",
                 "swmi_time_batch_device", "swmi_get_device_info", "swmi_score_banded_affine",
                 "swmi_score_banded_affine_device", "swmi_semiglobal_xdrop", "swmi_semiglobal_xdrop_device",
                 "swmi_init_all", "swmi_init_devices", "swmi_use_gpu", "swmi_num_gpus", "swmi_shard_bounds",
                 "swmi_score_batch_multi", "swmi_score_batch_packed_multi", "swmi_sharded_create", "swmi_sharded_score",
                 "swmi_sharded_wait", "swmi_sharded_scores_host", "swmi_sharded_gathered_device", "swmi_sharded_time",
                 "swmi_sharded_destroy", "swmi_semiglobal_kernels_for_batch", "swmi_semiglobal_release_workspaces",
                 "swmi_semiglobal_set_mapping", "swmi_semiglobal_set_exact", "swmi_semiglobal_window_stats"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(PKG, "lib", "libswmi.so"))
    for name in _declared_symbols():
        assert hasattr(lib, name), "libswmi.so does not export %s" % name
    assert lib.swmi_version() == 300


def test_product_library_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", os.path.join(PKG, "lib", "libswmi.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out and "swref" not in out
    syms = subprocess.run(["nm", "-D", os.path.join(PKG, "lib", "libswmi.so")], stdout=subprocess.PIPE, text=True).stdout
    assert "sw_oracle" not in syms and "swref_" not in syms


def test_argument_errors_do_not_need_a_device(swmi_mod):
    lib = swmi_mod.load()
    sm = match_matrix(10, -30)
    a = np.zeros(128, np.uint8)
    out = np.zeros(1, np.int32)
    vp = ctypes.c_void_p
    # NULL matrix
    assert lib.swmi_score_batch(a.ctypes.data, a.ctypes.data, 1, None, 15, out.ctypes.data) == swmi_mod.ERR_INVALID_ARGUMENT
    # negative gap is outside the domain
    assert lib.swmi_score_batch(a.ctypes.data, a.ctypes.data, 1, sm.ctypes.data, -1, out.ctypes.data) == swmi_mod.ERR_DOMAIN
    assert b"gap_penalty" in lib.swmi_last_error()
    # n = 0 is a no-op that needs neither buffers nor a device (the reference loop simply would not run)
    assert lib.swmi_score_batch(None, None, 0, sm.ctypes.data, 15, None) == swmi_mod.OK
    # NULL buffers with n > 0
    assert lib.swmi_score_batch(None, a.ctypes.data, 1, sm.ctypes.data, 15, out.ctypes.data) == swmi_mod.ERR_INVALID_ARGUMENT
    assert lib.swmi_set_schedule(3, 0) == swmi_mod.ERR_INVALID_ARGUMENT
    assert lib.swmi_set_schedule(8, 0) == swmi_mod.OK
    # the semi-global switches: a mapping that does not exist (13 and 24 went with their spilling builds), an exact flag that is no flag
    for bad in (0, 3, 13, 24, 45):
        assert lib.swmi_semiglobal_set_mapping(bad) == swmi_mod.ERR_INVALID_ARGUMENT
    for ok in (12, 23, 44, 1, 2, 4, -1):
        assert lib.swmi_semiglobal_set_mapping(ok) == swmi_mod.OK
    assert lib.swmi_semiglobal_set_exact(2) == swmi_mod.ERR_INVALID_ARGUMENT and lib.swmi_semiglobal_set_exact(-1) == swmi_mod.ERR_INVALID_ARGUMENT
    assert lib.swmi_semiglobal_set_exact(1) == swmi_mod.OK and lib.swmi_semiglobal_set_exact(0) == swmi_mod.OK
    counts = (ctypes.c_uint64 * 4)(7, 7, 7, 7)
    assert lib.swmi_semiglobal_window_stats(None, None) == swmi_mod.ERR_INVALID_ARGUMENT
    assert lib.swmi_semiglobal_window_stats(None, counts) != swmi_mod.OK and list(counts) == [0, 0, 0, 0]  # no device: refused, counters zeroed


def test_c_shard_rule_is_the_python_shard_rule(swmi_mod):
    """swmi_shard_bounds (what swmi_score_batch_multi / swmi_sharded_* split by, swmi_multi.cpp) and
    sharding.shard_bounds (what bench.py's ranks split by) are one rule; needs no device."""
    from swmi import sharding
    for n in (0, 1, 7, 8, 9, 1000003, 1 << 20, (1 << 29) + 3):
        for world in (1, 2, 3, 4, 7, 8):
            c = [swmi_mod.shard_bounds(n, r, world) for r in range(world)]
            assert c == [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert c[0][0] == 0 and c[-1][1] == n
            assert all(h0 == l1 for (_, h0), (l1, _) in zip(c, c[1:]))
    with pytest.raises(swmi_mod.SwmiError):
        swmi_mod.shard_bounds(10, 2, 2)


def test_kernel_choice_is_reported_without_a_device(swmi_mod):
    """swmi_score_kernel_for_batch: which kernel instantiation a launch runs (what bench.py prices with tools/isa_census.py).
    L = 4, 8 and 16 run the packed kernel (two alignments per register; 32 / 16 / 8 alignments per wavefront) -- cell body 0
    when every score + gap >= 0, 2 (vertical offsets) when every score + 2 gap >= 0 and the offsets fit, 1 (biased) otherwise;
    flag 8 and the other lane counts run the int32 kernel; small batches take more lanes."""
    big = 1 << 20
    swmi_mod.set_schedule(0, 0)
    try:
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(10, -30), 15) == ("sw128_pk_kernel<0,2>", 32)      # -30 + 2 * 15 = 0
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(1, -1), 1) == ("sw128_pk_kernel<0,0>", 32)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(127, -127), 127, mode=1) == ("sw128_pk_kernel<1,0>", 32)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(127, -128), 0, mode=2) == ("sw128_pk_kernel<2,1>", 32)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(5, -4), 0) == ("sw128_pk_kernel<0,1>", 32)          # -4 + 0 < 0
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(10, -30), 14) == ("sw128_pk_kernel<0,1>", 32)       # -30 + 28 < 0
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(127, -127), 64) == ("sw128_pk_kernel<0,2>", 32)     # 127 + 128 = 255 fits a byte
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(127, -127), 65) == ("sw128_pk_kernel<0,1>", 32)     # 127 + 130 does not
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(120, -120), 62) == ("sw128_pk_kernel<0,2>", 32)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(126, -120), 64, mode=2) == ("sw128_pk_kernel<2,2>", 32)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(100, -100), 100) == ("sw128_pk_kernel<0,0>", 32)    # every s + gap >= 0
        assert swmi_mod.score_kernel_for_batch(1000, match_matrix(10, -30), 15) == ("sw128_kernel<64,1,0,0>", 1)
        assert swmi_mod.score_kernel_for_batch(10000, match_matrix(10, -30), 15) == ("sw128_pk_kernel<0,2,16>", 8)
        assert swmi_mod.score_kernel_for_batch(50000, match_matrix(10, -30), 15) == ("sw128_pk_kernel<0,2,8>", 16)
        # the automatic choice switches where the measured kernel times cross (profiles/r02_small_batch_schedule.txt)
        per_wave = [swmi_mod.score_kernel_for_batch(n, match_matrix(10, -30), 15)[1] for n in (2048, 2049, 5120, 5121, 24576, 24577, 98304, 98305)]
        assert per_wave == [1, 2, 2, 8, 8, 16, 16, 32]
        swmi_mod.set_schedule(4, swmi_mod.NO_PACKED)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(10, -30), 15) == ("sw128_kernel<4,1,0,0>", 16)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(127, -127), 127) == ("sw128_kernel<4,0,0,0>", 16)   # 127 + 127 does not fold
        swmi_mod.set_schedule(8, 0)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(1, -1), 1) == ("sw128_pk_kernel<0,0,8>", 16)
        swmi_mod.set_schedule(8, swmi_mod.NO_PACKED)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(1, -1), 1) == ("sw128_kernel<8,1,0,0>", 8)
        swmi_mod.set_schedule(0, swmi_mod.NO_PACKED)
        assert swmi_mod.score_kernel_for_batch(10000, match_matrix(10, -30), 15) == ("sw128_kernel<16,1,0,0>", 4)
        swmi_mod.set_schedule(4, swmi_mod.USE_LUT)
        assert swmi_mod.score_kernel_for_batch(big, match_matrix(1, -1), 1) == ("sw128_lut_kernel<4,0>", 16)
    finally:
        swmi_mod.set_schedule(0, 0)


def test_python_binding_rejects_values_ctypes_would_wrap(swmi_mod):
    a = np.zeros((1, 128), np.uint8)
    for bad_gap in (256, 300, -129, 128 + 256):
        with pytest.raises(swmi_mod.SwmiError) as e:
            swmi_mod.score_batch(a, a, match_matrix(1, -1), bad_gap)
        assert e.value.code == swmi_mod.ERR_DOMAIN
    with pytest.raises(swmi_mod.SwmiError) as e:
        swmi_mod.score_batch(a, a, np.full(16, 200, np.int32), 1)       # would wrap to -56 in an int8 cast
    assert e.value.code == swmi_mod.ERR_DOMAIN


def test_multi_gpu_entry_points_without_a_device_fail_loudly(swmi_mod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device behaviour is covered on the CPU runner")
    lib = swmi_mod.load()
    assert lib.swmi_init_all(0) == swmi_mod.ERR_NO_DEVICE
    assert swmi_mod.num_gpus() == 0
    a = np.zeros((70, 128), np.uint8)
    with pytest.raises(swmi_mod.SwmiError) as e:
        swmi_mod.score_batch_multi(a, a, match_matrix(1, -1), 1)
    assert e.value.code == swmi_mod.ERR_NOT_INITIALIZED
    with pytest.raises(swmi_mod.SwmiError) as e:
        swmi_mod.ShardedBatch(1000)
    assert e.value.code == swmi_mod.ERR_NOT_INITIALIZED


def test_scoring_without_a_device_fails_loudly(swmi_mod):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device behaviour is covered on the CPU runner")
    lib = swmi_mod.load()
    rc = lib.swmi_init(0)
    assert rc == swmi_mod.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.swmi_last_error()
    with pytest.raises(swmi_mod.SwmiError) as e:
        swmi_mod.score_pair(np.zeros(128, np.uint8), np.zeros(128, np.uint8), match_matrix(1, -1), 1)
    assert e.value.code == swmi_mod.ERR_NOT_INITIALIZED


def test_host_generator_matches_the_oracle_generator(swmi_mod, oracle):
    # two independent implementations of the specification in include/swmi.h
    for seed, first in ((10000, 0), (1, 12345678901), (2**63 + 5, 2**40)):
        a, b = swmi_mod.generate_pairs_host(257, seed, first)
        oa, ob = oracle.generate(257, seed, first)
        assert np.array_equal(a, oa) and np.array_equal(b, ob)
    a, _ = swmi_mod.generate_pairs_host(2, 10000, 0)
    assert list(a[0][:16]) == [0, 3, 3, 1, 2, 0, 2, 3, 0, 1, 1, 2, 3, 0, 1, 2]   # known answer, pins the spec
    a2, _ = swmi_mod.generate_pairs_host(1, 10000, 1)
    assert np.array_equal(a2[0], a[1])                                           # counter-based: position independent


def test_pack_helper_round_trips(swmi_mod, golden):
    f = golden("f5_siblings")
    assert np.array_equal(swmi_mod.pack(f["unpacked"]), f["packed"])


def test_host_batch_granules(swmi_mod):
    """The pipeline schedule of swmi_score_batch (swmi_api.cpp next_granule): tapering granules, every pair exactly once,
    few copy commands, a small last granule so that almost no kernel time is left behind the last copy.  Needs no device."""
    for k in ("SWMI_HOST_GRANULE", "SWMI_HOST_SERIAL", "SWMI_HOST_TAPER", "SWMI_HOST_MIN_GRANULE", "SWMI_TEST_SCORE_GROUP"):
        assert k not in os.environ
    assert swmi_mod.host_granules(0) == []
    assert swmi_mod.host_granules(1) == [1]
    assert swmi_mod.host_granules(16384) == [16384] and swmi_mod.host_granules(5000) == [5000]
    assert swmi_mod.host_granules(1 << 20) == [786432, 196608, 49152, 16384]
    assert swmi_mod.host_granules(1 << 22) == [1 << 20, 1 << 20, 1 << 20, 786432, 196608, 49152, 16384]
    for n in (16385, 65537, 100000, (1 << 20) + 1, 3000001, (1 << 24) + 12345, (1 << 26) + 7):
        g = swmi_mod.host_granules(n)
        assert sum(g) == n and max(g) <= 1 << 20 and min(g[:-1] or [1 << 14]) >= 1 << 14
        assert g[-1] <= 1 << 14 or n <= 1 << 14 or len(g) == 1
        assert len(g) <= n // (1 << 20) + 16             # a handful of copy commands beyond the 1M-pair granules
    # The schedule follows the entry's bytes per pair (swmi_api.cpp next_granule): the taper ratio is kernel time over copy
    # time per pair, so that granule k's kernel is done when granule k + 1 has landed
    assert swmi_mod.host_granules(1 << 20, swmi_mod.ENTRY_PAIRS) == swmi_mod.host_granules(1 << 20)
    ovm = swmi_mod.host_granules(1 << 20, swmi_mod.ENTRY_ONE_VS_MANY)          # 128 B per pair: halves
    assert ovm == [524288, 262144, 131072, 65536, 32768, 16384, 16384]
    for entry, ratio_lo, ratio_hi in ((swmi_mod.ENTRY_PAIRS, 0.2, 0.3), (swmi_mod.ENTRY_ONE_VS_MANY, 0.45, 0.55)):
        for n in (1 << 20, (1 << 22) + 4097, (1 << 24) + 12345):
            g = swmi_mod.host_granules(n, entry)
            assert sum(g) == n and max(g) <= 1 << 20
            # inside one score group, away from the caps: consecutive granules shrink by the entry's ratio
            body = [x for x in g[: g.index(min(g))] if x < 1 << 20]
            ratios = [b / a for a, b in zip(body, body[1:]) if b < a and b > 4 * 16384]
            assert all(ratio_lo <= r <= ratio_hi for r in ratios), (entry, n, ratios)
    # 64 B per pair: copy and kernel are level -- equal granules of 128 K between a short first and last one, two issuing threads
    packed = swmi_mod.host_granules(1 << 20, swmi_mod.ENTRY_PACKED)
    assert packed == [32768, 98304] + [131072] * 6 + [98304, 32768]
    for n in (524288, (1 << 22) + 5000, (1 << 24) + 5):
        g = swmi_mod.host_granules(n, swmi_mod.ENTRY_PACKED)
        group = g[: g.index(32768, 1) + 1]                                       # the first score group's granules
        assert sum(g) == n and group[:2] == [32768, 98304] and group[-2:] == [98304, 32768]
        assert all(126976 <= x <= 135168 for x in group[2:-2])
    assert swmi_mod.host_granules(300000, swmi_mod.ENTRY_PACKED) == [65536] * 4 + [37856]     # too short for the shape
    assert swmi_mod.host_granules(100, 7) == []                                  # unknown entry


def test_rccl_probe_reports_a_missing_library(swmi_mod):
    """load_rccl()'s failure path (swmi_multi.cpp): a library that does not exist must come back as "not usable" with the
    loader's reason -- round 2 called dlerror() twice there and built a std::string from NULL.  Run in a child process: the
    library decides once per process which librccl it uses."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import swmi; ok, why = swmi.rccl_probe(); print(int(ok)); print(why)" % PKG)
    missing = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             env=dict(os.environ, SWMI_RCCL_LIB="/nonexistent/librccl-not-here.so"))
    assert missing.returncode == 0, missing.stderr
    lines = missing.stdout.strip().splitlines()
    assert lines[0] == "0" and "librccl-not-here" in lines[1]
    present = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                             env={k: v for k, v in os.environ.items() if k != "SWMI_RCCL_LIB"})
    assert present.returncode == 0, present.stderr
    lines = present.stdout.strip().splitlines()
    assert lines[0] in ("0", "1") and (lines[0] == "1" or len(lines) > 1)       # this image ships librccl: normally 1


def test_host_schedule_knobs_are_read_from_the_environment():
    """The experiment knobs of the host-batch schedule (csrc/swmi_host.h Knobs, read once by the first call that needs them):
    SWMI_HOST_GRANULE, SWMI_HOST_TAPER + SWMI_HOST_MIN_GRANULE, SWMI_HOST_SCHEDULE (explicit list, the last entry repeats),
    SWMI_TEST_SCORE_GROUP.  One child process per setting: the library reads its environment once."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import swmi; "
            "print(swmi.host_granules(1 << 20, 0)); print(swmi.host_granules(1 << 20, 1)); print(swmi.host_granules(300000, 2))" % PKG)

    def run(**env):
        clean = {k: v for k, v in os.environ.items() if not k.startswith("SWMI_")}
        r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(clean, **env))
        assert r.returncode == 0, r.stderr
        return [eval(line) for line in r.stdout.strip().splitlines()]
    pairs, packed, ovm = run(SWMI_HOST_GRANULE="262144")
    assert pairs == [262144] * 4 and packed == [262144] * 4 and ovm == [262144, 37856]
    pairs, packed, ovm = run(SWMI_HOST_TAPER="50", SWMI_HOST_MIN_GRANULE="65536")
    assert pairs == [524288, 262144, 131072, 65536, 65536] and packed == pairs and ovm == [147456, 73728, 65536, 13280]
    pairs, packed, ovm = run(SWMI_HOST_SCHEDULE="32768,65536,131072")
    assert pairs == [32768, 65536] + [131072] * 7 + [32768] and packed == pairs and ovm == [32768, 65536, 131072, 70624]
    pairs, packed, ovm = run(SWMI_TEST_SCORE_GROUP="524288")                 # two groups of 512K pairs, each with its own schedule
    assert pairs == [393216, 98304, 24576, 8192] * 2 and sum(packed) == 1 << 20 and packed[: len(packed) // 2] == packed[len(packed) // 2:]
    pairs, packed, ovm = run(SWMI_HOST_SCHEDULE="12,nonsense")               # entries outside [1024, 1M] are ignored: the default schedule
    assert pairs == [786432, 196608, 49152, 16384]
