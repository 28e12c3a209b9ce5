"""ctypes binding of libswmi.so (include/swmi.h) -- the Python-side mirror used by tests/ and bench.py.

The scoring entry points keep the reference's argument meaning (source.cpp:462-466):
``score_pair(seq1, seq2, score_matrix, gap_penalty) -> int`` with 128-byte sequences, a 16-entry int8
matrix indexed ``seq1_base * 4 + seq2_base`` and a non-negative gap penalty.  Nothing here computes
scores on the CPU: every call goes through the C ABI into the gfx950 kernels and raises ``SwmiError``
when the library or a usable device is missing.
"""
import ctypes
import os

import numpy as np

SEQ_LEN = 128
PACKED_LEN = 32

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWMI_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libswmi.so"))   # SWMI_LIB: A/B builds (tools/)

OK = 0
ERR_NOT_INITIALIZED = -1
ERR_NO_DEVICE = -2
ERR_UNSUPPORTED_ARCH = -3
ERR_INVALID_ARGUMENT = -4
ERR_DOMAIN = -5
ERR_ALIGNMENT = -6
ERR_HIP = -7
ERR_QUEUE_FULL = -8

NO_GAP_FOLD = 1
USE_I16 = 2
USE_LUT = 4
NO_PACKED = 8


class SwmiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("swmi error %d: %s" % (code, message))
        self.code = code


class DeviceInfo(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("compute_units", ctypes.c_int), ("clock_khz", ctypes.c_int),
                ("wavefront_size", ctypes.c_int), ("hbm_bytes", ctypes.c_size_t), ("arch", ctypes.c_char * 64),
                ("name", ctypes.c_char * 128)]


_lib = None


def load():
    """dlopen libswmi.so (built by smith-waterman-simd_amd/csrc/Makefile). Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SwmiError(ERR_NOT_INITIALIZED, "%s not built: run __graft_entry__.build() or make -C smith-waterman-simd_amd/csrc" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, u64, i8 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int8
    lib.swmi_last_error.restype = ctypes.c_char_p
    lib.swmi_init.argtypes = [ctypes.c_int]
    lib.swmi_init_all.argtypes = [ctypes.c_int]
    lib.swmi_init_devices.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.swmi_use_gpu.argtypes = [ctypes.c_int]
    lib.swmi_shard_bounds.argtypes = [sz, ctypes.c_int, ctypes.c_int, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.swmi_score_batch_multi.argtypes = [vp, vp, sz, vp, i8, vp]
    lib.swmi_score_batch_packed_multi.argtypes = [vp, vp, sz, vp, i8, vp]
    lib.swmi_sharded_create.argtypes = [sz, ctypes.c_int, ctypes.POINTER(vp)]
    lib.swmi_sharded_destroy.argtypes = [vp]
    lib.swmi_sharded_generate.argtypes = [vp, u64, u64]
    lib.swmi_sharded_upload.argtypes = [vp, vp, vp]
    lib.swmi_sharded_score.argtypes = [vp, vp, i8, ctypes.c_int]
    lib.swmi_sharded_wait.argtypes = [vp]
    lib.swmi_sharded_scores_host.argtypes = [vp, vp]
    lib.swmi_sharded_gathered_device.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp)]
    lib.swmi_sharded_gather_backend.argtypes = [vp]
    lib.swmi_sharded_gathered_host.argtypes = [vp, ctypes.c_int, vp]
    lib.swmi_sharded_time.argtypes = [vp, vp, i8, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)]
    lib.swmi_host_granules.argtypes = [sz, vp, sz]
    lib.swmi_host_granules.restype = sz
    lib.swmi_host_granules_for.argtypes = [sz, ctypes.c_int, vp, sz]
    lib.swmi_host_granules_for.restype = sz
    lib.swmi_selftest_pk_max3.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    lib.swmi_sharded_gather_note.argtypes = [vp, ctypes.c_char_p, sz]
    lib.swmi_rccl_probe.argtypes = [ctypes.c_char_p, sz]
    lib.swmi_score_kernel_for_batch.argtypes = [sz, vp, i8, ctypes.c_int, ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_int)]
    lib.swmi_banded_affine_kernel_for.argtypes = [ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_int)]
    lib.swmi_semiglobal_kernels_for_batch.argtypes = [sz, ctypes.c_char_p, sz, ctypes.c_char_p, sz]
    lib.swmi_score_pair.argtypes = [vp, vp, vp, i8]
    lib.swmi_score_batch.argtypes = [vp, vp, sz, vp, i8, vp]
    lib.swmi_score_batch_device.argtypes = [vp, vp, sz, vp, i8, vp, vp]
    lib.swmi_score_one_vs_many.argtypes = [vp, sz, vp, vp, i8, vp]
    lib.swmi_score_one_vs_many_device.argtypes = [vp, sz, vp, vp, i8, vp, vp]
    lib.swmi_score_batch_packed.argtypes = [vp, vp, sz, vp, i8, vp]
    lib.swmi_score_batch_packed_device.argtypes = [vp, vp, sz, vp, i8, vp, vp]
    lib.swmi_unpack.argtypes = [vp, sz, vp]
    lib.swmi_semiglobal_xdrop.argtypes = [vp, vp, sz, vp, vp, sz, vp]
    lib.swmi_semiglobal_xdrop_device.argtypes = [vp, vp, sz, vp, vp, sz, vp, vp]
    lib.swmi_semiglobal_xdrop_moves.argtypes = [vp, vp, sz, vp, vp, vp]
    lib.swmi_semiglobal_xdrop_moves_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    lib.swmi_semiglobal_expand_moves.argtypes = [vp, ctypes.c_uint32, vp, sz]
    lib.swmi_schedule_for_batch.argtypes = [sz]
    lib.swmi_semiglobal_time_device.argtypes = [vp, vp, sz, vp, vp, sz, vp, vp, ctypes.POINTER(ctypes.c_float)]
    lib.swmi_semiglobal_window_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.swmi_semiglobal_set_exact.argtypes = [ctypes.c_int]
    lib.swmi_score_banded_affine.argtypes = [vp, vp, sz, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp]
    lib.swmi_score_banded_affine_device.argtypes = [vp, vp, sz, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.swmi_queue_create.argtypes = [sz, vp, i8, ctypes.POINTER(vp)]
    lib.swmi_queue_submit.argtypes = [vp, vp, vp]
    lib.swmi_queue_submit.restype = ctypes.c_longlong
    lib.swmi_queue_wait.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(sz)]
    lib.swmi_queue_reset.argtypes = [vp]
    lib.swmi_queue_destroy.argtypes = [vp]
    lib.swmi_set_schedule.argtypes = [ctypes.c_int, ctypes.c_uint]
    lib.swmi_get_schedule.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint)]
    lib.swmi_generate_pairs_device.argtypes = [vp, vp, sz, u64, u64, vp]
    lib.swmi_generate_pairs_host.argtypes = [vp, vp, sz, u64, u64]
    lib.swmi_time_batch_device.argtypes = [vp, vp, sz, vp, i8, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    lib.swmi_get_device_info.argtypes = [ctypes.POINTER(DeviceInfo)]
    _lib = lib
    return lib


def _check(rc):
    if rc < 0:
        raise SwmiError(rc, load().swmi_last_error().decode())
    return rc


def _u8(a, shape_last):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.shape[-1] != shape_last:
        raise ValueError("last dimension must be %d" % shape_last)
    return a


def _sm(score_matrix):
    raw = np.asarray(score_matrix).reshape(-1)
    if raw.size != 16:
        raise ValueError("score_matrix must have 16 entries")
    if raw.dtype != np.int8 and (raw.min() < -128 or raw.max() > 127):   # the cast below would wrap silently
        raise SwmiError(ERR_DOMAIN, "score_matrix entries must lie in [-128, 127] (got %d..%d)" % (raw.min(), raw.max()))
    return np.ascontiguousarray(raw, dtype=np.int8)


def _gap(gap_penalty):
    """ctypes converts to int8 without an overflow check (256 would score as gap 0): check the range here, so that the
    C side's domain check sees what the caller meant."""
    g = int(gap_penalty)
    if g < -128 or g > 127:
        raise SwmiError(ERR_DOMAIN, "gap_penalty %d is outside the supported domain [0,127]" % g)
    return g


def match_matrix(match, mismatch):
    """4x4 matrix with `match` on the diagonal and `mismatch` elsewhere (source.cpp:3041-3045)."""
    sm = np.full((4, 4), mismatch, np.int8)
    np.fill_diagonal(sm, match)
    return sm.reshape(16)


def init(device=-1):
    _check(load().swmi_init(device))


def init_all(n_gpus=0):
    """Bind the first n_gpus visible GPUs (0 = all) in this one process; returns the number bound."""
    return _check(load().swmi_init_all(n_gpus))


def init_devices(devices):
    """Bind an explicit device list (a device may repeat: two contexts on one GPU)."""
    arr = (ctypes.c_int * len(devices))(*devices)
    return _check(load().swmi_init_devices(arr, len(devices)))


def num_gpus():
    return int(load().swmi_num_gpus())


def use_gpu(index):
    """Select which bound GPU this thread's single-GPU calls address."""
    _check(load().swmi_use_gpu(index))


def shutdown():
    _check(load().swmi_shutdown())


def shard_bounds(n, shard, n_shards):
    """[lo, hi) of shard `shard` of `n_shards` -- the C library's rule (swmi_shard_bounds), same as sharding.shard_bounds."""
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    _check(load().swmi_shard_bounds(n, shard, n_shards, ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def last_error():
    return load().swmi_last_error().decode()


def set_schedule(lanes_per_alignment=0, flags=0):
    _check(load().swmi_set_schedule(lanes_per_alignment, flags))


def get_schedule():
    lanes, flags = ctypes.c_int(), ctypes.c_uint()
    _check(load().swmi_get_schedule(ctypes.byref(lanes), ctypes.byref(flags)))
    return lanes.value, flags.value


def schedule_for_batch(n):
    """Lanes per alignment a launch of n pairs runs with (what the automatic setting resolves to)."""
    return int(load().swmi_schedule_for_batch(ctypes.c_size_t(n)))


def score_kernel_for_batch(n, score_matrix, gap_penalty, mode=0):
    """(kernel instantiation name, alignments per wavefront) a launch of n pairs with these parameters runs."""
    sm = _sm(score_matrix)
    name, per_wave = ctypes.create_string_buffer(96), ctypes.c_int()
    _check(load().swmi_score_kernel_for_batch(n, sm.ctypes.data, _gap(gap_penalty), mode, name, 96, ctypes.byref(per_wave)))
    return name.value.decode(), per_wave.value


def selftest_pk_max3():
    """(comparisons made, mismatches) of swmi_selftest_pk_max3: v_pk_maximum3_f16 as a packed integer max on [0, 0x7C00)^2."""
    checked, bad = ctypes.c_ulonglong(), ctypes.c_ulonglong()
    _check(load().swmi_selftest_pk_max3(ctypes.byref(checked), ctypes.byref(bad)))
    return checked.value, bad.value


ENTRY_PAIRS, ENTRY_PACKED, ENTRY_ONE_VS_MANY = 0, 1, 2


def host_granules(n, entry=ENTRY_PAIRS):
    """The pipeline granules a host-batch entry cuts n pairs into (needs no device): ENTRY_PAIRS = swmi_score_batch,
    ENTRY_PACKED = swmi_score_batch_packed, ENTRY_ONE_VS_MANY = swmi_score_one_vs_many."""
    count = load().swmi_host_granules_for(n, entry, None, 0)
    buf = (ctypes.c_size_t * max(count, 1))()
    load().swmi_host_granules_for(n, entry, buf, count)
    return [int(buf[k]) for k in range(count)]


def rccl_probe():
    """(usable, reason): whether librccl can be loaded with the entry points the score gather needs (needs no device)."""
    why = ctypes.create_string_buffer(512)
    ok = load().swmi_rccl_probe(why, 512)
    return bool(ok), why.value.decode()


def device_info():
    info = DeviceInfo()
    _check(load().swmi_get_device_info(ctypes.byref(info)))
    return {"device": info.device, "compute_units": info.compute_units, "clock_khz": info.clock_khz,
            "wavefront_size": info.wavefront_size, "hbm_bytes": info.hbm_bytes, "arch": info.arch.decode(),
            "name": info.name.decode()}


def score_pair(seq1, seq2, score_matrix, gap_penalty):
    """Mirror of SmithWaterman_simd4(seq1, seq2, score_matrix, gap_penalty) (source.cpp:462-466)."""
    a, b, sm = _u8(seq1, SEQ_LEN), _u8(seq2, SEQ_LEN), _sm(score_matrix)
    return _check(load().swmi_score_pair(a.ctypes.data, b.ctypes.data, sm.ctypes.data, _gap(gap_penalty)))


def _scores_out(out, n):
    """The result array of a host entry: a fresh one, or the caller's (C-contiguous int32[n]) -- a fresh array's pages are
    first touched by the copy that fills them, inside the call."""
    if out is None:
        return np.zeros(n, np.int32)
    if not (isinstance(out, np.ndarray) and out.dtype == np.int32 and out.flags.c_contiguous and out.size == n):
        raise ValueError("out must be a C-contiguous int32 array of %d scores" % n)
    return out


def score_batch(seq1s, seq2s, score_matrix, gap_penalty, out=None):
    a, b, sm = _u8(seq1s, SEQ_LEN), _u8(seq2s, SEQ_LEN), _sm(score_matrix)
    if a.shape != b.shape:
        raise ValueError("seq1s and seq2s must have the same shape")
    n = a.size // SEQ_LEN
    out = _scores_out(out, n)
    _check(load().swmi_score_batch(a.ctypes.data, b.ctypes.data, n, sm.ctypes.data, _gap(gap_penalty), out.ctypes.data))
    return out


def score_batch_multi(seq1s, seq2s, score_matrix, gap_penalty, packed=False):
    """swmi_score_batch[_packed]_multi: the host batch split over every bound GPU (init_all / init_devices)."""
    width = PACKED_LEN if packed else SEQ_LEN
    a, b, sm = _u8(seq1s, width), _u8(seq2s, width), _sm(score_matrix)
    if a.shape != b.shape:
        raise ValueError("seq1s and seq2s must have the same shape")
    n = a.size // width
    out = np.zeros(n, np.int32)
    fn = load().swmi_score_batch_packed_multi if packed else load().swmi_score_batch_multi
    _check(fn(a.ctypes.data, b.ctypes.data, n, sm.ctypes.data, _gap(gap_penalty), out.ctypes.data))
    return out


GATHER_NONE, GATHER_ROOT, GATHER_ALL = 0, 1, 2


class ShardedBatch:
    """A batch whose shards stay resident on the bound GPUs (swmi_sharded_* in include/swmi.h)."""

    def __init__(self, n, packed=False):
        self._b = ctypes.c_void_p()
        self.n = n
        _check(load().swmi_sharded_create(n, 1 if packed else 0, ctypes.byref(self._b)))

    def generate(self, seed, first_pair=0):
        _check(load().swmi_sharded_generate(self._b, seed, first_pair))

    def upload(self, seq1s, seq2s):
        a = np.ascontiguousarray(seq1s, dtype=np.uint8)
        b = np.ascontiguousarray(seq2s, dtype=np.uint8)
        _check(load().swmi_sharded_upload(self._b, a.ctypes.data, b.ctypes.data))

    def score(self, score_matrix, gap_penalty, gather=GATHER_NONE):
        sm = _sm(score_matrix)
        _check(load().swmi_sharded_score(self._b, sm.ctypes.data, _gap(gap_penalty), gather))

    def wait(self):
        _check(load().swmi_sharded_wait(self._b))

    def scores(self):
        out = np.zeros(self.n, np.int32)
        _check(load().swmi_sharded_scores_host(self._b, out.ctypes.data))
        return out

    def gathered_ptr(self, index=0):
        p = ctypes.c_void_p()
        _check(load().swmi_sharded_gathered_device(self._b, index, ctypes.byref(p)))
        return p.value

    def gather_backend(self):
        return {0: "undecided", 1: "p2p", 2: "rccl"}[_check(load().swmi_sharded_gather_backend(self._b))]

    def gather_note(self):
        """Why SWMI_GATHER_ALL runs on peer copies for this batch ("" while RCCL is in use or nothing is decided)."""
        text = ctypes.create_string_buffer(512)
        _check(load().swmi_sharded_gather_note(self._b, text, 512))
        return text.value.decode()

    def gathered(self, index=0):
        """The full score vector GPU `index` holds after a gather, copied to the host."""
        out = np.zeros(self.n, np.int32)
        _check(load().swmi_sharded_gathered_host(self._b, index, out.ctypes.data))
        return out

    def time(self, score_matrix, gap_penalty, gather=GATHER_NONE, iters=10):
        """({"kernel_ms": [...], "gather_ms": [...], "wall_ms": float}) averaged over `iters` back-to-back calls."""
        g = num_gpus()
        k, ga, w = (ctypes.c_float * g)(), (ctypes.c_float * g)(), ctypes.c_double()
        sm = _sm(score_matrix)
        _check(load().swmi_sharded_time(self._b, sm.ctypes.data, _gap(gap_penalty), gather, iters, k, ga, ctypes.byref(w)))
        return {"kernel_ms": [float(x) for x in k], "gather_ms": [float(x) for x in ga], "wall_ms": float(w.value)}

    def close(self):
        if self._b:
            load().swmi_sharded_destroy(self._b)
            self._b = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def score_one_vs_many(seq1s, seq2, score_matrix, gap_penalty, out=None):
    a, b, sm = _u8(seq1s, SEQ_LEN), _u8(seq2, SEQ_LEN), _sm(score_matrix)
    n = a.size // SEQ_LEN
    out = _scores_out(out, n)
    _check(load().swmi_score_one_vs_many(a.ctypes.data, n, b.ctypes.data, sm.ctypes.data, _gap(gap_penalty), out.ctypes.data))
    return out


def score_batch_packed(seq1s_packed, seq2s_packed, score_matrix, gap_penalty, out=None):
    a, b, sm = _u8(seq1s_packed, PACKED_LEN), _u8(seq2s_packed, PACKED_LEN), _sm(score_matrix)
    n = a.size // PACKED_LEN
    out = _scores_out(out, n)
    _check(load().swmi_score_batch_packed(a.ctypes.data, b.ctypes.data, n, sm.ctypes.data, _gap(gap_penalty), out.ctypes.data))
    return out


def score_banded_affine(seq1s, seq2s, score_matrix, gap_open, gap_extend):
    """Extension (BASELINE configs[4]): banded (128 diagonals) affine-gap local scores of n pairs of len-mers."""
    a = np.ascontiguousarray(seq1s, dtype=np.uint8)
    b = np.ascontiguousarray(seq2s, dtype=np.uint8)
    if a.shape != b.shape or a.ndim != 2:
        raise ValueError("seq1s and seq2s must both be (n, len)")
    n, length = a.shape
    sm = _sm(score_matrix)
    out = np.zeros(n, np.int32)
    _check(load().swmi_score_banded_affine(a.ctypes.data, b.ctypes.data, n, length, sm.ctypes.data, int(gap_open),
                                           int(gap_extend), out.ctypes.data))
    return out


def banded_affine_kernel_for(length, score_matrix, gap_open, gap_extend):
    """(kernel instantiation, alignments per wavefront) a banded-affine launch with these parameters runs (needs no device)."""
    sm = _sm(score_matrix)
    name = ctypes.create_string_buffer(128)
    per = ctypes.c_int()
    _check(load().swmi_banded_affine_kernel_for(int(length), sm.ctypes.data_as(ctypes.c_void_p), int(gap_open), int(gap_extend), name, 128,
                                               ctypes.byref(per)))
    return name.value.decode(), per.value


def score_banded_affine_device(d_seq1s, d_seq2s, n, length, score_matrix, gap_open, gap_extend, d_scores, stream=0):
    sm = _sm(score_matrix)
    _check(load().swmi_score_banded_affine_device(d_seq1s, d_seq2s, n, length, sm.ctypes.data, int(gap_open),
                                                  int(gap_extend), d_scores, stream))


SG_LEN = 16384
SG_MAX_TRACEBACK = 32769


def semiglobal_xdrop(seq1s, seq2s, cap=SG_MAX_TRACEBACK):
    """Mirror of SemiGlobal_AdaptiveBanded_XDrop_111_32_70 (source.cpp:1836-1976) for n pairs of 16384-mers.

    Returns (scores[n], list of n (len_k, 2) int32 arrays = the reference's traceback vectors, lengths[n] = the
    reference's .second.size() per alignment, which exceeds len_k only when `cap` cut the traceback short)."""
    a = np.ascontiguousarray(seq1s, dtype=np.uint8).reshape(-1, SG_LEN)
    b = np.ascontiguousarray(seq2s, dtype=np.uint8).reshape(-1, SG_LEN)
    n = a.shape[0]
    scores = np.zeros(n, np.int32)
    lengths = np.zeros(n, np.uint32)
    tb = np.zeros((n, cap, 2), np.int32)
    _check(load().swmi_semiglobal_xdrop(a.ctypes.data, b.ctypes.data, n, scores.ctypes.data, tb.ctypes.data, cap,
                                        lengths.ctypes.data))
    return scores, [tb[k, : min(int(lengths[k]), cap)].copy() for k in range(n)], lengths


SG_MOVE_WORDS = 1040


def semiglobal_xdrop_moves(seq1s, seq2s):
    """The same alignments with the traceback as MOVES (swmi_semiglobal_xdrop_moves): (scores[n], moves[n, SG_MOVE_WORDS] uint64,
    lengths[n]); move t of alignment k = (moves[k, t // 32] >> 2 * (t % 32)) & 3 in walking order (3 diagonal, 2 up, 1 left)."""
    a = np.ascontiguousarray(seq1s, dtype=np.uint8).reshape(-1, SG_LEN)
    b = np.ascontiguousarray(seq2s, dtype=np.uint8).reshape(-1, SG_LEN)
    n = a.shape[0]
    scores = np.zeros(n, np.int32)
    lengths = np.zeros(n, np.uint32)
    moves = np.zeros((n, SG_MOVE_WORDS), np.uint64)
    _check(load().swmi_semiglobal_xdrop_moves(a.ctypes.data, b.ctypes.data, n, scores.ctypes.data, moves.ctypes.data, lengths.ctypes.data))
    return scores, moves, lengths


def semiglobal_expand_moves(moves_row, length, cap=None):
    """One alignment's moves -> the reference's (length, 2) traceback array, on the host (swmi_semiglobal_expand_moves)."""
    row = np.ascontiguousarray(moves_row, dtype=np.uint64)
    cap = int(length) if cap is None else int(cap)
    tb = np.zeros((min(int(length), cap), 2), np.int32)
    _check(load().swmi_semiglobal_expand_moves(row.ctypes.data, int(length), tb.ctypes.data, cap))
    return tb


def semiglobal_xdrop_moves_device(d_seq1s, d_seq2s, n, d_scores, d_moves, d_lengths, stream=0):
    _check(load().swmi_semiglobal_xdrop_moves_device(d_seq1s, d_seq2s, n, d_scores, d_moves, d_lengths, stream))


def semiglobal_xdrop_device(d_seq1s, d_seq2s, n, d_scores, d_tracebacks, cap, d_lengths, stream=0):
    _check(load().swmi_semiglobal_xdrop_device(d_seq1s, d_seq2s, n, d_scores, d_tracebacks, cap, d_lengths, stream))


def semiglobal_time_device(d_seq1s, d_seq2s, n, d_scores, d_tracebacks, cap, d_lengths, stream=0):
    """(sweep_ms, traceback_ms) of one device call, from HIP events on `stream`."""
    ms = (ctypes.c_float * 2)()
    _check(load().swmi_semiglobal_time_device(d_seq1s, d_seq2s, n, d_scores, d_tracebacks, cap, d_lengths, stream, ms))
    return float(ms[0]), float(ms[1])


def semiglobal_set_mapping(sweep=-1):
    """Override which semi-global sweep runs (-1 = automatic); see swmi_semiglobal_set_mapping in include/swmi.h."""
    _check(load().swmi_semiglobal_set_mapping(int(sweep)))


def semiglobal_set_exact(exact_only=False):
    """True: the sweeps run the X-drop test in every round (no calm windows); same results, see include/swmi.h."""
    _check(load().swmi_semiglobal_set_exact(1 if exact_only else 0))


def semiglobal_window_stats(stream=0, walk=False):
    """(windows of 8 rounds the sweep wavefronts of the last device call on `stream` ran, how many of them were calm); with
    walk=True also (windows of 16 rounds the traceback wavefronts walked, how many of them twice: a walk left cells 8 .. 23)."""
    c = (ctypes.c_uint64 * 4)()
    _check(load().swmi_semiglobal_window_stats(ctypes.c_void_p(stream), c))
    return (int(c[0]), int(c[1]), int(c[2]), int(c[3])) if walk else (int(c[0]), int(c[1]))


def semiglobal_kernels_for_batch(n):
    """(sweep kernel name, traceback kernel name) a call with n alignments runs on the current GPU."""
    a, b = ctypes.create_string_buffer(96), ctypes.create_string_buffer(96)
    _check(load().swmi_semiglobal_kernels_for_batch(n, a, 96, b, 96))
    return a.value.decode(), b.value.decode()


def unpack(packed):
    p = _u8(packed, PACKED_LEN)
    n = p.size // PACKED_LEN
    out = np.zeros((n, SEQ_LEN), np.uint8)
    _check(load().swmi_unpack(p.ctypes.data, n, out.ctypes.data))
    return out.reshape(p.shape[:-1] + (SEQ_LEN,))


def pack(seqs):
    """Host-side inverse of unpack() (source.cpp:1581): base k of byte i at bits 2k..2k+1."""
    s = _u8(seqs, SEQ_LEN).reshape(-1, 32, 4) & 3
    return (s[..., 0] | (s[..., 1] << 2) | (s[..., 2] << 4) | (s[..., 3] << 6)).astype(np.uint8).reshape(seqs.shape[:-1] + (32,))


def score_batch_device(d_seq1s, d_seq2s, n, score_matrix, gap_penalty, d_scores, stream=0, packed=False):
    """Device pointers (ints). Asynchronous on `stream` (a hipStream_t value, 0 = the HIP null stream)."""
    sm = _sm(score_matrix)
    fn = load().swmi_score_batch_packed_device if packed else load().swmi_score_batch_device
    _check(fn(d_seq1s, d_seq2s, n, sm.ctypes.data, _gap(gap_penalty), d_scores, stream))


def score_one_vs_many_device(d_seq1s, n, d_seq2, score_matrix, gap_penalty, d_scores, stream=0):
    sm = _sm(score_matrix)
    _check(load().swmi_score_one_vs_many_device(d_seq1s, n, d_seq2, sm.ctypes.data, _gap(gap_penalty), d_scores, stream))


def generate_pairs_device(d_seq1s, d_seq2s, n, seed, first_pair=0, stream=0):
    _check(load().swmi_generate_pairs_device(d_seq1s, d_seq2s, n, seed, first_pair, stream))


def generate_pairs_host(n, seed, first_pair=0):
    a = np.zeros((n, SEQ_LEN), np.uint8)
    b = np.zeros((n, SEQ_LEN), np.uint8)
    _check(load().swmi_generate_pairs_host(a.ctypes.data, b.ctypes.data, n, seed, first_pair))
    return a, b


def time_batch_device(d_seq1s, d_seq2s, n, score_matrix, gap_penalty, d_scores, stream=0, iters=10):
    sm = _sm(score_matrix)
    ms = ctypes.c_float()
    _check(load().swmi_time_batch_device(d_seq1s, d_seq2s, n, sm.ctypes.data, _gap(gap_penalty), d_scores, stream, iters,
                                         ctypes.byref(ms)))
    return ms.value


class Queue:
    """Deferred queue behind the per-pair signature (swmi_queue_* in include/swmi.h)."""

    def __init__(self, max_pairs, score_matrix, gap_penalty):
        self._q = ctypes.c_void_p()
        sm = _sm(score_matrix)
        _check(load().swmi_queue_create(max_pairs, sm.ctypes.data, _gap(gap_penalty), ctypes.byref(self._q)))

    def submit(self, seq1, seq2):
        a, b = _u8(seq1, SEQ_LEN), _u8(seq2, SEQ_LEN)
        return _check(load().swmi_queue_submit(self._q, a.ctypes.data, b.ctypes.data))

    def wait(self):
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        _check(load().swmi_queue_wait(self._q, ctypes.byref(ptr), ctypes.byref(n)))
        if n.value == 0:
            return np.zeros(0, np.int32)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int32)), shape=(n.value,)).copy()

    def reset(self):
        _check(load().swmi_queue_reset(self._q))

    def close(self):
        if self._q:
            load().swmi_queue_destroy(self._q)
            self._q = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
