import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before libswmi.so: a process that uses both must load torch's HIP runtime first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "smith-waterman-simd_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    need = [os.path.join(PKG, "lib", "libswmi.so"), os.path.join(ROOT, "oracle", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    import __graft_entry__
    __graft_entry__.build()


_ensure_built()


class Oracle:
    """ctypes view of oracle/liboracle.so -- the CPU checker (test infrastructure only)."""

    def __init__(self):
        self.lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        self.lib.sw_oracle_score.restype = ctypes.c_int

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def score(self, seq1, seq2, sm, gap):
        a = np.ascontiguousarray(seq1, np.uint8)
        b = np.ascontiguousarray(seq2, np.uint8)
        m = np.ascontiguousarray(sm, np.int8)
        return self.lib.sw_oracle_score(self._p(a), self._p(b), self._p(m), int(gap))

    def batch(self, seq1s, seq2s, sm, gap):
        a = np.ascontiguousarray(seq1s, np.uint8)
        b = np.ascontiguousarray(seq2s, np.uint8)
        m = np.ascontiguousarray(sm, np.int8)
        n = a.size // 128
        out = np.zeros(n, np.int32)
        self.lib.sw_oracle_batch(self._p(a), self._p(b), ctypes.c_size_t(n), self._p(m), int(gap), self._p(out))
        return out

    def banded_affine(self, seq1s, seq2s, sm, gap_open, gap_extend):
        a = np.ascontiguousarray(seq1s, np.uint8)
        b = np.ascontiguousarray(seq2s, np.uint8)
        m = np.ascontiguousarray(sm, np.int8)
        n, length = a.shape
        out = np.zeros(n, np.int32)
        self.lib.sw_oracle_banded_affine_batch(self._p(a), self._p(b), ctypes.c_size_t(n), int(length), self._p(m),
                                               int(gap_open), int(gap_extend), self._p(out))
        return out

    def semiglobal(self, seq1, seq2):
        """(score, traceback[(len, 2)]) of oracle/sg_oracle.c for one pair of 16384-mers."""
        a = np.ascontiguousarray(seq1, np.uint8)
        b = np.ascontiguousarray(seq2, np.uint8)
        tb = np.zeros((32769, 2), np.int32)
        score, ln, oob = ctypes.c_int32(), ctypes.c_size_t(), ctypes.c_int()
        rc = self.lib.sg_oracle_xdrop(self._p(a), self._p(b), ctypes.byref(score), self._p(tb), ctypes.c_size_t(32769),
                                      ctypes.byref(ln), ctypes.byref(oob))
        assert rc == 0, "sg_oracle_xdrop failed (%d)" % rc
        return score.value, tb[: ln.value].copy()

    def calm_windows(self, seq1, seq2, window=8, margin=13):
        """(windows, calm windows, cells dropped inside calm windows) of oracle/sg_oracle.c's sg_oracle_calm_windows."""
        a = np.ascontiguousarray(seq1, np.uint8)
        b = np.ascontiguousarray(seq2, np.uint8)
        counts = (ctypes.c_long * 3)()
        assert self.lib.sg_oracle_calm_windows(self._p(a), self._p(b), int(window), int(margin), counts) == 0
        return int(counts[0]), int(counts[1]), int(counts[2])

    def generate(self, n, seed, first_pair=0):
        a = np.zeros((n, 128), np.uint8)
        b = np.zeros((n, 128), np.uint8)
        self.lib.sw_oracle_generate(self._p(a), self._p(b), ctypes.c_size_t(n), ctypes.c_uint64(seed), ctypes.c_uint64(first_pair))
        return a, b

    def unpack(self, packed):
        p = np.ascontiguousarray(packed, np.uint8).reshape(-1, 32)
        out = np.zeros((p.shape[0], 128), np.uint8)
        for k in range(p.shape[0]):
            self.lib.sw_oracle_unpack(self._p(p[k]), self._p(out[k]))
        return out

    def pack(self, seqs):
        s = np.ascontiguousarray(seqs, np.uint8).reshape(-1, 128)
        out = np.zeros((s.shape[0], 32), np.uint8)
        for k in range(s.shape[0]):
            self.lib.sw_oracle_pack(self._p(s[k]), self._p(out[k]))
        return out


@pytest.fixture(scope="session")
def oracle():
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def swmi_mod():
    import swmi
    swmi.load()
    return swmi


@pytest.fixture(scope="session")
def gpu(swmi_mod):
    """Initialised library on device 0. Fails (never skips) when the HIP path is unavailable."""
    swmi_mod.init(0)
    swmi_mod.set_schedule(0, 0)
    yield swmi_mod
    swmi_mod.set_schedule(0, 0)


def match_matrix(match, mismatch):
    sm = np.full((4, 4), mismatch, np.int8)
    np.fill_diagonal(sm, match)
    return sm.reshape(16)
