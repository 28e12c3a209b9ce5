#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs oracle/_ref/libswref.so, which oracle/Makefile
compiles from /root/reference/source.cpp where it lies):

    make -C oracle && python tests/golden/make_golden.py

The fixtures are DATA: input bytes, parameters and the int32 scores the reference's own
functions returned (scalar SmithWaterman source.cpp:35-60 for every vector; simd4
source.cpp:462-571, simd7 :758-850 and simd9 :953-1071 are required to agree wherever
the parameters lie in their valid domain).  Nothing of the
reference's source text is stored.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libswref.so"))
u8p = ctypes.POINTER(ctypes.c_uint8)
i8p = ctypes.POINTER(ctypes.c_int8)
i32p = ctypes.POINTER(ctypes.c_int32)
ref.swref_repeat.restype = ctypes.c_longlong


def P(a, t):
    return a.ctypes.data_as(t)


def ref_batch(variant, s1, s2, sm, gap):
    n = s1.shape[0]
    out = np.zeros(n, np.int32)
    s1 = np.ascontiguousarray(s1)
    s2 = np.ascontiguousarray(s2)
    sm = np.ascontiguousarray(sm, dtype=np.int8)
    rc = ref.swref_batch(variant, P(s1, u8p), P(s2, u8p), ctypes.c_size_t(n), P(sm, i8p), int(gap), P(out, i32p))
    assert rc == 0
    return out


def match_matrix(match, mismatch):
    sm = np.full((4, 4), mismatch, np.int8)
    np.fill_diagonal(sm, match)
    return sm.reshape(16)


def in_simd_domain(sm, gap):
    return sm.min() >= -127 and 0 <= gap <= 127


def in_simd9_domain(sm, gap):
    return in_simd_domain(sm, gap) and sm.min() >= -100 - gap and sm.max() <= 155 - gap


def score_all(s1, s2, params):
    """scores[p, k] from the reference scalar; SIMD variants asserted equal inside their domains."""
    out = np.zeros((len(params), s1.shape[0]), np.int32)
    for p, (sm, gap) in enumerate(params):
        sc = ref_batch(0, s1, s2, sm, gap)
        if in_simd_domain(sm, gap):
            for v in (1, 4, 7, 8):
                assert (ref_batch(v, s1, s2, sm, gap) == sc).all(), ("simd%d disagrees" % v, sm, gap)
        if in_simd9_domain(sm, gap):
            assert (ref_batch(9, s1, s2, sm, gap) == sc).all(), ("simd9 disagrees", sm, gap)
        out[p] = sc
    return out


def pack_params(params):
    sms = np.stack([np.asarray(sm, np.int8) for sm, _ in params])
    gaps = np.asarray([g for _, g in params], np.int32)
    return sms, gaps


def structured_pairs(rng):
    s1, s2 = [], []

    def add(a, b):
        s1.append(np.asarray(a, np.uint8))
        s2.append(np.asarray(b, np.uint8))

    rnd = lambda: rng.integers(0, 4, 128, dtype=np.uint8)
    for _ in range(16):                      # identical pairs -> 128 * match
        a = rnd(); add(a, a.copy())
    for _ in range(16):                      # reversed
        a = rnd(); add(a, a[::-1].copy())
    for x in range(4):                       # homopolymer vs homopolymer
        for y in range(4):
            add(np.full(128, x), np.full(128, y))
    for x in range(4):                       # homopolymer vs random
        a = rnd(); add(np.full(128, x), a); add(a, np.full(128, x))
    for k in range(32):                      # single deletion / insertion at varying positions
        a = rnd(); pos = 1 + 4 * k
        b = np.concatenate([a[:pos], a[pos + 1:], rng.integers(0, 4, 1, dtype=np.uint8)])
        add(a, b); add(b, a)
    for ident in (0.95, 0.9, 0.8, 0.7, 0.6, 0.5):   # substitutions only
        for _ in range(8):
            a = rnd(); b = a.copy()
            m = rng.random(128) > ident
            b[m] = (b[m] + rng.integers(1, 4, m.sum())) % 4
            add(a, b)
    for _ in range(32):                      # indel-rich ~80 % similar (in the spirit of source.cpp:2748-2771)
        a = rnd(); b = []
        i = 0
        while len(b) < 128:
            r = rng.random()
            if r < 0.05: b.append(rng.integers(0, 4))            # insertion
            elif r < 0.10: i += 1                                # deletion
            else:
                b.append(a[i % 128] if rng.random() < 0.9 else rng.integers(0, 4)); i += 1
        add(a, np.asarray(b[:128]))
    for _ in range(8):                       # all-mismatch (complement) and shifted copies
        a = rnd(); add(a, (a + 1) % 4)
    for sh in (1, 2, 3, 5, 8, 13, 21, 34, 55, 64, 89, 100, 120, 127):
        a = rnd(); add(a, np.roll(a, sh)); add(np.roll(a, sh), a)
    for per in (2, 3, 4, 5, 7):              # tandem repeats
        unit = rng.integers(0, 4, per, dtype=np.uint8)
        a = np.tile(unit, 128 // per + 1)[:128]
        add(a, np.roll(a, 1)); add(a, rnd())
    return np.stack(s1), np.stack(s2)


def main():
    rng = np.random.default_rng(20261004)
    core_params = [
        (match_matrix(10, -30), 15),    # SpeedTest / TestSimdSmithWaterman, source.cpp:3041-3046, 2954-2959
        (match_matrix(1, -1), 1),       # speedtest111x32, source.cpp:3202-3207
        (match_matrix(2, -3), 5),
        (match_matrix(5, -4), 0),
        (match_matrix(127, -127), 127),
        (match_matrix(127, -127), 1),
        (match_matrix(100, -100), 28),
        (match_matrix(1, 0), 0),
        (match_matrix(0, 0), 0),
        (match_matrix(10, 5), 3),
        (match_matrix(127, 127), 0),
    ]
    sms, gaps = pack_params(core_params)

    # F1: iid random pairs
    n1 = 4096
    s1 = rng.integers(0, 4, (n1, 128), dtype=np.uint8)
    s2 = rng.integers(0, 4, (n1, 128), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "f1_random.npz"), seq1=s1, seq2=s2, sm=sms, gap=gaps,
                        scores=score_all(s1, s2, core_params))

    # F2: structured pairs
    t1, t2 = structured_pairs(rng)
    np.savez_compressed(os.path.join(HERE, "f2_structured.npz"), seq1=t1, seq2=t2, sm=sms, gap=gaps,
                        scores=score_all(t1, t2, core_params))

    # F3: the reference drivers' own input stream (libstdc++ mt19937_64(10000), interleaved a[i], b[i])
    n3 = 100000
    h1 = np.zeros((n3, 128), np.uint8)
    h2 = np.zeros((n3, 128), np.uint8)
    ref.swref_harness_stream(ctypes.c_uint64(10000), ctypes.c_size_t(n3), P(h1, u8p), P(h2, u8p))
    hp = core_params[:2]
    hs = score_all(h1, h2, hp)
    hsm, hgap = pack_params(hp)
    keep = 512
    np.savez_compressed(os.path.join(HERE, "f3_harness.npz"), seq1=h1[:keep], seq2=h2[:keep], sm=hsm, gap=hgap,
                        scores=hs[:, :keep], sum_first_100000=hs.sum(axis=1).astype(np.int64),
                        min_first_100000=hs.min(axis=1), max_first_100000=hs.max(axis=1))

    # F4: parameter sweep -- random (asymmetric) int8 matrices and gaps, 192 random + 64 similar pairs each
    n4 = 256
    p4 = []
    for k in range(96):
        lo, hi = [(-127, 127), (-20, 20), (-5, 12), (-127, 0), (0, 127), (-60, 90)][k % 6]
        sm = rng.integers(lo, hi + 1, 16).astype(np.int8)
        gap = int(rng.choice([0, 1, 2, 3, 7, 15, 31, 64, 100, 126, 127]))
        p4.append((sm, gap))
    for k in range(8):   # sm = -128 entries: outside the SIMD variants' domain, scalar semantics only
        sm = rng.integers(-128, 128, 16).astype(np.int8)
        sm[rng.integers(0, 16, 3)] = -128
        p4.append((sm, int(rng.choice([0, 1, 9, 127]))))
    a4 = rng.integers(0, 4, (n4, 128), dtype=np.uint8)
    b4 = rng.integers(0, 4, (n4, 128), dtype=np.uint8)
    for k in range(192, 256):
        m = rng.random(128) > 0.85
        b4[k] = a4[k]
        b4[k][m] = rng.integers(0, 4, m.sum())
    sm4, gap4 = pack_params(p4)
    np.savez_compressed(os.path.join(HERE, "f4_param_sweep.npz"), seq1=a4, seq2=b4, sm=sm4, gap=gap4,
                        scores=score_all(a4, b4, p4))

    # F5: sibling functions for the "next" rows: (1,1,1) scorers and 2-bit unpack
    n5 = 1024
    a5 = rng.integers(0, 4, (n5, 128), dtype=np.uint8)
    b5 = rng.integers(0, 4, (n5, 128), dtype=np.uint8)
    b5[512:] = np.where(rng.random((512, 128)) > 0.8, rng.integers(0, 4, (512, 128)), a5[512:]).astype(np.uint8)
    s111 = np.array([ref.swref_score_111(P(a5[k], u8p), P(b5[k], u8p)) for k in range(n5)], np.int32)
    s8 = np.array([ref.swref_score_8bit111simd(P(a5[k], u8p), P(b5[k], u8p)) for k in range(n5)], np.int32)
    assert (s111 == s8).all()
    assert (s111 == ref_batch(0, a5, b5, match_matrix(1, -1), 1)).all()
    x32 = np.zeros((n5 // 32, 32), np.int32)
    for blk in range(n5 // 32):             # 32 seq1 x 1 seq2 (source.cpp:1227-1230): seq2 = b5[blk]
        blk1 = np.ascontiguousarray(a5[32 * blk:32 * blk + 32])
        d = {}
        for mark in (1, 2, 3):
            o = np.zeros(32, np.int32)
            ref.swref_111x32(mark, P(blk1, u8p), P(b5[blk], u8p), P(o, i32p))
            d[mark] = o
        assert (d[1] == d[2]).all() and (d[1] == d[3]).all()
        x32[blk] = d[1]
    packed = rng.integers(0, 256, (256, 32), dtype=np.uint8)
    unpacked = np.zeros((256, 128), np.uint8)
    for k in range(256):
        ref.swref_unpack(P(packed[k], u8p), P(unpacked[k], u8p))
    # one-vs-many with GENERAL parameters (the shape of source.cpp:1227-1230 behind swmi_score_one_vs_many): every seq1 of
    # F5 against seq2[0], from the reference scalar with the SIMD variants agreeing (score_all).  Draws nothing from rng, so
    # the fixtures generated after this point are unchanged.
    ovm_params = [core_params[k] for k in (0, 1, 2, 3, 4, 5, 6, 9)]
    ovm_sm, ovm_gap = pack_params(ovm_params)
    ovm_scores = score_all(a5, np.repeat(b5[:1], n5, axis=0), ovm_params)
    np.savez_compressed(os.path.join(HERE, "f5_siblings.npz"), seq1=a5, seq2=b5, scores_111=s111,
                        scores_111x32=x32, packed=packed, unpacked=unpacked,
                        ovm_sm=ovm_sm, ovm_gap=ovm_gap, ovm_scores=ovm_scores)
    # F6: semi-global adaptive-band X-drop aligner (source.cpp:1836-2725), SURVEY 8f row N4.  Scalar and the four SIMD
    # variants must return the same (score, traceback); the traceback is stored as one move code per step
    # (1 = diagonal, 2 = down (i+1), 3 = right (j+1)) from (0,0).
    n6 = 8
    g1 = np.zeros((n6, 16384), np.uint8)
    g2 = np.zeros((n6, 16384), np.uint8)
    ref.swref_semiglobal_stream(ctypes.c_uint64(10000), ctypes.c_size_t(n6), P(g1, u8p), P(g2, u8p))   # TestSemiGlobal inputs
    base = rng.integers(0, 4, 16384, dtype=np.uint8)
    extra = [
        (base, np.where(rng.random(16384) < 0.95, base, rng.integers(0, 4, 16384, dtype=np.uint8)).astype(np.uint8)),  # SpeedtestSemiGlobal shape
        (base, rng.integers(0, 4, 16384, dtype=np.uint8)),                                   # unrelated
        (base, base.copy()),                                                                 # identical
        (base, np.roll(base, 20)),                                                           # constant offset inside the band's reach
        (base, np.concatenate([base[:5000], base[5040:], rng.integers(0, 4, 40, dtype=np.uint8)])),    # 40-base deletion
        (base, np.concatenate([base[:8000], rng.integers(0, 4, 100, dtype=np.uint8), base[8000:-100]])),  # 100-base insertion: X-drop
        (np.zeros(16384, np.uint8), np.zeros(16384, np.uint8)),                              # homopolymer
        (np.tile(np.array([0, 1, 2, 3], np.uint8), 4096), np.tile(np.array([0, 1, 2, 3], np.uint8), 4096)[::-1].copy()),
    ]
    s1 = np.concatenate([g1, np.stack([e[0] for e in extra])])
    s2 = np.concatenate([g2, np.stack([e[1] for e in extra])])
    sg_scores, sg_lens, sg_moves, sg_ends = [], [], [], []
    tb = np.zeros((40000, 2), np.int32)
    for k in range(s1.shape[0]):
        res = {}
        for v in (0, 1, 2, 3, 4):
            sc, ln = ctypes.c_int32(), ctypes.c_size_t()
            a, b = np.ascontiguousarray(s1[k]), np.ascontiguousarray(s2[k])
            assert ref.swref_semiglobal(v, P(a, u8p), P(b, u8p), ctypes.byref(sc), P(tb, i32p), ctypes.c_size_t(40000), ctypes.byref(ln)) == 0
            res[v] = (sc.value, tb[: ln.value].copy())
        for v in (1, 2, 3, 4):
            assert res[v][0] == res[0][0] and np.array_equal(res[v][1], res[0][1]), ("semi-global variant %d disagrees" % v, k)
        path = res[0][1]
        assert tuple(path[0]) == (0, 0)
        d = np.diff(path, axis=0)
        moves = np.where((d[:, 0] == 1) & (d[:, 1] == 1), 1, np.where(d[:, 0] == 1, 2, 3)).astype(np.uint8)
        assert ((d == [1, 1]).all(1) | (d == [1, 0]).all(1) | (d == [0, 1]).all(1)).all()
        sg_scores.append(res[0][0]); sg_lens.append(len(path)); sg_moves.append(moves); sg_ends.append(path[-1])
    np.savez_compressed(os.path.join(HERE, "f6_semiglobal.npz"), seq1=s1, seq2=s2, scores=np.array(sg_scores, np.int32),
                        lengths=np.array(sg_lens, np.int32), ends=np.array(sg_ends, np.int32),
                        moves=np.concatenate(sg_moves), move_offsets=np.cumsum([0] + [len(m) for m in sg_moves]).astype(np.int64))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    sys.exit(main())
