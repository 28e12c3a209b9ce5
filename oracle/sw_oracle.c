/*
 * sw_oracle.c -- TEST INFRASTRUCTURE ONLY. Not part of the product path.
 *
 * Plain-C CPU restatement of the scoring recurrence of eukaryo/smith-waterman-simd's
 * hot path. It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can CHECK the HIP kernels; nothing under smith-waterman-simd_amd/
 * may include, link or call it.
 *
 * Parity pinning: this file is validated (tests/test_oracle_golden.py) against
 * the .npz fixtures under tests/golden/, which were produced by the real reference compiled in the
 * build container (oracle/Makefile -> oracle/_ref/libswref.so, scalar AND simd4/7/9
 * agreeing) by tests/golden/make_golden.py.
 *
 * What is restated (reference file:line):
 *   source.cpp:35-60   SmithWaterman()            -> sw_oracle_score()
 *       H(i,j) = max(0, H(i-1,j-1) + sm[seq1[i-1]*4 + seq2[j-1]],
 *                       H(i-1,j) - gap, H(i,j-1) - gap), H(0,.) = H(.,0) = 0,
 *       answer = max over all 128 x 128 cells.   (recurrence: source.cpp:49-53)
 *   source.cpp:1073-1103 SmithWaterman_111()      -> sw_oracle_score() with sm = +1/-1, gap = 1
 *   source.cpp:1580-1583 unpack()                 -> sw_oracle_unpack()  (base k of byte i = (src[i] >> 2k) & 3)
 *   source.cpp:3032-3082 SpeedTest() call loop    -> sw_oracle_batch()   (one call per pair)
 *
 * Differences from the reference, on purpose:
 *   - the 129 x 129 int table of source.cpp:43 is replaced by one rolling row
 *     (same values, 516 B instead of 66 KB);
 *   - bases are masked with & 3 before indexing the 4 x 4 matrix; the reference
 *     indexes std::array<int8_t,16> out of range for bases >= 4 (undefined behaviour,
 *     source.cpp:50), the build defines it as "base mod 4" on both CPU and GPU.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SW_LEN 128

int sw_oracle_score(const uint8_t *seq1, const uint8_t *seq2, const int8_t *sm, int gap)
{
    int row[SW_LEN + 1];          /* row[j] = H(i-1, j) while row i is being produced */
    int best = 0;
    memset(row, 0, sizeof row);
    for (int i = 1; i <= SW_LEN; ++i) {
        const int8_t *srow = sm + 4 * (seq1[i - 1] & 3);
        int diag = 0;             /* H(i-1, j-1) */
        int left = 0;             /* H(i,   j-1) */
        for (int j = 1; j <= SW_LEN; ++j) {
            const int up = row[j];
            int h = diag + srow[seq2[j - 1] & 3];
            if (up - gap > h) h = up - gap;
            if (left - gap > h) h = left - gap;
            if (h < 0) h = 0;
            if (h > best) best = h;
            diag = up;
            row[j] = h;
            left = h;
        }
    }
    return best;
}

/* pair k lives at byte offset 128*k of each array (the layout of include/swmi.h) */
void sw_oracle_batch(const uint8_t *seq1s, const uint8_t *seq2s, size_t n,
                     const int8_t *sm, int gap, int32_t *scores)
{
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < (long long)n; ++k)
        scores[k] = sw_oracle_score(seq1s + (size_t)k * SW_LEN, seq2s + (size_t)k * SW_LEN, sm, gap);
}

/* single-thread variant, used for the cpu_baseline "port" timing (cores = 1) */
void sw_oracle_batch_st(const uint8_t *seq1s, const uint8_t *seq2s, size_t n,
                        const int8_t *sm, int gap, int32_t *scores)
{
    for (size_t k = 0; k < n; ++k)
        scores[k] = sw_oracle_score(seq1s + k * SW_LEN, seq2s + k * SW_LEN, sm, gap);
}

/* source.cpp:1580-1583: 32 packed bytes -> 128 one-byte bases, 2 bits each, LSB first */
void sw_oracle_unpack(const uint8_t *src32, uint8_t *dst128)
{
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 4; ++k)
            dst128[4 * i + k] = (uint8_t)((src32[i] >> (2 * k)) & 3);
}

void sw_oracle_pack(const uint8_t *src128, uint8_t *dst32)
{
    for (int i = 0; i < 32; ++i) {
        unsigned v = 0;
        for (int k = 0; k < 4; ++k) v |= (unsigned)(src128[4 * i + k] & 3) << (2 * k);
        dst32[i] = (uint8_t)v;
    }
}

/* one-vs-many shape of source.cpp:1227-1230 (32 seq1 x 1 seq2 -> int[32]) with general parameters */
void sw_oracle_one_vs_many(const uint8_t *seq1s, size_t n_seq1, const uint8_t *seq2,
                           const int8_t *sm, int gap, int32_t *scores)
{
    for (size_t k = 0; k < n_seq1; ++k)
        scores[k] = sw_oracle_score(seq1s + k * SW_LEN, seq2, sm, gap);
}

/*
 * Synthetic-input generator shared (by specification, not by code) with the device
 * generator in smith-waterman-simd_amd/csrc: pair p, sequence s (0 = seq1, 1 = seq2),
 * 64-bit word w (0..3) of 32 bases each:
 *     x = splitmix64(seed ^ ((p * 2 + s) * 4 + w) * 0x9E3779B97F4A7C15)
 *     base k of the word = (x >> (2k)) & 3
 * (std::uniform_int_distribution of source.cpp:3034 is implementation-defined, so the
 * build never relies on it)
 */
static uint64_t sw_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void sw_oracle_generate(uint8_t *seq1s, uint8_t *seq2s, size_t n, uint64_t seed, uint64_t first_pair)
{
    for (size_t k = 0; k < n; ++k) {
        const uint64_t p = first_pair + k;
        for (int s = 0; s < 2; ++s) {
            uint8_t *dst = (s ? seq2s : seq1s) + k * SW_LEN;
            for (int w = 0; w < 4; ++w) {
                uint64_t x = sw_splitmix64(seed ^ (((p * 2 + (uint64_t)s) * 4 + (uint64_t)w) * 0x9E3779B97F4A7C15ull));
                for (int b = 0; b < 32; ++b) dst[32 * w + b] = (uint8_t)((x >> (2 * b)) & 3);
            }
        }
    }
}


/*
 * BASELINE.json configs[4] ("1024 x 1024 affine-gap, band width 128") -- an EXTENSION with no counterpart in the
 * reference (source.cpp has linear gaps only): PARITY UNPINNED BY THE REFERENCE.  This scalar
 * banded Gotoh defines the semantics the GPU kernel is tested against; tests/test_banded_affine.py cross-checks it
 * against an independent full-matrix numpy formulation and against the pinned linear-gap scorer where the two must agree.
 *
 *   cells (i, j), 1 <= i, j <= len, inside the band  -64 <= j - i <= 63   (128 diagonals)
 *   E(i,j) = max(E(i,j-1) - ext,  H(i,j-1) - open)        gap of length k costs open + (k-1) * ext
 *   F(i,j) = max(F(i-1,j) - ext,  H(i-1,j) - open)
 *   H(i,j) = max(0, H(i-1,j-1) + sm[seq1[i-1]*4 + seq2[j-1]], E(i,j), F(i,j))
 *   outside the band / the matrix: H = 0, E = F = -infinity;  score = max H over the band
 */
#define SW_BAND_LO (-64)
#define SW_BAND_HI (63)
#define SW_NEG_INF (-(1 << 29))

int sw_oracle_banded_affine(const uint8_t *seq1, const uint8_t *seq2, int len, const int8_t *sm, int open, int ext)
{
    /* full (len+1)^2 tables keep the restatement obvious; len <= 4096 */
    const size_t w = (size_t)len + 1;
    int *H = (int *)calloc(w * w, sizeof(int));
    int *E = (int *)malloc(w * w * sizeof(int));
    int *F = (int *)malloc(w * w * sizeof(int));
    int best = 0;
    if (!H || !E || !F) { free(H); free(E); free(F); return -1; }
    for (size_t k = 0; k < w * w; ++k) E[k] = F[k] = SW_NEG_INF;
    for (int i = 1; i <= len; ++i) {
        for (int j = 1; j <= len; ++j) {
            const int d = j - i;
            if (d < SW_BAND_LO || d > SW_BAND_HI) continue;       /* stays H = 0, E = F = -inf */
            const size_t c = (size_t)i * w + (size_t)j;
            const int e1 = E[c - 1] - ext, e2 = H[c - 1] - open;
            const int f1 = F[c - w] - ext, f2 = H[c - w] - open;
            int h = H[c - w - 1] + sm[4 * (seq1[i - 1] & 3) + (seq2[j - 1] & 3)];
            E[c] = e1 > e2 ? e1 : e2;
            F[c] = f1 > f2 ? f1 : f2;
            if (E[c] > h) h = E[c];
            if (F[c] > h) h = F[c];
            if (h < 0) h = 0;
            H[c] = h;
            if (h > best) best = h;
        }
    }
    free(H); free(E); free(F);
    return best;
}

void sw_oracle_banded_affine_batch(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int len, const int8_t *sm,
                                   int open, int ext, int32_t *scores)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (long long k = 0; k < (long long)n; ++k)
        scores[k] = sw_oracle_banded_affine(seq1s + (size_t)k * (size_t)len, seq2s + (size_t)k * (size_t)len, len, sm, open, ext);
}
