/*
 * sg_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as sw_oracle.c).
 *
 * Plain-C restatement of the reference's semi-global adaptive-band X-drop aligner
 *     SemiGlobal_AdaptiveBanded_XDrop_111_32_70        source.cpp:1836-1976
 * (match +1, mismatch -1, gap -1, band of 32 anti-diagonal cells, X-drop 70, two 16384-mers, result =
 * (score, traceback as a list of (i, j) from (0, 0) to the best cell)).  SURVEY.md section 8f row N4.
 * Pinned by tests/golden/f6_semiglobal.npz, generated from the real reference (scalar and its four SIMD variants
 * agreeing) by tests/golden/make_golden.py.
 *
 * Algorithm, as the reference runs it (file:line):
 *   - the band holds 32 cells of one anti-diagonal; lane k sits at row pos_y + 31 - k, column pos_x - 62 + k
 *     (the reference indexes padded copies of the sequences instead, :1861-1873); cell values carry an offset of
 *     +70 so that 0 can mean "dropped" (:1880);
 *   - every round the band steps right if lane 0's value is smaller than lane 31's, else down (:1895-1915);
 *   - a cell takes max(diag + s, left - 1, up - 1) over its non-dropped neighbours (:1921-1931), cells more than
 *     70 below the best value so far are dropped (:1938-1941); the sweep ends when a whole round is dropped (:1943)
 *     or the band leaves the padded matrix (:1903, :1913);
 *   - the traceback starts from the first cell of the best round, walking from lane 31 down, that holds the best
 *     value (:1957-1958) and prefers diagonal, then up, then left (:1962-1971).
 *
 * One deliberate difference: the reference reads one byte past each padded sequence when the band reaches the very
 * last position (seq1p[16416], seq2p[16447]; :1917-1919 with the bounds of :1903/:1913) -- undefined behaviour there,
 * "pad" here.  sg_oracle_xdrop() reports through *touched_oob whether an input reaches that state, and the fixture
 * generator refuses such inputs.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SG_LEN 16384
#define SG_BAND 32
#define SG_X 70
#define SG_MAX_ROUND (2 * (SG_LEN + 1) - 1)

typedef struct {
    uint16_t *cell;     /* [round][32] stored values (offset +70, 0 = dropped) */
    int32_t *top_y;     /* [round] row of lane 31 */
    int rounds;         /* rounds written */
} sg_table;

static int sg_get(const sg_table *t, long y, long x)
{
    if (y < 0 || y > SG_LEN || x < 0 || x > SG_LEN) return -1;
    const long r = y + x;
    if (r >= t->rounds) return -1;
    const long lane = 31 - (y - t->top_y[r]);
    if (lane < 0 || lane >= SG_BAND) return -1;
    const int v = t->cell[r * SG_BAND + lane];
    return v == 0 ? -1 : v;          /* -1 stands for the reference's minus infinity: never equal to a live value +-1 */
}

static int sg_base1(const uint8_t *seq1, long y)   /* character of row y (1-based), 0xF0 outside */
{
    return (y >= 1 && y <= SG_LEN) ? seq1[y - 1] : 0xF0;
}

static int sg_base2(const uint8_t *seq2, long x)
{
    return (x >= 1 && x <= SG_LEN) ? seq2[x - 1] : 0xF0;
}

/* returns 0; traceback[2k], traceback[2k+1] = (i, j) of step k for k < min(*len, cap) */
int sg_oracle_xdrop(const uint8_t *seq1, const uint8_t *seq2, int32_t *score, int32_t *traceback, size_t cap,
                    size_t *len, int *touched_oob)
{
    sg_table t;
    int cur[SG_BAND], hor[SG_BAND], ver[SG_BAND], dia[SG_BAND];
    long pos_y = 0, pos_x = 31;      /* reference coordinates: pos_x counts 31 leading pads */
    int best = SG_X, best_round = 0, round;
    t.cell = (uint16_t *)calloc((size_t)SG_MAX_ROUND * SG_BAND, sizeof(uint16_t));
    t.top_y = (int32_t *)calloc((size_t)SG_MAX_ROUND, sizeof(int32_t));
    if (!t.cell || !t.top_y) { free(t.cell); free(t.top_y); return -1; }
    memset(cur, 0, sizeof cur); memset(hor, 0, sizeof hor); memset(ver, 0, sizeof ver); memset(dia, 0, sizeof dia);
    cur[31] = SG_X;
    t.cell[31] = SG_X;
    t.top_y[0] = 0;
    if (touched_oob) *touched_oob = 0;

    for (round = 1; round < SG_MAX_ROUND; ++round) {
        if (cur[0] < cur[31]) {                       /* step right */
            for (int k = 0; k < SG_BAND; ++k) dia[k] = ver[k];
            for (int k = 0; k < SG_BAND; ++k) hor[k] = cur[k];
            for (int k = 0; k < SG_BAND - 1; ++k) ver[k] = cur[k + 1];
            ver[31] = 0;
            if (++pos_x > 32 + SG_LEN + 31) break;
        } else {                                      /* step down */
            for (int k = 0; k < SG_BAND; ++k) dia[k] = hor[k];
            for (int k = 0; k < SG_BAND; ++k) ver[k] = cur[k];
            for (int k = SG_BAND - 1; k > 0; --k) hor[k] = cur[k - 1];
            hor[0] = 0;
            if (++pos_y > 1 + SG_LEN) break;
        }
        if (touched_oob && (pos_y + 31 >= 1 + SG_LEN + 31 || pos_x >= 32 + SG_LEN + 31)) *touched_oob = 1;
        int round_best = 0;
        for (int k = 0; k < SG_BAND; ++k) {
            const long y = pos_y + 31 - k, x = pos_x - 62 + k;
            const int c1 = sg_base1(seq1, y), c2 = sg_base2(seq2, x);
            const int s = (c1 < 4 && c2 < 4 && c1 == c2) ? 1 : -1;
            int v = 0;
            if (dia[k] != 0 && dia[k] + s > v) v = dia[k] + s;
            if (hor[k] != 0 && hor[k] - 1 > v) v = hor[k] - 1;
            if (ver[k] != 0 && ver[k] - 1 > v) v = ver[k] - 1;
            cur[k] = v;
            if (v > round_best) round_best = v;
        }
        if (round_best > best) { best = round_best; best_round = round; }
        for (int k = 0; k < SG_BAND; ++k) {
            if (cur[k] < best - SG_X) cur[k] = 0;
            t.cell[(size_t)round * SG_BAND + k] = (uint16_t)cur[k];
        }
        t.top_y[round] = (int32_t)pos_y;
        if (round_best == 0) { ++round; break; }
    }
    t.rounds = round < SG_MAX_ROUND ? round : SG_MAX_ROUND;

    /* best cell: first lane from 31 downwards of the best round that holds the best value */
    long y = t.top_y[best_round], x = best_round - y;
    while (sg_get(&t, y, x) != best) { ++y; --x; }
    size_t n = 0;
    /* walk back to (0,0), collecting positions; emitted in ascending order afterwards */
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(SG_MAX_ROUND + 1));
    if (!tmp) { free(t.cell); free(t.top_y); return -1; }
    tmp[0] = (int32_t)y; tmp[1] = (int32_t)x; n = 1;
    while (y || x) {
        const int v = sg_get(&t, y, x);
        /* the reference indexes its 4x4 matrix with the raw bytes here (source.cpp:1961): undefined for a byte >= 4.  Such
         * bytes are outside its domain; this restatement gives them the sweep's meaning (mismatch against everything). */
        if (y && x && v == sg_get(&t, y - 1, x - 1) + ((seq1[y - 1] < 4 && seq1[y - 1] == seq2[x - 1]) ? 1 : -1) && sg_get(&t, y - 1, x - 1) > 0) { --y; --x; }
        else if (y && v == sg_get(&t, y - 1, x) - 1 && sg_get(&t, y - 1, x) > 0) { --y; }
        else if (x && v == sg_get(&t, y, x - 1) - 1 && sg_get(&t, y, x - 1) > 0) { --x; }
        else { free(tmp); free(t.cell); free(t.top_y); return -2; }   /* the reference asserts here */
        tmp[2 * n] = (int32_t)y; tmp[2 * n + 1] = (int32_t)x; ++n;
    }
    *score = best - SG_X;
    *len = n;
    for (size_t k = 0; k < n && k < cap; ++k) {
        traceback[2 * k] = tmp[2 * (n - 1 - k)];
        traceback[2 * k + 1] = tmp[2 * (n - 1 - k) + 1];
    }
    free(tmp); free(t.cell); free(t.top_y);
    return 0;
}

/* The invariant behind the GPU sweeps' "calm windows" (smith-waterman-simd_amd/csrc/sg_kernels.hip), checked on the reference's
 * own recurrence (:1895-1946) -- test infrastructure like everything here.  At the start of every window of `window` rounds
 * (rounds 1, 1 + window, ...) the band is called calm when none of its 32 cells is dropped and the lowest one lies at least
 * `margin` above the X-drop threshold best - 70.  The claim: inside a calm window the X-drop rule drops nothing (so the GPU may
 * skip the test there), for margin >= window + window / 2 + 1.  counts[0] = windows, counts[1] = calm windows, counts[2] =
 * cells the rule dropped inside calm windows (the claim: 0).  Returns 0. */
int sg_oracle_calm_windows(const uint8_t *seq1, const uint8_t *seq2, int window, int margin, long counts[3])
{
    int cur[SG_BAND], hor[SG_BAND], ver[SG_BAND], dia[SG_BAND];
    long pos_y = 0, pos_x = 31;
    int best = SG_X, round, calm = 0;
    memset(cur, 0, sizeof cur); memset(hor, 0, sizeof hor); memset(ver, 0, sizeof ver); memset(dia, 0, sizeof dia);
    cur[31] = SG_X;
    counts[0] = counts[1] = counts[2] = 0;
    for (round = 1; round < SG_MAX_ROUND; ++round) {
        if ((round - 1) % window == 0) {
            int low = cur[0];
            for (int k = 1; k < SG_BAND; ++k) if (cur[k] < low) low = cur[k];
            calm = low != 0 && low - (best - SG_X) >= margin;
            ++counts[0];
            counts[1] += calm;
        }
        if (cur[0] < cur[31]) {
            for (int k = 0; k < SG_BAND; ++k) dia[k] = ver[k];
            for (int k = 0; k < SG_BAND; ++k) hor[k] = cur[k];
            for (int k = 0; k < SG_BAND - 1; ++k) ver[k] = cur[k + 1];
            ver[31] = 0;
            if (++pos_x > 32 + SG_LEN + 31) break;
        } else {
            for (int k = 0; k < SG_BAND; ++k) dia[k] = hor[k];
            for (int k = 0; k < SG_BAND; ++k) ver[k] = cur[k];
            for (int k = SG_BAND - 1; k > 0; --k) hor[k] = cur[k - 1];
            hor[0] = 0;
            if (++pos_y > 1 + SG_LEN) break;
        }
        int round_best = 0;
        for (int k = 0; k < SG_BAND; ++k) {
            const long y = pos_y + 31 - k, x = pos_x - 62 + k;
            const int c1 = sg_base1(seq1, y), c2 = sg_base2(seq2, x);
            const int s = (c1 < 4 && c2 < 4 && c1 == c2) ? 1 : -1;
            int v = 0;
            if (dia[k] != 0 && dia[k] + s > v) v = dia[k] + s;
            if (hor[k] != 0 && hor[k] - 1 > v) v = hor[k] - 1;
            if (ver[k] != 0 && ver[k] - 1 > v) v = ver[k] - 1;
            cur[k] = v;
            if (v > round_best) round_best = v;
        }
        if (round_best > best) best = round_best;
        for (int k = 0; k < SG_BAND; ++k) {
            if (cur[k] < best - SG_X) {
                if (calm) ++counts[2];                 /* (a cell that is 0 already counts too: nothing may be dropped here) */
                cur[k] = 0;
            }
        }
        if (round_best == 0) break;
    }
    return 0;
}

