// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Builds the REAL reference (eukaryo/smith-waterman-simd, source.cpp) into
// oracle/_ref/libswref.so so that the C restatement in sw_oracle.c and the golden
// fixtures can be pinned to it, and so that bench.py can time the reference's own
// simd4 path on the GPU box's host cores (cpu_baseline.kind = "reference").
//
// No reference source is copied: the translation unit below #includes the file
// where it lies (SWREF_SOURCE is passed by oracle/Makefile, default
// /root/reference/source.cpp) with its main() renamed, and exports thin C wrappers
// around the reference's own functions.
#include <cstddef>
#include <cstdint>

#define main swref_reference_main
#include SWREF_SOURCE
#undef main

namespace {
using Seq = std::array<uint8_t, 128>;
using Mat = std::array<int8_t, 16>;
inline const Seq &seq(const uint8_t *p) { return *reinterpret_cast<const Seq *>(p); }
inline const Mat &mat(const int8_t *p) { return *reinterpret_cast<const Mat *>(p); }
typedef int (*sw_fn)(const Seq &, const Seq &, const Mat &, const int8_t);
sw_fn pick(int variant)
{
    switch (variant) {
    case 0: return SmithWaterman;          // source.cpp:35
    case 1: return SmithWaterman_simd;     // source.cpp:62
    case 2: return SmithWaterman_simd2;
    case 3: return SmithWaterman_simd3;
    case 4: return SmithWaterman_simd4;    // source.cpp:462
    case 5: return SmithWaterman_simd5;
    case 6: return SmithWaterman_simd6;
    case 7: return SmithWaterman_simd7;    // source.cpp:758
    case 8: return SmithWaterman_simd8;
    case 9: return SmithWaterman_simd9;    // source.cpp:953
    default: return nullptr;
    }
}
}  // namespace

extern "C" {

// variant: 0 = scalar SmithWaterman, N = SmithWaterman_simdN (1..9); returns -1 for an unknown variant
int swref_score(int variant, const uint8_t *seq1, const uint8_t *seq2, const int8_t *sm, int gap)
{
    sw_fn f = pick(variant);
    if (!f) return -1;
    return f(seq(seq1), seq(seq2), mat(sm), (int8_t)gap);
}

int swref_batch(int variant, const uint8_t *seq1s, const uint8_t *seq2s, size_t n,
                const int8_t *sm, int gap, int32_t *scores)
{
    sw_fn f = pick(variant);
    if (!f) return -1;
    for (size_t k = 0; k < n; ++k)
        scores[k] = f(seq(seq1s + 128 * k), seq(seq2s + 128 * k), mat(sm), (int8_t)gap);
    return 0;
}

// The reference harness shape (source.cpp:3074-3082): ONE pair scored `iters` times into a volatile sink.
long long swref_repeat(int variant, const uint8_t *seq1, const uint8_t *seq2, const int8_t *sm, int gap, int iters)
{
    sw_fn f = pick(variant);
    if (!f) return -1;
    long long sum = 0;
    for (int it = 0; it < iters; ++it) {
        volatile int score = f(seq(seq1), seq(seq2), mat(sm), (int8_t)gap);
        sum += score;
    }
    return sum;
}

// (1,1,1) siblings and the 2-bit unpack, for the "next" rows N1-N3 of SURVEY.md section 8(f)
int swref_score_111(const uint8_t *seq1, const uint8_t *seq2) { return SmithWaterman_111(seq(seq1), seq(seq2)); }
int swref_score_8bit111simd(const uint8_t *seq1, const uint8_t *seq2) { return SmithWaterman_8bit111simd(seq(seq1), seq(seq2)); }
void swref_111x32(int mark, const uint8_t *seq1x32, const uint8_t *seq2, int32_t *dest32)
{
    const auto &a = *reinterpret_cast<const std::array<uint8_t, 128 * 32> *>(seq1x32);
    std::array<int, 32> d;
    if (mark == 1) SmithWaterman_8b111x32mark1(a, seq(seq2), d);
    else if (mark == 2) SmithWaterman_8b111x32mark2(a, seq(seq2), d);
    else SmithWaterman_8b111x32mark3(a, seq(seq2), d);
    for (int i = 0; i < 32; ++i) dest32[i] = d[i];
}
void swref_unpack(const uint8_t *src32, uint8_t *dst128)
{
    unpack(*reinterpret_cast<const std::array<uint8_t, 32> *>(src32), *reinterpret_cast<std::array<uint8_t, 128> *>(dst128));
}

// Semi-global adaptive-band X-drop family (source.cpp:1836-2725), SURVEY 8f row N4.
// variant: 0 = scalar (:1836), 1 = _simd (:1978), 2 = _simd_mark2 (:2167), 3 = _simd_mark3 (:2355), 4 = _simd_mark4 (:2543).
// traceback receives up to `cap` (i, j) pairs as 2 x int32 each; *len = number of pairs the reference returned.
int swref_semiglobal(int variant, const uint8_t *seq1, const uint8_t *seq2, int32_t *score, int32_t *traceback,
                     size_t cap, size_t *len)
{
    using Long = std::array<uint8_t, 16384>;
    const Long &a = *reinterpret_cast<const Long *>(seq1);
    const Long &b = *reinterpret_cast<const Long *>(seq2);
    std::pair<int, std::vector<std::pair<int, int>>> r;
    switch (variant) {
    case 0: r = SemiGlobal_AdaptiveBanded_XDrop_111_32_70(a, b); break;
    case 1: r = SemiGlobal_AdaptiveBanded_XDrop_111_32_70_simd(a, b); break;
    case 2: r = SemiGlobal_AdaptiveBanded_XDrop_111_32_70_simd_mark2(a, b); break;
    case 3: r = SemiGlobal_AdaptiveBanded_XDrop_111_32_70_simd_mark3(a, b); break;
    case 4: r = SemiGlobal_AdaptiveBanded_XDrop_111_32_70_simd_mark4(a, b); break;
    default: return -1;
    }
    *score = r.first;
    *len = r.second.size();
    for (size_t k = 0; k < r.second.size() && k < cap; ++k) {
        traceback[2 * k] = r.second[k].first;
        traceback[2 * k + 1] = r.second[k].second;
    }
    return 0;
}

// The reference's test inputs for that family (TestSemiGlobal, source.cpp:2734-2771): a random 16384-mer and a copy
// with ~10 % substitutions, ~10 % insertions, ~10 % deletions; libstdc++ draw, `n` consecutive pairs of the stream.
void swref_semiglobal_stream(uint64_t seed, size_t n, uint8_t *seq1s, uint8_t *seq2s)
{
    std::mt19937_64 rnd(seed);
    std::uniform_int_distribution<int> dna(0, 3);
    std::uniform_int_distribution<int> dice(0, 99);
    for (size_t k = 0; k < n; ++k) {
        uint8_t *a = seq1s + 16384 * k, *b = seq2s + 16384 * k;
        for (int i = 0; i < 16384; ++i) a[i] = (uint8_t)dna(rnd);
        for (int i = 0, j = 0; i < 16384;) {
            if (j == 16384) b[i++] = (uint8_t)dna(rnd);
            else {
                const int p = dice(rnd);
                if (p < 10) { b[i++] = (uint8_t)dna(rnd); ++j; }
                else if (p < 20) { b[i++] = (uint8_t)dna(rnd); }
                else if (p < 30) { ++j; }
                else { b[i++] = a[j++]; }
            }
        }
    }
}

// Input streams exactly as the reference drivers draw them (libstdc++ here):
//   interleaved a[i], b[i] from mt19937_64(seed) + uniform_int_distribution<int>(0,3)
//   -- SpeedTest source.cpp:3033-3040 (one pair), TestSimdSmithWaterman source.cpp:2944-2953 (a stream of pairs).
void swref_harness_stream(uint64_t seed, size_t n_pairs, uint8_t *seq1s, uint8_t *seq2s)
{
    std::mt19937_64 rnd(seed);
    std::uniform_int_distribution<int> dna(0, 3);
    for (size_t k = 0; k < n_pairs; ++k)
        for (int i = 0; i < 128; ++i) {
            seq1s[128 * k + i] = (uint8_t)dna(rnd);
            seq2s[128 * k + i] = (uint8_t)dna(rnd);
        }
}

}  // extern "C"
