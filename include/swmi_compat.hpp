// swmi_compat.hpp -- header-only C++ mirror of the reference's call signature over the C ABI (swmi.h).
//
// The reference's boundary is the free function (source.cpp:462-466, :758-762, :953-957)
//     int SmithWaterman_simdN(const std::array<uint8_t,128>&, const std::array<uint8_t,128>&,
//                             const std::array<int8_t,16>&, const int8_t);
// A maintainer who wants the reference's own drivers (SpeedTest source.cpp:3032-3147,
// TestSimdSmithWaterman :2943-2982) to run on the GPU includes this header and calls
// SmithWaterman_mi355x(...) where they called SmithWaterman_simd4(...) -- or, to keep the 1M-call loop
// shape AND get batch throughput, submits through swmi::PairQueue (same arguments per call).
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "swmi.h"

// Same arguments, same return value as SmithWaterman_simd4 (source.cpp:462-466). One synchronous launch per call.
inline int SmithWaterman_mi355x(const std::array<uint8_t, 128> &seq1, const std::array<uint8_t, 128> &seq2,
                                const std::array<int8_t, 16> &score_matrix, const int8_t gap_penalty)
{
    const int r = swmi_score_pair(seq1.data(), seq2.data(), score_matrix.data(), gap_penalty);
    if (r < 0) throw std::runtime_error(std::string("swmi_score_pair: ") + swmi_last_error());
    return r;
}

// Same arguments, same return value as SemiGlobal_AdaptiveBanded_XDrop_111_32_70 and its _simd / _simd_mark2..4 variants
// (source.cpp:1836-1838, :1978, call sites TestSemiGlobal :2774-2778, SpeedtestSemiGlobal :2818-2856): (score, traceback from
// (0, 0) to the best cell).  One synchronous call per alignment -- correct, but ~11 ms of latency each: use the batch below.
inline std::pair<int, std::vector<std::pair<int, int>>> SemiGlobal_AdaptiveBanded_XDrop_mi355x(const std::array<uint8_t, 16384> &seq1,
                                                                                              const std::array<uint8_t, 16384> &seq2);

namespace swmi {

// One alignment's moves (swmi_semiglobal_xdrop_moves) -> the reference's traceback vector (source.cpp:1962-1975).
inline std::vector<std::pair<int, int>> expand_moves(const uint64_t *moves, uint32_t length)
{
    static_assert(sizeof(std::pair<int, int>) == 2 * sizeof(int32_t), "std::pair<int,int> must be two packed ints");
    std::vector<std::pair<int, int>> tb(length);
    if (swmi_semiglobal_expand_moves(moves, length, reinterpret_cast<int32_t *>(tb.data()), length) != SWMI_OK)
        throw std::runtime_error(std::string("swmi_semiglobal_expand_moves: ") + swmi_last_error());
    return tb;
}

// The reference's SpeedtestSemiGlobal loop (source.cpp:2818-2856) over arrays of pairs: result[k] ==
// SemiGlobal_AdaptiveBanded_XDrop_111_32_70(seq1s[k], seq2s[k]).  The GPU returns 2 bits per traceback step (8 KB per
// alignment over PCIe instead of the 262 KB its positions take); the positions are rebuilt here, on `threads` host threads
// (0 = as many as the machine reports, at most 64), one slice of 65536 alignments while the GPU works on the next.
inline std::vector<std::pair<int, std::vector<std::pair<int, int>>>> SemiGlobal_mi355x_batch(
    const std::vector<std::array<uint8_t, 16384>> &seq1s, const std::vector<std::array<uint8_t, 16384>> &seq2s, unsigned threads = 0)
{
    static_assert(sizeof(std::array<uint8_t, 16384>) == 16384, "std::array<uint8_t,16384> must be 16384 contiguous bytes");
    if (seq1s.size() != seq2s.size()) throw std::invalid_argument("SemiGlobal_mi355x_batch: seq1s and seq2s differ in length");
    const size_t n = seq1s.size();
    std::vector<std::pair<int, std::vector<std::pair<int, int>>>> out(n);
    if (n == 0) return out;
    if (threads == 0) threads = std::thread::hardware_concurrency();
    threads = threads < 1 ? 1 : threads > 64 ? 64 : threads;
    constexpr size_t kSlice = 65536;                   // two of the library's internal chunks: its copies and kernels overlap inside a call
    std::vector<int32_t> scores(n);
    std::vector<uint32_t> lengths(n);
    std::vector<uint64_t> moves(n * size_t(SWMI_SG_MOVE_WORDS));
    std::vector<std::thread> pool;                     // the expanders of the slice before the one the GPU works on
    std::string failed;
    auto join_all = [&] {
        for (auto &th : pool) th.join();
        pool.clear();
    };
    for (size_t off = 0; off < n && failed.empty(); off += kSlice) {
        const size_t m = n - off < kSlice ? n - off : kSlice;
        if (swmi_semiglobal_xdrop_moves(seq1s[off].data(), seq2s[off].data(), m, scores.data() + off,
                                        moves.data() + off * size_t(SWMI_SG_MOVE_WORDS), lengths.data() + off) != SWMI_OK)
            failed = swmi_last_error();
        join_all();
        if (!failed.empty()) break;
        const unsigned use = threads > m ? unsigned(m) : threads;
        for (unsigned t = 0; t < use; ++t)
            pool.emplace_back([&, off, m, t, use] {
                for (size_t k = off + m * t / use; k < off + m * (t + 1) / use; ++k)
                    out[k] = {scores[k], expand_moves(moves.data() + k * size_t(SWMI_SG_MOVE_WORDS), lengths[k])};
            });
    }
    join_all();
    if (!failed.empty()) throw std::runtime_error("swmi_semiglobal_xdrop_moves: " + failed);
    return out;
}

// The reference's 1M-call loop (source.cpp:3074-3082) over arrays of pairs, on every GPU the library is bound to:
// scores[k] == SmithWaterman(seq1s[k], seq2s[k], score_matrix, gap_penalty).  std::array<uint8_t,128> has no padding, so a
// vector of them IS the concatenated layout the C ABI takes.  With swmi_init(device) it runs on that one GPU, with
// swmi_init_all(G) the batch is cut into G contiguous shards, one host thread and stream set per GPU (swmi_score_batch_multi).
inline std::vector<int32_t> SmithWaterman_mi355x_batch(const std::vector<std::array<uint8_t, 128>> &seq1s,
                                                       const std::vector<std::array<uint8_t, 128>> &seq2s,
                                                       const std::array<int8_t, 16> &score_matrix, const int8_t gap_penalty)
{
    static_assert(sizeof(std::array<uint8_t, 128>) == 128, "std::array<uint8_t,128> must be 128 contiguous bytes");
    if (seq1s.size() != seq2s.size()) throw std::invalid_argument("SmithWaterman_mi355x_batch: seq1s and seq2s differ in length");
    std::vector<int32_t> scores(seq1s.size());
    const uint8_t *a = seq1s.empty() ? nullptr : seq1s[0].data(), *b = seq2s.empty() ? nullptr : seq2s[0].data();
    const int rc = swmi_num_gpus() > 1
                       ? swmi_score_batch_multi(a, b, seq1s.size(), score_matrix.data(), gap_penalty, scores.data())
                       : swmi_score_batch(a, b, seq1s.size(), score_matrix.data(), gap_penalty, scores.data());
    if (rc != SWMI_OK) throw std::runtime_error(std::string("SmithWaterman_mi355x_batch: ") + swmi_last_error());
    return scores;
}

// Batches per-pair calls: submit() has the reference's argument list and returns a ticket; scores() drains.
class PairQueue {
public:
    PairQueue(size_t max_pairs, const std::array<int8_t, 16> &score_matrix, int8_t gap_penalty)
    {
        if (swmi_queue_create(max_pairs, score_matrix.data(), gap_penalty, &q_) != SWMI_OK)
            throw std::runtime_error(std::string("swmi_queue_create: ") + swmi_last_error());
    }
    ~PairQueue() { swmi_queue_destroy(q_); }
    PairQueue(const PairQueue &) = delete;
    PairQueue &operator=(const PairQueue &) = delete;

    long long submit(const std::array<uint8_t, 128> &seq1, const std::array<uint8_t, 128> &seq2)
    {
        const long long t = swmi_queue_submit(q_, seq1.data(), seq2.data());
        if (t < 0) throw std::runtime_error(std::string("swmi_queue_submit: ") + swmi_last_error());
        return t;
    }
    // scores()[ticket] == SmithWaterman(seq1, seq2, score_matrix, gap_penalty) of that submit()
    std::vector<int32_t> scores()
    {
        const int32_t *p = nullptr;
        size_t n = 0;
        if (swmi_queue_wait(q_, &p, &n) != SWMI_OK) throw std::runtime_error(std::string("swmi_queue_wait: ") + swmi_last_error());
        return std::vector<int32_t>(p, p + n);
    }
    void reset() { swmi_queue_reset(q_); }

private:
    swmi_queue *q_ = nullptr;
};

}  // namespace swmi

inline std::pair<int, std::vector<std::pair<int, int>>> SemiGlobal_AdaptiveBanded_XDrop_mi355x(const std::array<uint8_t, 16384> &seq1,
                                                                                              const std::array<uint8_t, 16384> &seq2)
{
    int32_t score = 0;
    uint32_t length = 0;
    std::vector<uint64_t> moves(SWMI_SG_MOVE_WORDS);
    if (swmi_semiglobal_xdrop_moves(seq1.data(), seq2.data(), 1, &score, moves.data(), &length) != SWMI_OK)
        throw std::runtime_error(std::string("swmi_semiglobal_xdrop_moves: ") + swmi_last_error());
    return {score, swmi::expand_moves(moves.data(), length)};
}
