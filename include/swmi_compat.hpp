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

namespace swmi {

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
