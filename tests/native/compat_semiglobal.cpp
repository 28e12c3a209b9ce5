// Exercises the semi-global part of include/swmi_compat.hpp the way a maintainer of the reference would replace the calls of
// TestSemiGlobal / SpeedtestSemiGlobal (source.cpp:2774-2778, :2818-2856): the per-alignment overload with the reference's
// argument list and the batch overload, whose tracebacks are rebuilt on host threads from the 2-bit moves the GPU returns.
// Reads alignments from a raw file (n x 2 x 16384 bytes); prints per alignment: score, traceback length, end cell, and a
// checksum of the whole (i, j) list, for the batch route; exits non-zero if the single-call route disagrees on alignment 0.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "swmi_compat.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    std::vector<std::array<uint8_t, 16384>> a, b;
    auto x = std::make_unique<std::array<uint8_t, 16384>>(), y = std::make_unique<std::array<uint8_t, 16384>>();
    while (fread(x->data(), 1, 16384, f) == 16384 && fread(y->data(), 1, 16384, f) == 16384) { a.push_back(*x); b.push_back(*y); }
    fclose(f);
    if (swmi_init(-1) != SWMI_OK) { fprintf(stderr, "%s\n", swmi_last_error()); return 4; }
    const auto all = swmi::SemiGlobal_mi355x_batch(a, b, 3);
    for (const auto &r : all) {
        unsigned long long sum = 0;
        for (size_t k = 0; k < r.second.size(); ++k) sum = sum * 1000003ull + (unsigned long long)(r.second[k].first * 32771 + r.second[k].second);
        printf("%d %zu %d %d %llu\n", r.first, r.second.size(), r.second.back().first, r.second.back().second, sum);
    }
    const auto one = SemiGlobal_AdaptiveBanded_XDrop_mi355x(a[0], b[0]);
    if (one != all[0]) return 5;
    swmi_shutdown();
    return 0;
}
