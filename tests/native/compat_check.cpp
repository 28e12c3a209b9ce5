// Exercises include/swmi_compat.hpp the way a maintainer of the reference would: the per-pair overload with the
// reference's exact argument list, and the PairQueue that keeps the loop shape of SpeedTest (source.cpp:3074-3082).
// Reads pairs from a raw file (n x 2 x 128 bytes), prints one score per line for three routes: per-pair / queued, and the
// whole-array overload -- on one GPU context, then again with the GPU bound twice (the multi-GPU split, swmi_multi.cpp).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "swmi_compat.hpp"

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const char *path = argv[1];
    const int match = atoi(argv[2]), mismatch = atoi(argv[3]), gap = atoi(argv[4]);
    FILE *f = fopen(path, "rb");
    if (!f) return 3;
    std::vector<std::array<uint8_t, 128>> a, b;
    std::array<uint8_t, 128> x, y;
    while (fread(x.data(), 1, 128, f) == 128 && fread(y.data(), 1, 128, f) == 128) { a.push_back(x); b.push_back(y); }
    fclose(f);
    std::array<int8_t, 16> sm;
    for (int i = 0; i < 16; ++i) sm[i] = int8_t(i % 5 == 0 ? match : mismatch);
    if (swmi_init(-1) != SWMI_OK) { fprintf(stderr, "%s\n", swmi_last_error()); return 4; }
    swmi::PairQueue q(a.size(), sm, int8_t(gap));
    for (size_t k = 0; k < a.size(); ++k) q.submit(a[k], b[k]);
    const std::vector<int32_t> queued = q.scores();
    const std::vector<int32_t> whole = swmi::SmithWaterman_mi355x_batch(a, b, sm, int8_t(gap));
    for (size_t k = 0; k < a.size(); ++k) {
        const int direct = k < 8 ? SmithWaterman_mi355x(a[k], b[k], sm, int8_t(gap)) : queued[k];
        printf("%d %d %d\n", direct, queued[k], whole[k]);
    }
    try {
        SmithWaterman_mi355x(a[0], b[0], sm, int8_t(-1));      // outside the domain: the overload throws
        return 5;
    } catch (const std::runtime_error &) {
    }
    swmi_shutdown();
    // the same array call with two GPU contexts in this one process (one GPU bound twice stands in for two GPUs)
    const int twice[2] = {0, 0};
    if (swmi_init_devices(twice, 2) != 2) { fprintf(stderr, "%s\n", swmi_last_error()); return 6; }
    const std::vector<int32_t> split = swmi::SmithWaterman_mi355x_batch(a, b, sm, int8_t(gap));
    if (split != whole) return 7;
    swmi_shutdown();
    return 0;
}
