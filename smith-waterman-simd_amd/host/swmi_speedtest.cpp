#include <algorithm>
// swmi_speedtest.cpp -- the reference's timing driver shape (SpeedTest, source.cpp:3032-3147; speedtest111x32,
// :3189-3273) over libswmi: same parameters, same "<name> version: <ms> ms / 1M" lines, GPU behind the call.
//
// Three ways of reaching the kernel are timed, all through the C ABI:
//   per-pair   : SmithWaterman_mi355x(a, b, sm, gap) in a loop  -- the literal drop-in; launch-latency bound
//   queued     : the same 1,000,000 calls submitted through swmi::PairQueue (host batches, GPU scores)
//   batch      : one swmi_score_batch() over 1M DISTINCT pairs (counter-based generator), host buffers
//   device     : swmi_time_batch_device() on inputs resident in HBM (kernel only)
//   N-gpu      : the same loop pointed at every GPU of the node from this ONE process (swmi_init_all): the host batch
//                split by swmi_score_batch_multi, and resident shards scored + gathered by swmi_sharded_*
//                (SWMI_SPEEDTEST_DEVICES=0,0 binds a list instead -- two contexts on one GPU rehearse the path)
// No CPU scoring happens in this program; the CPU baseline is timed by bench.py's cpu_baseline leg.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/swmi_compat.hpp"

namespace {
double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
void die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, swmi_last_error());
    exit(1);
}
}  // namespace

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000;   // calls (reference: 1,000,000)
    const size_t n_sync = argc > 2 ? strtoull(argv[2], nullptr, 10) : 20000; // per-pair synchronous calls to time
    // argv[3]: alignments in the SpeedtestSemiGlobal section (reference: 10,000)
    if (swmi_init(-1) != SWMI_OK) die("swmi_init");
    swmi_device_info info;
    swmi_get_device_info(&info);
    printf("device %d: %s (%s), %d CUs\n", info.device, info.name, info.arch, info.compute_units);

    // the reference harness scores ONE pair 1M times (source.cpp:3033-3040); pair 0 of the generator stands in
    // for its mt19937_64(10000) draw, which is implementation-defined across standard libraries
    std::array<uint8_t, 128> a, b;
    swmi_generate_pairs_host(a.data(), b.data(), 1, 10000, 0);
    struct Params { const char *tag; int match, mismatch, gap; };
    const Params sets[2] = {{"SpeedTest (10,-30,15)", 10, -30, 15}, {"speedtest111x32 (1,-1,1)", 1, -1, 1}};
    for (const Params &ps : sets) {
        std::array<int8_t, 16> sm;
        for (int i = 0; i < 16; ++i) sm[i] = int8_t(i % 5 == 0 ? ps.match : ps.mismatch);   // source.cpp:3041-3045
        const int8_t gap = int8_t(ps.gap);
        printf("== %s\n", ps.tag);
        {
            const double t0 = now_ms();
            long long sink = 0;
            for (size_t it = 0; it < n_sync; ++it) {
                volatile int score = SmithWaterman_mi355x(a, b, sm, gap);
                sink += score;
            }
            const double ms = now_ms() - t0;
            printf("mi355x per-pair version: %.0f ms / %zuK  (score %lld; %.1f us per call, launch-latency bound)\n", ms,
                   n_sync / 1000, sink / (long long)n_sync, 1000.0 * ms / double(n_sync));
        }
        {
            swmi::PairQueue q(n, sm, gap);
            const double t0 = now_ms();
            for (size_t it = 0; it < n; ++it) q.submit(a, b);
            const std::vector<int32_t> scores = q.scores();
            const double ms = now_ms() - t0;
            long long sum = 0;
            for (int32_t s : scores) sum += s;
            printf("mi355x queued version: %.0f ms / %.0fM  (score %lld)\n", ms, n / 1e6, sum / (long long)n);
        }
        {
            std::vector<uint8_t> s1(n * 128), s2(n * 128);
            std::vector<int32_t> out(n);
            swmi_generate_pairs_host(s1.data(), s2.data(), n, 10000, 0);
            if (swmi_score_batch(s1.data(), s2.data(), n, sm.data(), gap, out.data()) != SWMI_OK) die("swmi_score_batch");
            double ms = 1e30;                   // best of three: the link's clocks take a call or two to come up after the per-pair phase
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now_ms();
                if (swmi_score_batch(s1.data(), s2.data(), n, sm.data(), gap, out.data()) != SWMI_OK) die("swmi_score_batch");
                ms = std::min(ms, now_ms() - t0);
            }
            long long sum = 0;
            for (int32_t s : out) sum += s;
            printf("mi355x batch version: %.1f ms / %.0fM distinct pairs incl. PCIe  (%.1f M alignments/s, checksum %lld)\n", ms,
                   n / 1e6, n / ms / 1e3, sum);
        }
        {   // the same pairs in the reference's 2-bit wire format (unpack(), source.cpp:1580-1583): a quarter of the PCIe bytes
            std::vector<uint8_t> s1(n * 128), s2(n * 128), p1(n * 32), p2(n * 32);
            std::vector<int32_t> out(n);
            swmi_generate_pairs_host(s1.data(), s2.data(), n, 10000, 0);
            for (size_t i = 0; i < n * 32; ++i) {
                p1[i] = uint8_t(s1[4 * i] | (s1[4 * i + 1] << 2) | (s1[4 * i + 2] << 4) | (s1[4 * i + 3] << 6));
                p2[i] = uint8_t(s2[4 * i] | (s2[4 * i + 1] << 2) | (s2[4 * i + 2] << 4) | (s2[4 * i + 3] << 6));
            }
            if (swmi_score_batch_packed(p1.data(), p2.data(), n, sm.data(), gap, out.data()) != SWMI_OK) die("swmi_score_batch_packed");
            double ms = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now_ms();
                if (swmi_score_batch_packed(p1.data(), p2.data(), n, sm.data(), gap, out.data()) != SWMI_OK) die("swmi_score_batch_packed");
                ms = std::min(ms, now_ms() - t0);
            }
            long long sum = 0;
            for (int32_t s : out) sum += s;
            printf("mi355x packed batch version: %.1f ms / %.0fM distinct pairs incl. PCIe  (%.1f M alignments/s, checksum %lld)\n", ms,
                   n / 1e6, n / ms / 1e3, sum);
        }
        {
            void *d1 = nullptr, *d2 = nullptr, *ds = nullptr;
            if (hipMalloc(&d1, n * 128) != hipSuccess || hipMalloc(&d2, n * 128) != hipSuccess ||
                hipMalloc(&ds, n * 4) != hipSuccess) {
                fprintf(stderr, "hipMalloc failed\n");
                return 1;
            }
            if (swmi_generate_pairs_device(d1, d2, n, 10000, 0, nullptr) != SWMI_OK) die("swmi_generate_pairs_device");
            float ms = 0.f;
            if (swmi_time_batch_device(d1, d2, n, sm.data(), gap, ds, nullptr, 3, &ms) != SWMI_OK) die("warmup");
            if (swmi_time_batch_device(d1, d2, n, sm.data(), gap, ds, nullptr, 20, &ms) != SWMI_OK) die("swmi_time_batch_device");
            printf("mi355x device version: %.3f ms / %.0fM  (%.1f M alignments/s, %.2f TCUPS, inputs resident in HBM)\n", ms,
                   n / 1e6, n / ms / 1e3, n * 16384.0 / ms / 1e9);
            (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(ds);
        }
    }
    // == SpeedtestSemiGlobal (source.cpp:2804-2860): ONE pair of 16384-mers, 5 % substitutions (:2808-2812), aligned 10K
    // times with score + traceback.  Here the 10K calls are one batch of 10K copies of that pair.
    {
        const size_t n_sg = argc > 3 ? strtoull(argv[3], nullptr, 10) : 10000;
        constexpr size_t kLen = SWMI_SG_LEN, kCap = SWMI_SG_MAX_TRACEBACK;
        std::vector<uint8_t> one_a(kLen), one_b(kLen);
        {   // 16384 bases = 128 consecutive 128-mers of the generator; every 20th position of b redrawn (dice(rnd) == 0)
            std::vector<uint8_t> g1(kLen), g2(kLen), g3(kLen), g4(kLen);
            swmi_generate_pairs_host(g1.data(), g2.data(), kLen / 128, 10000, 0);
            swmi_generate_pairs_host(g3.data(), g4.data(), kLen / 128, 10001, 0);
            for (size_t i = 0; i < kLen; ++i) {
                one_a[i] = g1[i];
                one_b[i] = ((g3[i] * 4u + g4[i]) * 7u + i) % 20u == 0 ? g2[i] : g1[i];
            }
        }
        printf("== SpeedtestSemiGlobal (16384 x 16384, band 32, X-drop 70, score + traceback)\n");
        std::vector<uint8_t> s1(n_sg * kLen), s2(n_sg * kLen);
        for (size_t k = 0; k < n_sg; ++k) {
            memcpy(&s1[k * kLen], one_a.data(), kLen);
            memcpy(&s2[k * kLen], one_b.data(), kLen);
        }
        std::vector<int32_t> scores(n_sg), tb(n_sg * kCap * 2);
        std::vector<uint32_t> lengths(n_sg);
        if (swmi_semiglobal_xdrop(s1.data(), s2.data(), n_sg, scores.data(), tb.data(), kCap, lengths.data()) != SWMI_OK)
            die("swmi_semiglobal_xdrop");     // first call: device buffers are allocated (and kept)
        const double t0 = now_ms();
        if (swmi_semiglobal_xdrop(s1.data(), s2.data(), n_sg, scores.data(), tb.data(), kCap, lengths.data()) != SWMI_OK)
            die("swmi_semiglobal_xdrop");
        const double ms = now_ms() - t0;
        printf("mi355x batch version: %.0f ms / %zuK incl. PCIe  (score %d, %u traceback positions, %.1f k alignments/s)\n", ms,
               n_sg / 1000, scores[0], lengths[0], n_sg / ms);
        {   // the same call with the traceback as 2-bit moves (8 KB per alignment back over the link instead of 262 KB), and
            // the positions rebuilt on the host, as swmi_compat.hpp's SemiGlobal_mi355x_batch does it
            std::vector<uint64_t> moves(n_sg * size_t(SWMI_SG_MOVE_WORDS));
            std::vector<int32_t> sc2(n_sg);
            std::vector<uint32_t> len2(n_sg);
            if (swmi_semiglobal_xdrop_moves(s1.data(), s2.data(), n_sg, sc2.data(), moves.data(), len2.data()) != SWMI_OK) die("swmi_semiglobal_xdrop_moves");
            const double t1 = now_ms();
            if (swmi_semiglobal_xdrop_moves(s1.data(), s2.data(), n_sg, sc2.data(), moves.data(), len2.data()) != SWMI_OK) die("swmi_semiglobal_xdrop_moves");
            const double ms_moves = now_ms() - t1;
            printf("mi355x batch version, traceback as moves: %.0f ms / %zuK incl. PCIe  (%.1f k alignments/s)\n", ms_moves, n_sg / 1000, n_sg / ms_moves);
            const unsigned hw = std::thread::hardware_concurrency();
            for (unsigned threads : {1u, 4u, 16u, hw > 16 ? (hw > 64 ? 64u : hw) : 0u}) {
                if (!threads) continue;
                std::vector<int32_t> tb2(n_sg * kCap * 2);
                const double t2 = now_ms();
                std::vector<std::thread> pool;
                auto work = [&](size_t lo, size_t hi) {
                    for (size_t k = lo; k < hi; ++k)
                        swmi_semiglobal_expand_moves(&moves[k * size_t(SWMI_SG_MOVE_WORDS)], len2[k], &tb2[k * kCap * 2], kCap);
                };
                for (unsigned t = 1; t < threads; ++t) pool.emplace_back(work, n_sg * t / threads, n_sg * (t + 1) / threads);
                work(0, n_sg / threads);
                for (auto &th : pool) th.join();
                const double ms_exp = now_ms() - t2;
                const bool same = sc2 == scores && len2 == lengths && memcmp(tb2.data(), tb.data(), size_t(lengths[0]) * 8) == 0 &&
                                  memcmp(&tb2[(n_sg - 1) * kCap * 2], &tb[(n_sg - 1) * kCap * 2], size_t(lengths[n_sg - 1]) * 8) == 0;
                printf("   host expansion of the moves alone on %2u thread(s): %.0f ms / %zuK (%.1f k alignments/s); %s\n", threads, ms_exp,
                       n_sg / 1000, n_sg / ms_exp, same ? "same positions" : "POSITIONS DIFFER");
            }
        }
        {   // the reference's own result type, std::pair<int, std::vector<std::pair<int,int>>> per alignment, through the header-only
            // overload: moves over the link, positions rebuilt on host threads one slice behind the GPU
            std::vector<std::array<uint8_t, 16384>> va(n_sg), vb(n_sg);
            memcpy(va[0].data(), s1.data(), n_sg * kLen);
            memcpy(vb[0].data(), s2.data(), n_sg * kLen);
            for (unsigned threads : {4u, 16u}) {
                const double t3 = now_ms();
                const auto res = swmi::SemiGlobal_mi355x_batch(va, vb, threads);
                const double ms_all = now_ms() - t3;
                bool same = res.size() == n_sg;
                for (size_t k = 0; same && k < n_sg; k += 997)
                    same = res[k].first == scores[k] && res[k].second.size() == lengths[k] &&
                           memcmp(res[k].second.data(), &tb[k * kCap * 2], size_t(lengths[k]) * 8) == 0;
                printf("mi355x SemiGlobal_mi355x_batch (vector of (score, traceback) pairs, %u expander threads): %.0f ms / %zuK incl. PCIe "
                       "(%.1f k alignments/s); %s\n", threads, ms_all, n_sg / 1000, n_sg / ms_all, same ? "same positions" : "POSITIONS DIFFER");
            }
        }
        void *d1 = nullptr, *d2 = nullptr, *dsc = nullptr, *dlen = nullptr, *dtb = nullptr;
        if (hipMalloc(&d1, n_sg * kLen) != hipSuccess || hipMalloc(&d2, n_sg * kLen) != hipSuccess ||
            hipMalloc(&dsc, n_sg * 4) != hipSuccess || hipMalloc(&dlen, n_sg * 4) != hipSuccess ||
            hipMalloc(&dtb, n_sg * kCap * 8) != hipSuccess) {
            fprintf(stderr, "hipMalloc failed\n");
            return 1;
        }
        (void)hipMemcpy(d1, s1.data(), n_sg * kLen, hipMemcpyHostToDevice);
        (void)hipMemcpy(d2, s2.data(), n_sg * kLen, hipMemcpyHostToDevice);
        float phase[2] = {0.f, 0.f};
        if (swmi_semiglobal_time_device(d1, d2, n_sg, dsc, dtb, kCap, dlen, nullptr, phase) != SWMI_OK) die("warmup");
        if (swmi_semiglobal_time_device(d1, d2, n_sg, dsc, dtb, kCap, dlen, nullptr, phase) != SWMI_OK) die("swmi_semiglobal_time_device");
        printf("mi355x device version: %.1f ms / %zuK  (sweep %.1f + traceback %.1f ms, %.1f k alignments/s, inputs resident in HBM)\n",
               phase[0] + phase[1], n_sg / 1000, phase[0], phase[1], n_sg / (phase[0] + phase[1]));
        (void)hipFree(d1); (void)hipFree(d2); (void)hipFree(dsc); (void)hipFree(dlen); (void)hipFree(dtb);
    }
    swmi_shutdown();

    // == the 1M-call loop of SpeedTest (source.cpp:3074-3082) over every GPU of the node, driven by this one process
    {
        int G = 0;
        if (const char *list = getenv("SWMI_SPEEDTEST_DEVICES")) {
            std::vector<int> devs;
            for (const char *p = list; *p;) {
                devs.push_back(atoi(p));
                while (*p && *p != ',') ++p;
                if (*p == ',') ++p;
            }
            G = swmi_init_devices(devs.data(), (int)devs.size());
        } else {
            G = swmi_init_all(0);
        }
        if (G <= 0) die("swmi_init_all");
        printf("== SpeedTest (10,-30,15) on %d GPU context(s) from one process\n", G);
        std::array<int8_t, 16> sm;
        for (int i = 0; i < 16; ++i) sm[i] = int8_t(i % 5 == 0 ? 10 : -30);
        const int8_t gap = 15;
        long long want_sum = 0;
        {
            std::vector<uint8_t> s1(n * 128), s2(n * 128);
            std::vector<int32_t> out(n);
            swmi_generate_pairs_host(s1.data(), s2.data(), n, 10000, 0);
            if (swmi_score_batch_multi(s1.data(), s2.data(), n, sm.data(), gap, out.data()) != SWMI_OK) die("swmi_score_batch_multi");
            double ms = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                const double t0 = now_ms();
                if (swmi_score_batch_multi(s1.data(), s2.data(), n, sm.data(), gap, out.data()) != SWMI_OK) die("swmi_score_batch_multi");
                ms = std::min(ms, now_ms() - t0);
            }
            for (int32_t s : out) want_sum += s;
            printf("mi355x %d-gpu batch version: %.1f ms / %.0fM distinct pairs incl. PCIe  (%.1f M alignments/s, checksum %lld)\n", G, ms,
                   n / 1e6, n / ms / 1e3, want_sum);
        }
        const size_t per_gpu[2] = {n, argc > 4 ? strtoull(argv[4], nullptr, 10) : 0};   // argv[4]: a second, larger size per GPU (e.g. 67108864)
        for (size_t pg : per_gpu) {
            if (pg == 0) continue;
            const size_t total = pg * size_t(G);
            swmi_sharded_batch *sb = nullptr;
            if (swmi_sharded_create(total, 0, &sb) != SWMI_OK) die("swmi_sharded_create");
            if (swmi_sharded_generate(sb, 10000, 0) != SWMI_OK) die("swmi_sharded_generate");
            std::vector<float> k_ms(G), g_ms(G);
            double wall = 0;
            for (int mode : {SWMI_GATHER_NONE, SWMI_GATHER_ROOT, SWMI_GATHER_ALL}) {
                if (swmi_sharded_time(sb, sm.data(), gap, mode, 3, nullptr, nullptr, nullptr) != SWMI_OK) die("warmup");
                if (swmi_sharded_time(sb, sm.data(), gap, mode, 20, k_ms.data(), g_ms.data(), &wall) != SWMI_OK) die("swmi_sharded_time");
                float kmax = 0, gmax = 0;
                for (int g = 0; g < G; ++g) { kmax = k_ms[g] > kmax ? k_ms[g] : kmax; gmax = g_ms[g] > gmax ? g_ms[g] : gmax; }
                printf("mi355x %d-gpu device version: %.3f ms / %.0fM per GPU, %s  (%.1f M alignments/s whole job; slowest kernel %.3f ms, "
                       "slowest gather %.3f ms)\n", G, wall, pg / 1e6,
                       mode == SWMI_GATHER_NONE ? "scores left sharded" : mode == SWMI_GATHER_ROOT ? "gathered to GPU 0 (peer DMA)"
                   : swmi_sharded_gather_backend(sb) == 2 ? "all-gathered (RCCL)" : "all-gathered (peer copies: a GPU is bound twice)",
                       total / wall / 1e3, kmax, gmax);
            }
            if (pg == n) {      // the first shard-0-sized prefix of the gathered vector must be the host batch's scores
                std::vector<int32_t> got(total);
                if (swmi_sharded_gathered_host(sb, 0, got.data()) != SWMI_OK) die("swmi_sharded_gathered_host");
                long long sum = 0;
                for (size_t k = 0; k < n; ++k) sum += got[k];
                printf("gathered vector: first %.0fM scores checksum %lld (%s the host batch)\n", n / 1e6, sum,
                       sum == want_sum ? "equals" : "DIFFERS FROM");
                if (sum != want_sum) return 1;
            }
            swmi_sharded_destroy(sb);
        }
        swmi_shutdown();
    }
    return 0;
}
