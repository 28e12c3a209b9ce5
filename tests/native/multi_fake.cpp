// multi_fake.cpp -- the multi-GPU host logic of libswmi (swmi_multi.cpp, swmi_api.cpp) on THREE fake GPUs (fake_hip.cpp) and a
// fake RCCL (fake_rccl.cpp): the branches a one-GPU box cannot run -- peer copies between DISTINCT devices, the multi-rank RCCL
// all-gather, the grouped broadcasts of ragged shards, empty shards -- executed for real as far as the host code goes: what is
// launched on which device and stream, in which order, with which counts and offsets, and where every score ends up.
// The fake launchers return as the score of a pair the number stored in its first four bytes, so every gathered vector must
// read 0, 1, 2, ...  Built and run by tests/test_multi_fake.py (g++, no GPU, no HIP runtime).
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/swmi.h"

extern "C" size_t fake_hip_log_size();
extern "C" const char *fake_hip_log_at(size_t);
extern "C" void fake_hip_log_clear();

#define CHECK(cond)                                                                                          \
    do {                                                                                                     \
        if (!(cond)) {                                                                                       \
            fprintf(stderr, "CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #cond, swmi_last_error()); \
            exit(1);                                                                                         \
        }                                                                                                    \
    } while (0)

static std::vector<std::string> hip_log()
{
    std::vector<std::string> v;
    for (size_t k = 0; k < fake_hip_log_size(); ++k) v.emplace_back(fake_hip_log_at(k));
    return v;
}
static int count_of(const std::vector<std::string> &log, const char *needle)
{
    int c = 0;
    for (auto &s : log) c += s.find(needle) != std::string::npos;
    return c;
}
static void numbered_pairs(size_t n, std::vector<uint8_t> &a, std::vector<uint8_t> &b)
{
    a.assign(n * 128, 0); b.assign(n * 128, 0);
    for (size_t k = 0; k < n; ++k) { const uint32_t id = (uint32_t)k; memcpy(&a[k * 128], &id, 4); }
}
static bool counts_up(const std::vector<int32_t> &v) { for (size_t k = 0; k < v.size(); ++k) if (v[k] != (int32_t)k) return false; return true; }

int main(int argc, char **argv)
{
    CHECK(argc == 2);                           // path of the fake librccl
    const bool fail_init = getenv("FAKE_NCCL_FAIL_INIT") != nullptr;
    setenv("SWMI_RCCL_LIB", argv[1], 1);
    void *rccl = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    CHECK(rccl);
    auto rlog_size = (size_t (*)())dlsym(rccl, "fake_rccl_log_size");
    auto rlog_at = (const char *(*)(size_t))dlsym(rccl, "fake_rccl_log_at");
    auto rlog_clear = (void (*)())dlsym(rccl, "fake_rccl_log_clear");
    auto rccl_errors = (int (*)())dlsym(rccl, "fake_rccl_errors");
    CHECK(rlog_size && rlog_at && rlog_clear && rccl_errors);
    auto rccl_log = [&] { std::vector<std::string> v; for (size_t k = 0; k < rlog_size(); ++k) v.emplace_back(rlog_at(k)); return v; };

    int8_t sm[16];
    for (int i = 0; i < 16; ++i) sm[i] = int8_t(i % 5 == 0 ? 10 : -30);
    CHECK(swmi_init_all(0) == 3 && swmi_num_gpus() == 3);
    CHECK(count_of(hip_log(), "enable_peer") == 6);                     // every ordered pair of distinct devices

    // ---- host arrays over three GPUs: each shard through its GPU's pipeline into the caller's slice ----
    for (size_t n : {size_t(1), size_t(2), size_t(3), size_t(1000), size_t(70001)}) {
        std::vector<uint8_t> a, b;
        numbered_pairs(n, a, b);
        std::vector<int32_t> out(n, -1);
        fake_hip_log_clear();
        CHECK(swmi_score_batch_multi(a.data(), b.data(), n, sm, 15, out.data()) == SWMI_OK);
        CHECK(counts_up(out));
        const auto log = hip_log();
        for (int g = 0; g < 3; ++g) {
            size_t lo, hi;
            CHECK(swmi_shard_bounds(n, g, 3, &lo, &hi) == SWMI_OK);
            const std::string dev = "dev" + std::to_string(g) + " launch_score";
            CHECK((count_of(log, dev.c_str()) > 0) == (hi > lo));        // an empty shard launches nothing
        }
    }

    // ---- resident shards, ragged (334 + 333 + 333), gather to the root over peer copies ----
    const size_t n = 1000;
    std::vector<uint8_t> a, b;
    numbered_pairs(n, a, b);
    swmi_sharded_batch *sb = nullptr;
    CHECK(swmi_sharded_create(n, 0, &sb) == SWMI_OK);
    CHECK(swmi_sharded_upload(sb, a.data(), b.data()) == SWMI_OK);
    fake_hip_log_clear();
    CHECK(swmi_sharded_score(sb, sm, 15, SWMI_GATHER_ROOT) == SWMI_OK && swmi_sharded_wait(sb) == SWMI_OK);
    {
        const auto log = hip_log();
        CHECK(count_of(log, "launch_score") == 3);
        CHECK(count_of(log, "dev0 launch_score n334") == 1 && count_of(log, "dev1 launch_score n333") == 1 && count_of(log, "dev2 launch_score n333") == 1);
        // the root's own shard is a device-to-device copy; the others are peer copies issued on the SOURCE device
        CHECK(count_of(log, "memcpy_peer") == 2);
        CHECK(count_of(log, "dev1 memcpy_peer dst_dev0 src_dev1 bytes1332") == 1 && count_of(log, "dev2 memcpy_peer dst_dev0 src_dev2 bytes1332") == 1);
        // every kernel is launched before the first gather copy (all GPUs compute, then exchange)
        size_t last_launch = 0, first_peer = log.size();
        for (size_t k = 0; k < log.size(); ++k) {
            if (log[k].find("launch_score") != std::string::npos) last_launch = k;
            if (log[k].find("memcpy_peer") != std::string::npos && k < first_peer) first_peer = k;
        }
        CHECK(last_launch < first_peer);
    }
    std::vector<int32_t> got(n, -1);
    CHECK(swmi_sharded_gathered_host(sb, 0, got.data()) == SWMI_OK && counts_up(got));
    CHECK(swmi_sharded_gathered_host(sb, 1, got.data()) == SWMI_ERR_INVALID_ARGUMENT);      // nothing was gathered to GPU 1
    std::fill(got.begin(), got.end(), -1);
    CHECK(swmi_sharded_scores_host(sb, got.data()) == SWMI_OK && counts_up(got));

    // ---- SWMI_GATHER_ALL: three RCCL ranks, ragged shards -> grouped broadcasts, the same order on every rank ----
    rlog_clear();
    CHECK(swmi_sharded_score(sb, sm, 15, SWMI_GATHER_ALL) == SWMI_OK && swmi_sharded_wait(sb) == SWMI_OK);
    char note[256];
    if (fail_init) {
        CHECK(swmi_sharded_gather_backend(sb) == 1);
        CHECK(swmi_sharded_gather_note(sb, note, sizeof note) == SWMI_OK && strstr(note, "ncclCommInitAll failed"));
        CHECK(strstr(swmi_last_error(), "peer copies"));
    } else {
        CHECK(swmi_sharded_gather_backend(sb) == 2);
        CHECK(swmi_sharded_gather_note(sb, note, sizeof note) == SWMI_OK && note[0] == 0);
        const auto rl = rccl_log();
        CHECK(rl[0] == "comm_init_all dev0 dev1 dev2");
        CHECK(count_of(rl, "group_start") == 1 && count_of(rl, "group_end") == 1 && count_of(rl, "broadcast") == 9 && count_of(rl, "all_gather") == 0);
        for (int r = 0; r < 3; ++r)
            for (int root = 0; root < 3; ++root) {
                const std::string s = "broadcast rank" + std::to_string(r) + " root" + std::to_string(root) + " count" + (root == 0 ? "334" : "333");
                CHECK(count_of(rl, s.c_str()) == 1);
            }
        CHECK(rccl_errors() == 0);
    }
    for (int g = 0; g < 3; ++g) {
        std::fill(got.begin(), got.end(), -1);
        CHECK(swmi_sharded_gathered_host(sb, g, got.data()) == SWMI_OK && counts_up(got));
    }
    // the timing helper drives the same sequence
    float kms[3], gms[3];
    double wall = 0;
    CHECK(swmi_sharded_time(sb, sm, 15, SWMI_GATHER_ALL, 2, kms, gms, &wall) == SWMI_OK && kms[2] > 0);
    CHECK(swmi_sharded_destroy(sb) == SWMI_OK);

    // ---- equal shards (999 = 3 x 333): one ncclAllGather per rank ----
    numbered_pairs(999, a, b);
    CHECK(swmi_sharded_create(999, 0, &sb) == SWMI_OK && swmi_sharded_upload(sb, a.data(), b.data()) == SWMI_OK);
    rlog_clear();
    CHECK(swmi_sharded_score(sb, sm, 15, SWMI_GATHER_ALL) == SWMI_OK && swmi_sharded_wait(sb) == SWMI_OK);
    if (!fail_init) {
        const auto rl = rccl_log();
        CHECK(count_of(rl, "all_gather") == 3 && count_of(rl, "count333") == 3 && count_of(rl, "broadcast") == 0 && rccl_errors() == 0);
    }
    got.assign(999, -1);
    for (int g = 0; g < 3; ++g) CHECK(swmi_sharded_gathered_host(sb, g, got.data()) == SWMI_OK && counts_up(got));
    CHECK(swmi_sharded_destroy(sb) == SWMI_OK);

    // ---- fewer pairs than GPUs: an empty shard launches nothing and takes part in no broadcast ----
    CHECK(swmi_sharded_create(2, 0, &sb) == SWMI_OK);
    CHECK(swmi_sharded_generate(sb, 1, 0) == SWMI_OK);      // the fake generator numbers the pairs too
    fake_hip_log_clear();
    rlog_clear();
    CHECK(swmi_sharded_score(sb, sm, 15, SWMI_GATHER_ALL) == SWMI_OK && swmi_sharded_wait(sb) == SWMI_OK);
    CHECK(count_of(hip_log(), "launch_score") == 2 && count_of(hip_log(), "dev2 launch_score") == 0);
    if (!fail_init) CHECK(count_of(rccl_log(), "broadcast") == 6 && count_of(rccl_log(), "root2") == 0 && rccl_errors() == 0);
    got.assign(2, -1);
    for (int g = 0; g < 3; ++g) CHECK(swmi_sharded_gathered_host(sb, g, got.data()) == SWMI_OK && counts_up(got));
    CHECK(swmi_sharded_destroy(sb) == SWMI_OK);

    // ---- the semi-global aligner's per-stream workspace and the lifetime of its window counters (the kernels are stand-ins) ----
    {
        CHECK(swmi_use_gpu(2) == SWMI_OK);
        uint64_t counts[4] = {9, 9, 9, 9};
        CHECK(swmi_semiglobal_window_stats(nullptr, counts) == SWMI_ERR_INVALID_ARGUMENT);       // no call on this stream yet
        CHECK(strstr(swmi_last_error(), "no semi-global call") && counts[0] == 0 && counts[3] == 0);
        // (a fake GPU's memory is the host's: any 16-byte aligned buffer is a "device" pointer)
        void *d1 = aligned_alloc(64, 4 * 16384), *d2 = aligned_alloc(64, 4 * 16384), *ds = aligned_alloc(64, 64), *dl = aligned_alloc(64, 64),
             *dm = aligned_alloc(64, 4 * SWMI_SG_MOVE_WORDS * 8);
        CHECK(d1 && d2 && ds && dl && dm);
        CHECK(swmi_semiglobal_xdrop_moves_device(d1, d2, 4, ds, dm, dl, nullptr) == SWMI_OK);
        CHECK(swmi_semiglobal_window_stats(nullptr, counts) == SWMI_OK);                            // (whatever the stand-in left there)
        CHECK(swmi_semiglobal_window_stats(nullptr, nullptr) == SWMI_ERR_INVALID_ARGUMENT);
        CHECK(swmi_semiglobal_release_workspaces() == SWMI_OK);
        CHECK(swmi_semiglobal_window_stats(nullptr, counts) == SWMI_ERR_INVALID_ARGUMENT);       // the workspace is gone
        for (void *q : {d1, d2, ds, dl, dm}) free(q);
    }

    // ---- the host-batch pipeline on one fake GPU: granule order, slots, one score copy at the end ----
    CHECK(swmi_use_gpu(1) == SWMI_OK);
    const size_t big = (size_t(1) << 20) + 12345;
    numbered_pairs(big, a, b);
    std::vector<int32_t> out(big, -1);
    fake_hip_log_clear();
    CHECK(swmi_score_batch(a.data(), b.data(), big, sm, 15, out.data()) == SWMI_OK && counts_up(out));
    {
        const auto log = hip_log();
        size_t gr[16];
        const size_t ng = swmi_host_granules(big, gr, 16);
        CHECK(ng >= 4 && (size_t)count_of(log, "dev1 launch_score") == ng);
        CHECK((size_t)count_of(log, "kind1") == 2 * ng);                 // two host-to-device copies per granule
        CHECK(count_of(log, "kind2") == 1);                              // ONE device-to-host copy, after everything else
        size_t k2 = 0, last_launch = 0;
        for (size_t k = 0; k < log.size(); ++k) {
            if (log[k].find("kind2") != std::string::npos) k2 = k;
            if (log[k].find("launch_score") != std::string::npos) last_launch = k;
        }
        CHECK(k2 > last_launch);
    }
    CHECK(swmi_shutdown() == SWMI_OK);

    // ---- a host batch that spans SEVERAL score groups (swmi_api.cpp score_host_batch; production group: 16M pairs, here the
    // test-only knob makes it 64K): every group's scores leave in ONE copy at the right host offset, and that copy has drained
    // before the next group's first kernel overwrites the device score vector.  The fake holds device-to-host copies back
    // until their stream is synchronised, so a missing drain shows as wrong scores, not only as a wrong log. ----
    for (int serial = 0; serial < 2; ++serial) {
        setenv("SWMI_TEST_SCORE_GROUP", "65536", 1);
        setenv("SWMI_HOST_MIN_GRANULE", "8192", 1);                 // ten granules per group for the packed entry: its two issuing
                                                                    //   threads and the early score copy take part
        if (serial) setenv("SWMI_HOST_SERIAL", "1", 1);
        CHECK(swmi_init(1) == SWMI_OK);
        const size_t group = 65536, many = 3 * group + 12345;       // four groups, the last one ragged
        numbered_pairs(many, a, b);
        for (int entry = 0; entry < 3; ++entry) {                   // pairs, 2-bit packed (the fake reads the id at stride 32), one-vs-many
            std::vector<uint8_t> pa;
            if (entry == 1) {
                pa.assign(many * 32, 0);
                for (size_t k = 0; k < many; ++k) memcpy(&pa[k * 32], &a[k * 128], 4);
            }
            std::vector<int32_t> sc(many, -1);
            fake_hip_log_clear();
            const int rc = entry == 0 ? swmi_score_batch(a.data(), b.data(), many, sm, 15, sc.data())
                         : entry == 1 ? swmi_score_batch_packed(pa.data(), pa.data(), many, sm, 15, sc.data())
                                      : swmi_score_one_vs_many(a.data(), many, b.data(), sm, 15, sc.data());
            CHECK(rc == SWMI_OK && counts_up(sc));
            const auto log = hip_log();
            size_t gr[256];
            const size_t ng = swmi_host_granules_for(many, entry, gr, 256);
            CHECK(ng <= 256 && (size_t)count_of(log, entry == 2 ? "launch_one_vs_many" : "launch_score") == ng);
            // walk the log group by group
            size_t at = 0, gi = 0;
            int groups_with_two_copies = 0;
            for (size_t g0 = 0; g0 < many; g0 += group) {
                const size_t gn = many - g0 < group ? many - g0 : group;
                size_t launched = 0, d2h_bytes = 0;
                int d2h = 0;
                bool synced_after_d2h = false;
                for (; at < log.size(); ++at) {
                    const std::string &l = log[at];
                    if (l.find("launch_") != std::string::npos) {
                        if (launched == gn) break;                  // the next group's first kernel
                        // (non-serial) no kernel of a group behind its score copy -- the packed entry's EARLY copy (everything but
                        // the last two granules, issued by the helper thread) may have the last kernels behind it
                        CHECK(d2h == 0 || serial || (entry == 1 && d2h == 1));
                        launched += (size_t)atoll(l.c_str() + l.find(" n") + 2);      // (two issuing threads: any order inside a group)
                        ++gi;
                    } else if (l.find("kind2") != std::string::npos) {
                        ++d2h;
                        d2h_bytes += (size_t)atoll(l.c_str() + l.find("bytes") + 5);
                        synced_after_d2h = false;
                    } else if (l.find("stream_sync") != std::string::npos && d2h) {
                        synced_after_d2h = true;
                    }
                }
                CHECK(launched == gn && d2h_bytes == gn * 4);
                CHECK(serial || d2h == 1 || (entry == 1 && d2h == 2));      // ONE score copy per group (packed entry: early part + rest)
                groups_with_two_copies += d2h == 2;
                CHECK(synced_after_d2h);                            // ... drained before the next group starts (or the call returns)
            }
            CHECK(gi == ng);
            CHECK(serial || entry != 1 || groups_with_two_copies == 4);     // the packed entry's early copy did take place in every group
            (void)gr;
        }
        CHECK(swmi_shutdown() == SWMI_OK);
    }
    unsetenv("SWMI_TEST_SCORE_GROUP");
    unsetenv("SWMI_HOST_MIN_GRANULE");
    unsetenv("SWMI_HOST_SERIAL");
    dlclose(rccl);
    printf("multi fake ok\n");
    return 0;
}
