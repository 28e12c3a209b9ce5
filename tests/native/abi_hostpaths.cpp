// abi_hostpaths.cpp -- the product's HOST code (swmi_api.cpp, swmi_multi.cpp) under AddressSanitizer + UBSan.
// Exercises every path of the C ABI that needs no device: argument and domain checks, the shard rule, the failure paths
// of init / queue / sharded-batch creation, the thread-local error text and the schedule setting from racing threads.
// On a box WITH a usable GPU the same calls take their success paths instead (and are cleaned up); both are accepted.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/swmi.h"

#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            fprintf(stderr, "CHECK failed at line %d: %s (last error: %s)\n", __LINE__, #cond, swmi_last_error()); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

int main()
{
    int8_t sm[16];
    for (int i = 0; i < 16; ++i) sm[i] = int8_t(i % 5 == 0 ? 10 : -30);
    std::vector<uint8_t> a(128 * 70, 1), b(128 * 70, 2);
    std::vector<int32_t> out(70, -1);

    CHECK(swmi_version() == SWMI_VERSION);
    CHECK(strlen(swmi_last_error()) == 0);

    // the shard rule needs no device
    for (size_t n : {size_t(0), size_t(1), size_t(7), size_t(8), size_t(1000003), size_t(1) << 29}) {
        for (int G : {1, 2, 3, 8}) {
            size_t prev = 0;
            for (int g = 0; g < G; ++g) {
                size_t lo = 99, hi = 99;
                CHECK(swmi_shard_bounds(n, g, G, &lo, &hi) == SWMI_OK);
                CHECK(lo == prev && hi >= lo && hi - lo <= n / G + 1);
                prev = hi;
            }
            CHECK(prev == n);
        }
    }
    size_t lo, hi;
    CHECK(swmi_shard_bounds(10, 3, 3, &lo, &hi) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_shard_bounds(10, 0, 0, &lo, &hi) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_shard_bounds(10, 0, 2, nullptr, &hi) == SWMI_ERR_INVALID_ARGUMENT);

    // argument / domain errors come before any device use
    CHECK(swmi_score_batch(a.data(), b.data(), 1, nullptr, 15, out.data()) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_score_batch(a.data(), b.data(), 1, sm, -1, out.data()) == SWMI_ERR_DOMAIN);
    CHECK(strstr(swmi_last_error(), "gap_penalty") != nullptr);
    CHECK(swmi_score_batch(nullptr, nullptr, 0, sm, 15, nullptr) == SWMI_OK);
    CHECK(swmi_score_batch(nullptr, b.data(), 1, sm, 15, out.data()) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_score_batch_multi(nullptr, b.data(), 1, sm, 15, out.data()) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_score_batch_multi(nullptr, nullptr, 0, sm, 15, nullptr) == SWMI_OK);
    CHECK(swmi_score_batch_packed_multi(a.data(), b.data(), 1, sm, -3, out.data()) == SWMI_ERR_DOMAIN);
    CHECK(swmi_score_banded_affine(a.data(), b.data(), 1, 32, sm, 5, 1, out.data()) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_score_banded_affine(a.data(), b.data(), 1, 128, sm, 200, 1, out.data()) == SWMI_ERR_DOMAIN);
    CHECK(swmi_score_batch_device((void *)8, (void *)16, 1, sm, 15, (void *)32, nullptr) == SWMI_ERR_ALIGNMENT);
    CHECK(swmi_semiglobal_xdrop_device((void *)16, (void *)16, 1, (void *)16, (void *)24, 8, (void *)16, nullptr) == SWMI_ERR_ALIGNMENT);
    CHECK(swmi_set_schedule(3, 0) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_set_schedule(4, 16) == SWMI_ERR_INVALID_ARGUMENT);     // flags 1, 2, 4, 8 exist
    CHECK(swmi_use_gpu(-1) == SWMI_ERR_INVALID_ARGUMENT);
    CHECK(swmi_queue_destroy(nullptr) == SWMI_OK);
    CHECK(swmi_sharded_destroy(nullptr) == SWMI_OK);
    CHECK(swmi_sharded_wait(nullptr) == SWMI_ERR_INVALID_ARGUMENT);

    // the host-batch pipeline schedule needs no device: every pair exactly once, tapering
    {
        size_t g[64];
        CHECK(swmi_host_granules(0, g, 64) == 0);
        const size_t c = swmi_host_granules(size_t(1) << 20, g, 64);
        CHECK(c == 4 && g[0] == 786432 && g[1] == 196608 && g[2] == 49152 && g[3] == 16384);
        size_t total = 0;
        const size_t c2 = swmi_host_granules((size_t(1) << 24) + 777, g, 2);     // more granules than the array holds
        CHECK(c2 > 2 && g[0] == size_t(1) << 20);
        (void)total;
    }
    // librccl is loaded on first use; the probe needs no device.  SWMI_RCCL_LIB naming a missing file rehearses the failure
    // path (tests/test_sanitizers.py runs this program a second time that way)
    {
        char why[256] = "x";
        const int usable = swmi_rccl_probe(why, sizeof why);
        if (getenv("SWMI_RCCL_LIB")) CHECK(usable == 0 && strlen(why) > 0);
        else CHECK(usable == 0 || (usable == 1 && strlen(why) == 0));
        CHECK(swmi_rccl_probe(nullptr, 0) == usable);
    }
    CHECK(swmi_sharded_gather_note(nullptr, nullptr, 0) == SWMI_ERR_INVALID_ARGUMENT);

    // before init: everything that needs a device says so (no CPU fallback)
    CHECK(swmi_num_gpus() == 0);
    CHECK(swmi_score_batch(a.data(), b.data(), 70, sm, 15, out.data()) == SWMI_ERR_NOT_INITIALIZED);
    CHECK(swmi_score_pair(a.data(), b.data(), sm, 15) == SWMI_ERR_NOT_INITIALIZED);
    CHECK(swmi_score_batch_multi(a.data(), b.data(), 70, sm, 15, out.data()) == SWMI_ERR_NOT_INITIALIZED);
    swmi_queue *q = (swmi_queue *)0x1;
    CHECK(swmi_queue_create(16, sm, 15, &q) == SWMI_ERR_NOT_INITIALIZED && q == nullptr);
    swmi_sharded_batch *sb = (swmi_sharded_batch *)0x1;
    CHECK(swmi_sharded_create(1000, 0, &sb) == SWMI_ERR_NOT_INITIALIZED && sb == nullptr);
    swmi_device_info info;
    CHECK(swmi_get_device_info(&info) == SWMI_ERR_NOT_INITIALIZED);
    CHECK(swmi_shutdown() == SWMI_OK);                       // shutdown without init is a no-op

    // the schedule setting is one atomic word: racing writers and readers never see a mixed (lanes, flags) pair
    CHECK(swmi_set_schedule(0, 0) == SWMI_OK);
    std::atomic<bool> stop{false};
    std::atomic<int> bad{0};
    std::thread w1([&] { for (int k = 0; k < 20000; ++k) swmi_set_schedule(8, 1); });
    std::thread w2([&] { for (int k = 0; k < 20000; ++k) swmi_set_schedule(16, 0); });
    std::thread r([&] {
        while (!stop.load()) {
            int l = -1; unsigned f = 99;
            swmi_get_schedule(&l, &f);
            const bool ok = (l == 0 && f == 0) || (l == 8 && f == 1) || (l == 16 && f == 0);
            if (!ok) bad.fetch_add(1);
            (void)swmi_schedule_for_batch(1000);
        }
    });
    w1.join(); w2.join(); stop.store(true); r.join();
    CHECK(bad.load() == 0);
    CHECK(swmi_set_schedule(0, 0) == SWMI_OK);

    // error text is per thread
    std::thread t([&] {
        CHECK(strlen(swmi_last_error()) == 0);
        CHECK(swmi_score_batch(a.data(), b.data(), 1, sm, -7, out.data()) == SWMI_ERR_DOMAIN);
        CHECK(strstr(swmi_last_error(), "-7") != nullptr);
    });
    t.join();
    CHECK(strstr(swmi_last_error(), "-7") == nullptr);

    // init: fails cleanly without a device; with one, the whole lifecycle runs under the sanitizers too
    const int rc = swmi_init_all(0);
    if (rc < 0) {
        CHECK(rc == SWMI_ERR_NO_DEVICE || rc == SWMI_ERR_UNSUPPORTED_ARCH || rc == SWMI_ERR_HIP);
        CHECK(strlen(swmi_last_error()) > 0);
        CHECK(swmi_init(0) < 0);
        const int devs[2] = {0, 0};
        CHECK(swmi_init_devices(devs, 2) < 0);
        CHECK(swmi_init_devices(nullptr, 2) == SWMI_ERR_INVALID_ARGUMENT);
        CHECK(swmi_num_gpus() == 0);
    } else {
        CHECK(swmi_num_gpus() == rc);
        CHECK(swmi_score_batch_multi(a.data(), b.data(), 70, sm, 15, out.data()) == SWMI_OK);
        CHECK(swmi_queue_create(16, sm, 15, &q) == SWMI_OK);
        CHECK(swmi_queue_submit(q, a.data(), b.data()) == 0);
        CHECK(swmi_queue_destroy(q) == SWMI_OK);
        CHECK(swmi_sharded_create(1000, 0, &sb) == SWMI_OK);
        CHECK(swmi_sharded_destroy(sb) == SWMI_OK);
        // handles that outlive swmi_shutdown(): calls fail cleanly, destroy still works (no use of freed contexts)
        CHECK(swmi_queue_create(16, sm, 15, &q) == SWMI_OK);
        CHECK(swmi_sharded_create(1000, 0, &sb) == SWMI_OK);
        CHECK(swmi_shutdown() == SWMI_OK);
        CHECK(swmi_queue_wait(q, nullptr, nullptr) == SWMI_ERR_NOT_INITIALIZED);
        CHECK(swmi_sharded_score(sb, sm, 15, SWMI_GATHER_NONE) == SWMI_ERR_NOT_INITIALIZED);
        CHECK(swmi_sharded_wait(sb) == SWMI_ERR_NOT_INITIALIZED);
        CHECK(swmi_queue_destroy(q) == SWMI_OK);
        CHECK(swmi_sharded_destroy(sb) == SWMI_OK);
    }
    CHECK(swmi_shutdown() == SWMI_OK);
    printf("abi hostpaths ok\n");
    return 0;
}
