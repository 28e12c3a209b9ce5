// hbm_stream.hip -- what hand-written streaming kernels reach on ONE MI355X: fill, copy and a read-only sweep with 16 bytes
// per lane, default and non-temporal cache policy, a few grid shapes.  The yardstick for the semi-global traceback kernels
// (sg_walk_lane_kernel reads 17.5 GB of records, sg_expand_kernel writes 8.6 GB of positions at 65536 alignments):
// round 3 compared them with torch's fill_ / copy_ (5.5 / 5.0 TB/s), which are not the chip's ceiling.
// Build: make -C tools/microbench hbm_stream     Run: tools/microbench/hbm_stream [GiB per buffer, default 8]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <functional>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__);  \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

// grid-stride, UNROLL independent 16-byte accesses per lane and trip; a workgroup touches UNROLL contiguous blocks of
// 256 x 16 B = 4 KiB that lie `gridDim.x * 4 KiB` apart (so that neighbouring workgroups share DRAM pages at any one time)
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) fill_kernel(v4u *__restrict__ dst, size_t n16, unsigned value)
{
    const size_t stride = (size_t)gridDim.x * 256;
    const v4u v = {value, value + 1, value + 2, value + 3};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v, dst + i + u * stride);
            else dst[i + u * stride] = v;
        }
    }
    for (; i < n16; i += stride) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <int UNROLL, bool NT_LOAD, bool NT_STORE>
__global__ void __launch_bounds__(256) copy_kernel(v4u *__restrict__ dst, const v4u *__restrict__ src, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT_LOAD ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT_STORE) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) {
        const v4u v = NT_LOAD ? __builtin_nontemporal_load(src + i) : src[i];
        if (NT_STORE) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) read_kernel(const v4u *__restrict__ src, size_t n16, unsigned *__restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v4u acc = {0, 0, 0, 0};
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= NT ? __builtin_nontemporal_load(src + i) : src[i];
    const unsigned x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x9E3779B9u) *sink = x;                      // (never true for the fill pattern: keeps the loads alive)
}

// one 16-byte access per lane, no loop: what a "one thread per element" launch gives
template <bool NT>
__global__ void __launch_bounds__(256) fill_flat_kernel(v4u *__restrict__ dst, size_t n16, unsigned value)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4u v = {value, value + 1, value + 2, value + 3};
    if (i < n16) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

// The semi-global walk's access shape (sg_walk_lane_kernel): one wavefront per workgroup owns WALKS walks; window w of the
// batch is a block of n x 128 bytes, of which the wavefront wants its WALKS x 128 contiguous bytes; it goes through the
// windows top down with DEPTH windows requested ahead and does nothing with the data (one XOR per piece).  What this reaches
// is the ceiling of the walk's fetch pattern at a given number of wavefronts and bytes in flight.
template <int DEPTH, int WALKS>
__global__ void __launch_bounds__(64) walk_shape_kernel(const v4u *__restrict__ src, uint32_t n, int windows, unsigned *__restrict__ sink)
{
    constexpr int P = WALKS / 8;                          // 16-byte pieces per lane and window
    const int lane = threadIdx.x;
    const v4u *mine = src + ((size_t)blockIdx.x * WALKS * 8 + lane);      // piece i at + i * 64
    const size_t win_stride = (size_t)n * 8;              // uint4 per window
    v4u buf[DEPTH][P];
    v4u acc = {0, 0, 0, 0};
    int w = windows - 1;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < P; ++i) buf[d][i] = mine[(size_t)(w - d > 0 ? w - d : 0) * win_stride + i * 64];
    for (; w >= 0; w -= DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < P; ++i) acc ^= buf[d][i];
            const int nw = w - d - DEPTH;
#pragma unroll
            for (int i = 0; i < P; ++i) buf[d][i] = mine[(size_t)(nw > 0 ? nw : 0) * win_stride + i * 64];
        }
    }
    const unsigned x = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (x == 0x9E3779B9u) *sink = x;
}

// The semi-global expand kernel's store shape (sg_expand_kernel): every wavefront writes its own region of `region` bytes
// front to back, 4 KB (four 1 KB non-temporal stores) per trip; COOP = the four wavefronts of a workgroup take one region
// together, 16 KB contiguous per trip, and the workgroup goes through its four regions one after another.
template <bool COOP>
__global__ void __launch_bounds__(256) expand_shape_kernel(v4u *__restrict__ dst, size_t region16 /* uint4 per region */, unsigned value)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const v4u v = {value, value + 1, value + 2, value + 3};
    if (!COOP) {
        v4u *out = dst + ((size_t)blockIdx.x * 4 + wv) * region16;
        for (size_t base = 0; base + 256 <= region16; base += 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v, out + base + 64 * j + lane);
        }
    } else {
        for (int r = 0; r < 4; ++r) {
            v4u *out = dst + ((size_t)blockIdx.x * 4 + r) * region16;
            for (size_t base = (size_t)wv * 256; base + 256 <= region16; base += 1024) {
#pragma unroll
                for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(v, out + base + 64 * j + lane);
            }
        }
    }
}

// What a ONE-SHOT expand would look like: one wavefront per 4 KB chunk of a region (33 chunks used per 128 KB region), which
// first reads the part of the region's 4 KB of move words that lies before (or behind) its chunk -- whichever is shorter,
// 64 words per pass, cache hits after the region's first wavefront -- and then writes its 4 KB.
template <int STORES, bool PREFIX>                       // STORES x 1 KB per wavefront
__global__ void __launch_bounds__(256) expand_flat_shape_kernel(v4u *__restrict__ dst, const unsigned long long *__restrict__ moves,
                                                                size_t region16, int chunks, unsigned value)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv;
    const size_t r = wave / chunks;
    const int j = (int)(wave % chunks);
    const unsigned long long *mine = moves + r * 520;     // 4160 bytes of moves per region
    const int before = j * 16, behind = (chunks - j) * 16;               // (in words of 32 moves = 256 bytes of positions)
    const int first = before <= behind ? 0 : before, words = !PREFIX ? 0 : (before <= behind ? before : behind) * STORES / 4;
    unsigned long long acc = 0;
    for (int w = lane; w < words; w += 64) acc += __popcll(mine[first + w]);
    unsigned red = (unsigned)acc;
    red += (unsigned)__builtin_amdgcn_update_dpp(0, (int)red, 0x111, 0xf, 0xf, true);
    red += (unsigned)__builtin_amdgcn_update_dpp(0, (int)red, 0x112, 0xf, 0xf, true);
    red += (unsigned)__builtin_amdgcn_update_dpp(0, (int)red, 0x114, 0xf, 0xf, true);
    red += (unsigned)__builtin_amdgcn_update_dpp(0, (int)red, 0x118, 0xf, 0xf, true);
    const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)red, 15) + (unsigned)__builtin_amdgcn_readlane((int)red, 31) +
                         (unsigned)__builtin_amdgcn_readlane((int)red, 47) + (unsigned)__builtin_amdgcn_readlane((int)red, 63);
    const v4u v = {value + tot, value + 1, value + 2, value + 3};
    v4u *out = dst + r * region16 + (size_t)j * (64 * STORES);
#pragma unroll
    for (int q = 0; q < STORES; ++q) __builtin_nontemporal_store(v, out + 64 * q + lane);
}

// ... and the one-shot form WITH a checkpoint: one wavefront per 1 KB chunk reads one 8-byte checkpoint and 64 bytes of the
// region's move words (what it would need to place itself: a running position every 128 moves from the walk + the moves
// between it and the chunk) and writes its kilobyte.
__global__ void __launch_bounds__(256) expand_chk_shape_kernel(v4u *__restrict__ dst, const unsigned long long *__restrict__ moves,
                                                               size_t region16, unsigned value)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + wv;
    const size_t r = wave / 128;
    const int j = (int)(wave % 128);
    const unsigned long long *mine = moves + r * 664;    // 4160 bytes of moves + 130 checkpoints of 8 bytes per region
    const unsigned long long chk = mine[520 + j];
    const unsigned long long w = lane < 8 ? mine[4 * j + (lane & 7)] : 0ull;
    unsigned red = (unsigned)__popcll(w) + (unsigned)(chk & 0xFFFF);
    red += (unsigned)__builtin_amdgcn_update_dpp(0, (int)red, 0x111, 0xf, 0xf, true);
    red += (unsigned)__builtin_amdgcn_update_dpp(0, (int)red, 0x112, 0xf, 0xf, true);
    red += (unsigned)__builtin_amdgcn_update_dpp(0, (int)red, 0x114, 0xf, 0xf, true);
    const unsigned tot = (unsigned)__builtin_amdgcn_readlane((int)red, 7);
    const v4u v = {value + tot, value + 1, value + 2, value + 3};
    __builtin_nontemporal_store(v, dst + r * region16 + (size_t)j * 64 + lane);
}

static double time_ms(hipStream_t st, int reps, const std::function<void()> &launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    launch();
    CHECK(hipStreamSynchronize(st));
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a, st));
        launch();
        CHECK(hipEventRecord(b, st));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return t[t.size() / 2];
}

int main(int argc, char **argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 8.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) & ~size_t(4095), n16 = bytes / 16;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs; %.2f GiB per buffer (%.2f GB); median of 9 launches, HIP events\n", prop.gcnArchName, cus, gib, bytes / 1e9);
    v4u *a = nullptr, *b = nullptr;
    unsigned *sink = nullptr;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&sink, 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    CHECK(hipMemsetAsync(a, 1, bytes, st));
    CHECK(hipMemsetAsync(b, 2, bytes, st));
    CHECK(hipStreamSynchronize(st));
    auto report = [&](const char *what, int wg_per_cu, double moved, double ms) {
        printf("%-44s %3d WG/CU  %8.3f ms  %6.2f TB/s\n", what, wg_per_cu, ms, moved / ms / 1e9);
        fflush(stdout);
    };
    {
        const unsigned grid = (unsigned)((n16 + 255) / 256);
        report("fill, one 16 B store per lane", 0, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL(fill_flat_kernel<false>, dim3(grid), dim3(256), 0, st, a, n16, 7u); }));
        report("fill nt, one 16 B store per lane", 0, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL(fill_flat_kernel<true>, dim3(grid), dim3(256), 0, st, a, n16, 7u); }));
    }
    for (int wg : {4, 8, 16, 32}) {
        const unsigned grid = (unsigned)(cus * wg);
        report("fill, grid-stride x4", wg, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((fill_kernel<4, false>), dim3(grid), dim3(256), 0, st, a, n16, 7u); }));
        report("fill nt, grid-stride x4", wg, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((fill_kernel<4, true>), dim3(grid), dim3(256), 0, st, a, n16, 7u); }));
        report("read, grid-stride x4", wg, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(grid), dim3(256), 0, st, a, n16, sink); }));
        report("read nt, grid-stride x4", wg, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((read_kernel<4, true>), dim3(grid), dim3(256), 0, st, a, n16, sink); }));
        report("read, grid-stride x8", wg, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, st, a, n16, sink); }));
        report("read nt, grid-stride x8", wg, bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, st, a, n16, sink); }));
        report("copy, grid-stride x4 (read + written)", wg, 2.0 * bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((copy_kernel<4, false, false>), dim3(grid), dim3(256), 0, st, b, a, n16); }));
        report("copy nt load + nt store, x4 (read + written)", wg, 2.0 * bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((copy_kernel<4, true, true>), dim3(grid), dim3(256), 0, st, b, a, n16); }));
        report("copy nt store only, x4 (read + written)", wg, 2.0 * bytes, time_ms(st, 9, [&] { hipLaunchKernelGGL((copy_kernel<4, false, true>), dim3(grid), dim3(256), 0, st, b, a, n16); }));
    }
    // the walk's shape: n = 65536 walks, 128 bytes per walk and window, as many windows as the buffer holds
    {
        const uint32_t n = 65536;
        const int windows = (int)(bytes / ((size_t)n * 128));
        const double moved = (double)windows * n * 128;
        printf("# walk shape: %u walks x %d windows x 128 B = %.2f GB\n", n, windows, moved / 1e9);
#define WALK_CASE(D, WK)                                                                                                     \
        {                                                                                                                    \
            char label[96];                                                                                                  \
            snprintf(label, sizeof label, "walk shape: %d walks/wavefront, %d windows ahead", WK, D);                        \
            const double ms = time_ms(st, 9, [&] { hipLaunchKernelGGL((walk_shape_kernel<D, WK>), dim3(n / WK), dim3(64), 0, st, a, n, windows, sink); }); \
            printf("%-52s %5u wavefronts  %8.3f ms  %6.2f TB/s\n", label, n / WK, ms, moved / ms / 1e9);                     \
            fflush(stdout);                                                                                                  \
        }
        WALK_CASE(1, 64) WALK_CASE(2, 64) WALK_CASE(3, 64) WALK_CASE(4, 64) WALK_CASE(6, 64)
        WALK_CASE(2, 32) WALK_CASE(3, 32) WALK_CASE(4, 32) WALK_CASE(8, 32)
        WALK_CASE(2, 16) WALK_CASE(4, 16) WALK_CASE(8, 16)
#undef WALK_CASE
    }
    // the expand kernel's shape: regions of 128 KB (the traceback of a 16395-position path), as many as the buffer holds
    {
        const size_t region = 128 * 1024, regions = bytes / region & ~size_t(3);
        const double moved = (double)regions * region;
        printf("# expand shape: %zu regions of %zu KB = %.2f GB\n", regions, region / 1024, moved / 1e9);
        double ms = time_ms(st, 9, [&] { hipLaunchKernelGGL((expand_shape_kernel<false>), dim3((unsigned)(regions / 4)), dim3(256), 0, st, a, region / 16, 7u); });
        printf("%-60s %8.3f ms  %6.2f TB/s\n", "expand shape: one wavefront per region, 4 KB per trip", ms, moved / ms / 1e9);
        ms = time_ms(st, 9, [&] { hipLaunchKernelGGL((expand_shape_kernel<true>), dim3((unsigned)(regions / 4)), dim3(256), 0, st, a, region / 16, 7u); });
        printf("%-60s %8.3f ms  %6.2f TB/s\n", "expand shape: four wavefronts per region, 16 KB per trip", ms, moved / ms / 1e9);
        fflush(stdout);
        // one-shot forms: one wavefront per 4 KB / 1 KB chunk of a region, with and without the prefix over the move words (buffer b)
#define FLAT_CASE(STORES, PREFIX, LABEL)                                                                                     \
        {                                                                                                                    \
            const int chunks = 128 / STORES;                                                                                 \
            ms = time_ms(st, 9, [&] { hipLaunchKernelGGL((expand_flat_shape_kernel<STORES, PREFIX>), dim3((unsigned)(regions * chunks / 4)), dim3(256), 0, st, a, \
                                                         reinterpret_cast<const unsigned long long *>(b), region / 16, chunks, 7u); }); \
            printf("%-60s %8.3f ms  %6.2f TB/s\n", LABEL, ms, moved / ms / 1e9);                                             \
            fflush(stdout);                                                                                                  \
        }
        FLAT_CASE(4, true, "expand shape: one wavefront per 4 KB chunk, prefix from moves")
        FLAT_CASE(4, false, "expand shape: one wavefront per 4 KB chunk, no prefix")
        FLAT_CASE(1, true, "expand shape: one wavefront per 1 KB chunk, prefix from moves")
        FLAT_CASE(1, false, "expand shape: one wavefront per 1 KB chunk, no prefix")
#undef FLAT_CASE
        ms = time_ms(st, 9, [&] { hipLaunchKernelGGL(expand_chk_shape_kernel, dim3((unsigned)(regions * 128 / 4)), dim3(256), 0, st, a,
                                                     reinterpret_cast<const unsigned long long *>(b), region / 16, 7u); });
        printf("%-60s %8.3f ms  %6.2f TB/s\n", "expand shape: one wavefront per 1 KB chunk, checkpoint + 8 words", ms, moved / ms / 1e9);
        fflush(stdout);
    }
    CHECK(hipFree(a));
    CHECK(hipFree(b));
    CHECK(hipFree(sink));
    return 0;
}
