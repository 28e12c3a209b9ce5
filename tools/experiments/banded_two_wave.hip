// banded_two_wave.hip -- experiment for BASELINE.json configs[4] ("1024 x 1024 affine gap, band 128: multi-wavefront
// anti-diagonal tiling of a single alignment").  NOT part of libswmi: a standalone program that measures the two mappings
// a band of 128 diagonals allows and prints the numbers DESIGN.md section 9 quotes.
//
//   A  one wavefront per alignment, lane m owns the diagonals 2m and 2m+1 and alternates between them (what libswmi ships:
//      every anti-diagonal of the band has exactly 64 cells, so one wavefront holds all of it at 100 % lane use and every
//      neighbour is one DPP lane shift away);
//   B  two wavefronts per alignment, thread d owns diagonal d (SURVEY's sketch): on every anti-diagonal only the threads
//      of one parity have a cell, and the value that crosses between thread 63 and thread 64 goes through LDS with a
//      workgroup barrier per anti-diagonal.
//
// Both compute the same Gotoh recurrences (oracle/sw_oracle.c sw_oracle_banded_affine, open >= ext) and are checked
// against a scalar host version here.  Build and run:  make -C tools/experiments && tools/experiments/banded_two_wave
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

namespace {
constexpr int kMatch = 2, kMismatch = -3, kOpen = 5, kExt = 1;

__device__ __forceinline__ int sat_sub(int a, int b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }
__device__ __forceinline__ int shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true); }
__device__ __forceinline__ int shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true); }

// staging shared by both mappings: entry kp = position kp - 32; seq1 as score rows (4 x int8), seq2 as one-hot dwords
__device__ void stage(const uint8_t *s1, const uint8_t *s2, int len, uint32_t *arow, uint32_t *boh, int tid, int nthreads)
{
    const uint32_t row[4] = {0xFDFDFD02u, 0xFDFD02FDu, 0xFD02FDFDu, 0x02FDFDFDu};   // sm[a][b] = a == b ? 2 : -3
    for (int kp = tid; kp < len + 72; kp += nthreads) {
        const int k = kp - 32;
        const bool in = (unsigned)k < (unsigned)len;
        arow[kp] = in ? row[s1[k] & 3] : 0x80808080u;
        boh[kp] = in ? (1u << (8u * (s2[k] & 3))) : 0u;
    }
}

// A: one wavefront, two diagonals per lane (the recurrences of sw_banded_affine_kernel, saturated form)
__global__ void __launch_bounds__(64) one_wave(const uint8_t *seq1s, const uint8_t *seq2s, int32_t *scores, int len)
{
    extern __shared__ uint32_t lds[];
    uint32_t *arow = lds, *boh = lds + len + 72;
    const int lane = threadIdx.x;
    stage(seq1s + (size_t)blockIdx.x * len, seq2s + (size_t)blockIdx.x * len, len, arow, boh, lane, 64);
    __syncthreads();
    const uint32_t *pa = arow + (64 - lane), *pb = boh + lane;
    int a_cur = (int)pa[0], b_cur = (int)pb[0], best = 0;
    const int oe = kOpen - kExt;
    int h0 = 0, me0 = 0, mf0 = 0, h1 = 0, me1 = 0, mf1 = 0;
    for (int u = 0; u < len; ++u) {
        const int b_next = (int)pb[u + 1], a_next = (int)pa[u + 1];
        {
            const int e = sat_sub(shr1(me1), kExt), f = sat_sub(mf1, kExt);
            const int t = __builtin_amdgcn_sdot4(a_cur, b_cur, h0, true);
            h0 = max(max(t, f), e);
            const int hm = sat_sub(h0, oe);
            me0 = max(e, hm); mf0 = max(f, hm);
        }
        b_cur = b_next;
        {
            const int e = sat_sub(me0, kExt), f = sat_sub(shl1(mf0), kExt);
            const int t = __builtin_amdgcn_sdot4(a_cur, b_cur, h1, true);
            h1 = max(max(t, f), e);
            const int hm = sat_sub(h1, oe);
            me1 = max(e, hm); mf1 = max(f, hm);
        }
        best = max(best, max(h0, h1));
        a_cur = a_next;
    }
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (lane == 0) scores[blockIdx.x] = best;
}

// B: two wavefronts, one diagonal per thread.  Thread d (0..127) owns diagonal d of the band; in iteration u the even
// threads compute their cell in the first half-step and the odd threads in the second -- the same cells, in the same
// order, as mapping A (thread d = 2m / 2m+1 is lane m's even / odd diagonal).  The neighbour across the boundary between
// the wavefronts (threads 63 | 64) is exchanged through LDS, one workgroup barrier per half-step.
__global__ void __launch_bounds__(128) two_waves(const uint8_t *seq1s, const uint8_t *seq2s, int32_t *scores, int len)
{
    extern __shared__ uint32_t lds[];
    uint32_t *arow = lds, *boh = lds + len + 72;
    __shared__ int edge_me[2], edge_mf[2];                // me of thread 63 -> thread 64; mf of thread 64 -> thread 63; two
                                                          // slots alternate, so one barrier per half-step orders them
    __shared__ int wave_best[2];
    const int d = threadIdx.x, lane = d & 63, m = d >> 1;
    const bool odd = d & 1;
    stage(seq1s + (size_t)blockIdx.x * len, seq2s + (size_t)blockIdx.x * len, len, arow, boh, d, 128);
    if (d == 0) { edge_me[0] = edge_me[1] = 0; edge_mf[0] = edge_mf[1] = 0; }
    __syncthreads();
    const uint32_t *pa = arow + (64 - m), *pb = boh + m + (odd ? 1 : 0);
    int best = 0, h = 0, me = 0, mf = 0;                  // this thread's last cell
    const int oe = kOpen - kExt;
    for (int u = 0; u < len; ++u) {
        const int a_cur = (int)pa[u], b_cur = (int)pb[u];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // neighbours: left = thread d-1 (its me), up = thread d+1 (its mf); across the wavefront boundary via LDS
            int me_left = shr1(me), mf_up = shl1(mf);
            if (d == 64) me_left = edge_me[half];
            if (d == 63) mf_up = edge_mf[half];
            if ((half == 1) == odd) {                     // this thread's parity has a cell on this anti-diagonal
                const int e = sat_sub(me_left, kExt), f = sat_sub(mf_up, kExt);
                const int t = __builtin_amdgcn_sdot4(a_cur, b_cur, h, true);
                h = max(max(t, f), e);
                const int hm = sat_sub(h, oe);
                me = max(e, hm); mf = max(f, hm);
                best = max(best, h);
            }
            if (d == 63) edge_me[half ^ 1] = me;
            if (d == 64) edge_mf[half ^ 1] = mf;
            __syncthreads();
        }
    }
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (lane == 0) wave_best[d >> 6] = best;
    __syncthreads();
    if (d == 0) scores[blockIdx.x] = max(wave_best[0], wave_best[1]);
}

int host_gotoh(const uint8_t *a, const uint8_t *b, int len)      // banded (-64 <= j - i <= 63) local affine, open >= ext
{
    std::vector<int> H((len + 1) * (size_t)(len + 1), 0), E(H.size(), 0), F(H.size(), 0);
    int best = 0;
    auto at = [len](int i, int j) { return (size_t)i * (len + 1) + j; };
    for (int i = 1; i <= len; ++i)
        for (int j = std::max(1, i - 64); j <= std::min(len, i + 63); ++j) {
            const bool l_in = j - 1 - i >= -64 && j - 1 >= 1, u_in = j - (i - 1) <= 63 && i - 1 >= 1;
            const int e = l_in ? std::max(std::max(E[at(i, j - 1)] - kExt, H[at(i, j - 1)] - kOpen), 0) : 0;
            const int f = u_in ? std::max(std::max(F[at(i - 1, j)] - kExt, H[at(i - 1, j)] - kOpen), 0) : 0;
            const int s = (a[i - 1] & 3) == (b[j - 1] & 3) ? kMatch : kMismatch;
            const int h = std::max(std::max(H[at(i - 1, j - 1)] + s, 0), std::max(e, f));
            E[at(i, j)] = e; F[at(i, j)] = f; H[at(i, j)] = h;
            best = std::max(best, h);
        }
    return best;
}

template <typename K>
float time_kernel(K launch, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int k = 0; k < 3; ++k) launch();
    CHECK(hipEventRecord(e0));
    for (int k = 0; k < iters; ++k) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}
}  // namespace

int main()
{
    const int len = 1024;
    const size_t n_max = 65536;
    std::vector<uint8_t> a(n_max * len), b(n_max * len);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t i = 0; i < a.size(); ++i) {
        a[i] = uint8_t(rnd() & 3);
        b[i] = (rnd() % 100 < 8) ? uint8_t(rnd() & 3) : a[i];           // ~8 % substitutions: long alignments
    }
    uint8_t *d1, *d2;
    int32_t *ds;
    CHECK(hipMalloc(&d1, a.size())); CHECK(hipMalloc(&d2, b.size())); CHECK(hipMalloc(&ds, n_max * 4));
    CHECK(hipMemcpy(d1, a.data(), a.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d2, b.data(), b.size(), hipMemcpyHostToDevice));
    const size_t lds = 2 * (size_t)(len + 72) * 4;
    std::vector<int32_t> sa(n_max), sb(n_max);
    hipLaunchKernelGGL(one_wave, dim3(n_max), dim3(64), lds, 0, d1, d2, ds, len);
    CHECK(hipMemcpy(sa.data(), ds, n_max * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(two_waves, dim3(n_max), dim3(128), lds, 0, d1, d2, ds, len);
    CHECK(hipMemcpy(sb.data(), ds, n_max * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, bad = 0;
    for (size_t k = 0; k < n_max; ++k) diff += sa[k] != sb[k];
    for (size_t k = 0; k < 64; ++k) bad += sa[k] != host_gotoh(&a[k * len], &b[k * len], len);
    printf("scores: mapping A vs B differ on %zu of %zu alignments; A vs scalar host Gotoh differ on %zu of 64\n", diff, n_max, bad);
    if (diff || bad) return 1;
    for (size_t n : {size_t(1), size_t(256), size_t(1024), size_t(4096), size_t(16384), n_max}) {
        const int iters = n <= 4096 ? 50 : 10;
        const float ta = time_kernel([&] { hipLaunchKernelGGL(one_wave, dim3(n), dim3(64), lds, 0, d1, d2, ds, len); }, iters);
        const float tb = time_kernel([&] { hipLaunchKernelGGL(two_waves, dim3(n), dim3(128), lds, 0, d1, d2, ds, len); }, iters);
        printf("n %6zu x %d-mers, band 128:  A one wavefront per alignment %9.3f ms   B two wavefronts per alignment %9.3f ms   (B / A = %.2f)\n",
               n, len, ta, tb, tb / ta);
    }
    return 0;
}
