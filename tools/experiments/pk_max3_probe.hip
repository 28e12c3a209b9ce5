// Probe: does v_pk_maximum3_f16 applied to NON-NEGATIVE 16-bit INTEGER bit patterns (two per dword) return the integer
// maximum?  (Positive IEEE half-precision patterns order like integers; values below 1024 are fp16 denormals, so the
// answer depends on the denormal mode the kernel runs in; patterns >= 0x7C00 are Inf / NaN and must not occur.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(const uint32_t *a, const uint32_t *b, const uint32_t *c, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    out[i] = r;
}
int main()
{
    const int n = 1 << 20;
    std::vector<uint32_t> a(n), b(n), c(n), o(n);
    uint64_t x = 12345;
    auto rnd = [&x]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (uint32_t)(x >> 20); };
    auto val = [&](int k) { const uint32_t r = rnd(); return k % 3 == 0 ? r % 1100u : k % 3 == 1 ? r % 31744u : r % 300u; };   // small (denormal), any, tiny
    for (int i = 0; i < n; ++i) { a[i] = val(i) | val(i + 1) << 16; b[i] = val(i + 2) | val(i) << 16; c[i] = val(i + 1) | val(i + 2) << 16; }
    a[0] = 0; b[0] = 0; c[0] = 0; a[1] = 1; b[1] = 0; c[1] = 0; a[2] = 0x7BFF7BFFu; b[2] = 1023u | 1024u << 16; c[2] = 0;
    uint32_t *da, *db, *dc, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int i = 0; i < n; ++i) {
        auto m3 = [](uint32_t p, uint32_t q, uint32_t r) { return p > q ? (p > r ? p : r) : (q > r ? q : r); };
        const uint32_t want = m3(a[i] & 0xffff, b[i] & 0xffff, c[i] & 0xffff) | m3(a[i] >> 16, b[i] >> 16, c[i] >> 16) << 16;
        if (o[i] != want) { if (bad < 5) printf("mismatch at %d: %08x %08x %08x -> %08x, want %08x\n", i, a[i], b[i], c[i], o[i], want); ++bad; }
    }
    printf("v_pk_maximum3_f16 on non-negative integer patterns: %zu of %d mismatches\n", bad, n);
    return bad ? 1 : 0;
}
