// Microbench 5: how many per-lane byte loads (global_load_sbyte from a small per-wave table that lives in L1/L2)
// can the vector-memory path of a CU sustain per cycle, alone and beside a saturating VALU stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// table per wave: 5 planes x 16 rows x 64 bytes = 5120 B
template <int VALU_PER_LOAD>
__global__ void __launch_bounds__(256) k_lut(const signed char* __restrict__ tables, int* out, int iters, int seed) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const signed char* t = tables + wave * 5120;
  int acc0 = seed, acc1 = seed + 1, acc2 = lane, acc3 = 3;
  unsigned plane = (lane * 7 + seed) % 5;
  for (int it = 0; it < iters; ++it) {
    int s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = t[plane * 1024 + i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc0 += s[i];
#pragma unroll
      for (int v = 0; v < VALU_PER_LOAD; ++v) {   // half-rate filler (v_max3-like)
        asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(acc1) : "v"(acc2), "v"(acc3));
      }
    }
    plane = (plane + 1 + (acc0 & 1)) % 5;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1;
}
typedef void (*kern_t)(const signed char*, int*, int, int);
static double run(kern_t k, int blocks, int iters, const signed char* tab, int* dout) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, dout, iters / 8, 1); CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, tab, dout, iters, 1);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best * 1e-3;
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  const int maxblocks = cus * 8;
  signed char* tab; CHECK(hipMalloc(&tab, (size_t)maxblocks * 4 * 5120)); CHECK(hipMemset(tab, 1, (size_t)maxblocks * 4 * 5120));
  int* dout; CHECK(hipMalloc(&dout, sizeof(int) * 256 * maxblocks));
  struct { const char* name; kern_t k; int valu; } ks[] = {{"loads only", k_lut<0>, 0}, {"1 max3 per load", k_lut<1>, 1}, {"2 max3 per load", k_lut<2>, 2}, {"3 max3 per load", k_lut<3>, 3}, {"5 max3 per load", k_lut<5>, 5}};
  for (int wps : {2, 4, 8}) {
    int blocks = cus * wps;
    printf("--- waves/SIMD = %d (per-CU table footprint %d KB)\n", wps, wps * 4 * 5);
    for (auto& e : ks) {
      double s = run(e.k, blocks, 512, tab, dout);
      double loads = (double)blocks * 4 * 512 * 16;   // wave-level load instructions
      printf("%-18s %8.3f ms  %.2f clk per load-instr per CU   (VALU alone would need %.1f)\n", e.name, s * 1e3,
             cus * clk * s / loads, e.valu * 4.1 / 4.0);
    }
  }
  return 0;
}
