// Microbench 4: do VGPR bank conflicts (3 sources in the same bank, bank = vgpr % 4) slow v_max3_i32 / v_dot4_i32_i8?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// explicit registers: dst rotates over v[16..23]; sources chosen by the macro arguments
#define BODY(OP, S0, S1, S2) \
  OP " v16, " S0 ", " S1 ", " S2 "\n\t" OP " v17, " S0 ", " S1 ", " S2 "\n\t" OP " v18, " S0 ", " S1 ", " S2 "\n\t" OP " v19, " S0 ", " S1 ", " S2 "\n\t" \
  OP " v20, " S0 ", " S1 ", " S2 "\n\t" OP " v21, " S0 ", " S1 ", " S2 "\n\t" OP " v22, " S0 ", " S1 ", " S2 "\n\t" OP " v23, " S0 ", " S1 ", " S2 "\n\t"

#define DEFK(NAME, OP, S0, S1, S2)                                                   \
__global__ void __launch_bounds__(256) NAME(int* out, int iters, int seed) {         \
  asm volatile("v_mov_b32 v0, %0\n\t v_mov_b32 v1, %0\n\t v_mov_b32 v2, %0\n\t v_mov_b32 v3, %0\n\t v_mov_b32 v4, %0\n\t v_mov_b32 v5, %0\n\t" \
               "v_mov_b32 v6, %0\n\t v_mov_b32 v7, %0\n\t v_mov_b32 v8, %0\n\t v_mov_b32 v9, %0\n\t v_mov_b32 v10, %0\n\t v_mov_b32 v11, %0\n\t v_mov_b32 v12, %0" \
               :: "v"(seed + (int)threadIdx.x) : "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12"); \
  for (int it = 0; it < iters; ++it) {                                               \
    asm volatile(BODY(OP, S0, S1, S2) BODY(OP, S0, S1, S2) BODY(OP, S0, S1, S2) BODY(OP, S0, S1, S2) \
                 BODY(OP, S0, S1, S2) BODY(OP, S0, S1, S2) BODY(OP, S0, S1, S2) BODY(OP, S0, S1, S2) \
                 ::: "v16","v17","v18","v19","v20","v21","v22","v23");               \
  }                                                                                  \
  int r; asm volatile("v_add_u32 %0, v16, v23" : "=v"(r));                          \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                    \
}
DEFK(k_max3_diff, "v_max3_i32", "v1", "v2", "v3")
DEFK(k_max3_same, "v_max3_i32", "v0", "v4", "v8")
DEFK(k_max3_two,  "v_max3_i32", "v0", "v4", "v1")
DEFK(k_dot4_diff, "v_dot4_i32_i8", "v1", "v2", "v3")
DEFK(k_dot4_same, "v_dot4_i32_i8", "v0", "v4", "v8")
DEFK(k_add3_same, "v_add3_u32", "v0", "v4", "v8")

typedef void (*kern_t)(int*, int, int);
static double run(kern_t k, int blocks, int iters, int* dout) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, iters / 8, 1); CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dout, iters, 1);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  return best * 1e-3;
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  int* dout; CHECK(hipMalloc(&dout, sizeof(int) * 256 * cus * 16));
  struct { const char* name; kern_t k; } ks[] = {{"max3 v1,v2,v3 (3 banks)", k_max3_diff}, {"max3 v0,v4,v8 (1 bank)", k_max3_same},
    {"max3 v0,v4,v1 (2 banks)", k_max3_two}, {"dot4 v1,v2,v3", k_dot4_diff}, {"dot4 v0,v4,v8", k_dot4_same}, {"add3 v0,v4,v8", k_add3_same}};
  for (int wps : {4, 8}) {
    int blocks = cus * wps; printf("--- waves/SIMD = %d\n", wps);
    for (auto& e : ks) {
      double s = run(e.k, blocks, 4096, dout);
      double winstr = (double)blocks * 4 * 4096 * 64;
      printf("%-26s %8.3f ms  %.2f clk/instr\n", e.name, s * 1e3, (cus * 4.0) * clk * s / winstr);
    }
  }
  return 0;
}
