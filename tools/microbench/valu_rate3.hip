// Microbench 3: (a) do gfx950 16-bit VOP2 ops clear or preserve dst[31:16]?  (b) issue rate when dst != src0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_sem(unsigned* out) {
  unsigned a = 0xAAAA0005u, b = 0xBBBB0009u, d;
  d = 0xDDDDDDDDu; asm volatile("v_max_i16 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); out[0] = d;
  d = 0xDDDDDDDDu; asm volatile("v_sub_u16 %0, %1, %2" : "+v"(d) : "v"(b), "v"(a)); out[1] = d;
  d = 0xDDDDDDDDu; asm volatile("v_add_u16 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); out[2] = d;
  d = 0xDDDDDDDDu; asm volatile("v_sub_u16_e64 %0, %1, %2 clamp" : "+v"(d) : "v"(a), "v"(b)); out[3] = d;
  d = 0xDDDDDDDDu; asm volatile("v_max_u16 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b)); out[4] = d;
  unsigned n1 = 0x0000FFE2u /* -30 */, p = 0x00000014u;
  d = 0xDDDDDDDDu; asm volatile("v_max_i16 %0, %1, %2" : "+v"(d) : "v"(n1), "v"(p)); out[5] = d;
  unsigned n2 = 0xFFFFFFE2u;
  d = 0xDDDDDDDDu; asm volatile("v_max_i16 %0, %1, %2" : "+v"(d) : "v"(n2), "v"(p)); out[6] = d;
  d = 0xDDDDDDDDu; asm volatile("v_max_i16 %0, %1, %2" : "+v"(d) : "v"(n2), "v"(n1)); out[7] = d;
}

#define DEFK3(NAME, OP, SUFFIX)                                                      \
__global__ void __launch_bounds__(256) NAME(int* out, int iters, int seed) {         \
  int r0 = threadIdx.x ^ seed, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3,                \
      r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                            \
  for (int it = 0; it < iters; ++it) {                                               \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                  \
      asm volatile(OP " %0, %1, %3" SUFFIX "\n\t" OP " %1, %2, %4" SUFFIX "\n\t"     \
                   OP " %2, %3, %5" SUFFIX "\n\t" OP " %3, %4, %6" SUFFIX "\n\t"     \
                   OP " %4, %5, %7" SUFFIX "\n\t" OP " %5, %6, %0" SUFFIX "\n\t"     \
                   OP " %6, %7, %1" SUFFIX "\n\t" OP " %7, %0, %2" SUFFIX            \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); \
    }                                                                                \
  }                                                                                  \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
}
DEFK3(k3_max_i16, "v_max_i16", "")
DEFK3(k3_max_i32, "v_max_i32", "")
DEFK3(k3_add_u32, "v_add_u32", "")
DEFK3(k3_sub_u32c, "v_sub_u32_e64", " clamp")
DEFK3(k3_sub_u16c, "v_sub_u16_e64", " clamp")
DEFK3(k3_add_u16, "v_add_u16", "")
DEFK3(k3_and, "v_and_b32", "")
DEFK3(k3_dot4c, "v_dot4c_i32_i8", "")
DEFK3(k3_pkmax, "v_pk_max_i16", "")

// mixed body like the SW cell, all asm, distinct registers, to see real mix rates
__global__ void __launch_bounds__(256) k_mix16(int* out, int iters, int seed) {
  int h0 = threadIdx.x & 15, h1 = h0 + 1, h2 = h0 + 2, h3 = h0 + 3, best = 0, up = 1, d = 0;
  int p0 = seed * 0x01020304, p1 = p0 + 1, p2 = p0 + 2, p3 = p0 + 3, oh = 1 << (8 * (threadIdx.x & 3)), gap = seed & 3;
  for (int it = 0; it < iters; ++it) {
#define CELL16(H, P) \
    asm volatile("v_dot4c_i32_i8 %1, %5, %6\n\t v_max_i16 %3, %0, %2\n\t v_sub_u16_e64 %3, %3, %7 clamp\n\t v_max_i16 %3, %3, %1\n\t" \
                 "v_max_i16 %4, %4, %3\n\t v_mov_b32 %1, %0\n\t v_mov_b32 %0, %3\n\t v_mov_b32 %2, %3" \
                 : "+v"(H), "+v"(d), "+v"(up), "=&v"(t), "+v"(best) : "v"(P), "v"(oh), "v"(gap));
    int t;
    CELL16(h0, p0) CELL16(h1, p1) CELL16(h2, p2) CELL16(h3, p3)
    CELL16(h0, p1) CELL16(h1, p2) CELL16(h2, p3) CELL16(h3, p0)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = h0 + h1 + h2 + h3 + best + up + d;
}
__global__ void __launch_bounds__(256) k_mix32(int* out, int iters, int seed) {
  int h0 = threadIdx.x & 15, h1 = h0 + 1, h2 = h0 + 2, h3 = h0 + 3, best = 0, up = 1, d = 0;
  int p0 = seed * 0x01020304, p1 = p0 + 1, p2 = p0 + 2, p3 = p0 + 3, oh = 1 << (8 * (threadIdx.x & 3)), gap = seed & 3;
  for (int it = 0; it < iters; ++it) {
#define CELL32(H, P) \
    asm volatile("v_dot4c_i32_i8 %1, %5, %6\n\t v_max3_i32 %3, %0, %2, %1\n\t v_max_i32 %4, %4, %3\n\t v_sub_u32_e64 %3, %3, %7 clamp\n\t" \
                 "v_mov_b32 %1, %0\n\t v_mov_b32 %0, %3\n\t v_mov_b32 %2, %3" \
                 : "+v"(H), "+v"(d), "+v"(up), "=&v"(t), "+v"(best) : "v"(P), "v"(oh), "v"(gap));
    int t;
    CELL32(h0, p0) CELL32(h1, p1) CELL32(h2, p2) CELL32(h3, p3)
    CELL32(h0, p1) CELL32(h1, p2) CELL32(h2, p3) CELL32(h3, p0)
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = h0 + h1 + h2 + h3 + best + up + d;
}

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
  unsigned* dsem; CHECK(hipMalloc(&dsem, 64)); unsigned hsem[8];
  hipLaunchKernelGGL(k_sem, dim3(1), dim3(1), 0, 0, dsem); CHECK(hipMemcpy(hsem, dsem, 32, hipMemcpyDeviceToHost));
  const char* names[8] = {"max_i16(AAAA0005,BBBB0009)", "sub_u16(BBBB0009-AAAA0005)", "add_u16", "sub_u16 clamp(5-9)", "max_u16", "max_i16(0000FFE2,14)", "max_i16(FFFFFFE2,14)", "max_i16(FFFFFFE2,0000FFE2)"};
  for (int i = 0; i < 8; ++i) printf("%-30s dst(init DDDDDDDD) = %08X\n", names[i], hsem[i]);
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  int* dout; CHECK(hipMalloc(&dout, sizeof(int) * 256 * cus * 16));
  struct { const char* name; kern_t k; } ks[] = {{"max_i16 3addr", k3_max_i16}, {"max_i32 3addr", k3_max_i32}, {"add_u32 3addr", k3_add_u32},
    {"sub_u32 clamp 3addr", k3_sub_u32c}, {"sub_u16 clamp 3addr", k3_sub_u16c}, {"add_u16 3addr", k3_add_u16}, {"and_b32 3addr", k3_and},
    {"dot4c 3addr", k3_dot4c}, {"pk_max_i16 3addr", k3_pkmax}};
  for (int wps : {2, 4, 8}) {
    int blocks = cus * wps; printf("--- waves/SIMD = %d\n", wps);
    for (auto& e : ks) {
      double s = run(e.k, blocks, 4096, dout);
      double winstr = (double)blocks * 4 * 4096 * 64;
      printf("%-22s %8.3f ms  %.2f clk/instr\n", e.name, s * 1e3, (cus * 4.0) * clk * s / winstr);
    }
    double s16 = run(k_mix16, blocks, 4096, dout), s32 = run(k_mix32, blocks, 4096, dout);
    double cells = (double)blocks * 4 * 4096 * 8;
    printf("mix16 (8 instr/cell) %.3f ms  %.2f clk/cell   mix32 (7 instr/cell) %.3f ms %.2f clk/cell\n", s16 * 1e3,
           (cus * 4.0) * clk * s16 / cells, s32 * 1e3, (cus * 4.0) * clk * s32 / cells);
  }
  return 0;
}
