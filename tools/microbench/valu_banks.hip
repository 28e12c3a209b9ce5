// Microbench 8: do VGPR BANKS matter?  The same instruction with its source registers in different banks (register
// number mod 4) and with two / three of them in the same bank.
// Same method as valu_rate*.hip: 64 independent copies per iteration, destinations rotate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define R8(A, B, C, D, E, F, G, H) A "\n\t" B "\n\t" C "\n\t" D "\n\t" E "\n\t" F "\n\t" G "\n\t" H "\n\t"
#define X8(S) S S S S S S S S

#define DEFK(NAME, BODY8, CLOBBERS...)                                               \
__global__ void __launch_bounds__(256) NAME(int* out, int iters, int seed) {         \
  asm volatile("v_mov_b32 v0, %0\n\t v_mov_b32 v1, %0\n\t v_mov_b32 v2, %0\n\t v_mov_b32 v3, %0\n\t v_mov_b32 v4, %0\n\t v_mov_b32 v5, %0\n\t" \
               "v_mov_b32 v6, %0\n\t v_mov_b32 v7, %0\n\t s_mov_b64 s[56:57], -1\n\t s_mov_b64 vcc, -1"                     \
               :: "v"(seed + (int)threadIdx.x) : "v0","v1","v2","v3","v4","v5","v6","v7","s56","s57","vcc");                \
  for (int it = 0; it < iters; ++it) {                                               \
    asm volatile(X8(BODY8) ::: "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
                 "s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51","s52","s53","s54","s55", CLOBBERS);     \
  }                                                                                  \
  int r; asm volatile("v_add_u32 %0, v16, v23" : "=v"(r));                          \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                    \
}

#define V1(OP, SRC)  R8(OP " v16, " SRC, OP " v17, " SRC, OP " v18, " SRC, OP " v19, " SRC, OP " v20, " SRC, OP " v21, " SRC, OP " v22, " SRC, OP " v23, " SRC)
#define V64(OP, SRC) R8(OP " v[16:17], " SRC, OP " v[18:19], " SRC, OP " v[20:21], " SRC, OP " v[22:23], " SRC, OP " v[24:25], " SRC, OP " v[26:27], " SRC, OP " v[28:29], " SRC, OP " v[30:31], " SRC)
#define S64(OP, SRC) R8(OP " s[40:41], " SRC, OP " s[42:43], " SRC, OP " s[44:45], " SRC, OP " s[46:47], " SRC, OP " s[48:49], " SRC, OP " s[50:51], " SRC, OP " s[52:53], " SRC, OP " s[54:55], " SRC)
#define S32(OP, SRC) R8(OP " s40, " SRC, OP " s41, " SRC, OP " s42, " SRC, OP " s43, " SRC, OP " s44, " SRC, OP " s45, " SRC, OP " s46, " SRC, OP " s47, " SRC)
#define VCC8(OP, SRC) R8(OP " vcc, " SRC, OP " vcc, " SRC, OP " vcc, " SRC, OP " vcc, " SRC, OP " vcc, " SRC, OP " vcc, " SRC, OP " vcc, " SRC, OP " vcc, " SRC)



DEFK(k_max3_banks_123, V1("v_pk_maximum3_f16", "v1, v2, v3"), "memory")
DEFK(k_max3_banks_115, V1("v_pk_maximum3_f16", "v1, v5, v3"), "memory")
DEFK(k_max3_banks_111, V1("v_pk_maximum3_f16", "v1, v5, v9"), "memory")
DEFK(k_perm_banks_123, V1("v_perm_b32", "v1, v2, v3"), "memory")
DEFK(k_perm_banks_111, V1("v_perm_b32", "v1, v5, v9"), "memory")
DEFK(k_add_banks_12,   V1("v_add_u32", "v1, v2"), "memory")
DEFK(k_add_banks_11,   V1("v_add_u32", "v1, v5"), "memory")
DEFK(k_pksub_banks_12, V1("v_pk_sub_u16", "v1, v2 clamp"), "memory")
DEFK(k_pksub_banks_11, V1("v_pk_sub_u16", "v1, v5 clamp"), "memory")
DEFK(k_max3i_banks_123, V1("v_max3_i32", "v1, v2, v3"), "memory")
DEFK(k_max3i_banks_111, V1("v_max3_i32", "v1, v5, v9"), "memory")

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
  struct { const char* name; kern_t k; } ks[] = {
    {"v_pk_maximum3_f16 banks 1,2,3", k_max3_banks_123}, {"v_pk_maximum3_f16 banks 1,1,3", k_max3_banks_115}, {"v_pk_maximum3_f16 banks 1,1,1", k_max3_banks_111},
    {"v_perm_b32 banks 1,2,3", k_perm_banks_123}, {"v_perm_b32 banks 1,1,1", k_perm_banks_111},
    {"v_add_u32 banks 1,2", k_add_banks_12}, {"v_add_u32 banks 1,1", k_add_banks_11},
    {"v_pk_sub_u16 banks 1,2", k_pksub_banks_12}, {"v_pk_sub_u16 banks 1,1", k_pksub_banks_11},
    {"v_max3_i32 banks 1,2,3", k_max3i_banks_123}, {"v_max3_i32 banks 1,1,1", k_max3i_banks_111}};
  for (int wps : {1, 2, 4}) {
    int blocks = cus * wps; printf("--- waves/SIMD = %d\n", wps);
    for (auto& e : ks) {
      double s = run(e.k, blocks, 2048, dout);
      double winstr = (double)blocks * 4 * 2048 * 64;
      printf("%-32s %8.3f ms  %.2f clk/instr\n", e.name, s * 1e3, (cus * 4.0) * clk * s / winstr);
    }
  }
  return 0;
}
