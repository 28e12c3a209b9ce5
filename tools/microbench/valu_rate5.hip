// Microbench 5: issue cost of the VALU instructions tools/isa_census.py used to price by default ("unmeasured"): compares
// (VCC and SGPR destinations), 64-bit shifts / adds / moves, cross-lane reads, bit counts, and a few candidates looked at
// for DESIGN.md section 5.1.  Same method as valu_rate*.hip: 64 independent copies per iteration, destinations rotate.
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

DEFK(k_cmp_vcc,      VCC8("v_cmp_lt_i32", "v1, v2"), "vcc")
DEFK(k_cmp_sgpr,     S64("v_cmp_lt_i32", "v1, v2"), "memory")
DEFK(k_cmp_ne_sgpr,  S64("v_cmp_ne_u32", "v1, v2"), "memory")
DEFK(k_cndmask_vcc,  V1("v_cndmask_b32", "v1, v2, vcc"), "memory")
DEFK(k_cndmask_sgpr, V1("v_cndmask_b32", "v1, v2, s[56:57]"), "memory")
DEFK(k_lshl_b64,     V64("v_lshlrev_b64", "4, v[2:3]"), "memory")
DEFK(k_lshr_b64,     V64("v_lshrrev_b64", "4, v[2:3]"), "memory")
DEFK(k_lshl_add_u64, V64("v_lshl_add_u64", "v[2:3], 0, v[4:5]"), "memory")
DEFK(k_mov_b64,      V64("v_mov_b64", "v[2:3]"), "memory")
DEFK(k_readfirstlane, S32("v_readfirstlane_b32", "v1"), "memory")
DEFK(k_readlane,     S32("v_readlane_b32", "v1, 3"), "memory")
DEFK(k_bcnt,         V1("v_bcnt_u32_b32", "v1, v2"), "memory")
DEFK(k_not,          V1("v_not_b32", "v1"), "memory")
DEFK(k_ffbh,         V1("v_ffbh_u32", "v1"), "memory")
DEFK(k_bfm,          V1("v_bfm_b32", "v1, v2"), "memory")
DEFK(k_max3_u32,     V1("v_max3_u32", "v1, v2, v3"), "memory")
DEFK(k_sub_dpp,      V1("v_sub_u32_dpp", "v1, v2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), "memory")
DEFK(k_add_sdwa,     V1("v_add_u32_sdwa", "v1, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1"), "memory")
DEFK(k_or3,          V1("v_or3_b32", "v1, v2, v3"), "memory")
DEFK(k_bitop3,       V1("v_bitop3_b32", "v1, v2, v3 bitop3:0xca"), "memory")
DEFK(k_alignbit,     V1("v_alignbit_b32", "v1, v2, 2"), "memory")
DEFK(k_bfe_u32,      V1("v_bfe_u32", "v1, 4, 2"), "memory")
DEFK(k_dot2_i16,     V1("v_dot2_i32_i16", "v1, v2, v3"), "memory")
DEFK(k_dot8_i4,      V1("v_dot8_i32_i4", "v1, v2, v3"), "memory")
DEFK(k_pk_max3_f16,  V1("v_pk_maximum3_f16", "v1, v2, v3"), "memory")
DEFK(k_pk_max_f16,   V1("v_pk_max_f16", "v1, v2"), "memory")
DEFK(k_max3_f16,     V1("v_max3_f16", "v1, v2, v3"), "memory")

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
    {"v_cmp_lt_i32 -> vcc", k_cmp_vcc}, {"v_cmp_lt_i32 -> sgpr pair", k_cmp_sgpr}, {"v_cmp_ne_u32 -> sgpr pair", k_cmp_ne_sgpr},
    {"v_cndmask_b32 (vcc)", k_cndmask_vcc}, {"v_cndmask_b32 (sgpr pair)", k_cndmask_sgpr},
    {"v_lshlrev_b64", k_lshl_b64}, {"v_lshrrev_b64", k_lshr_b64}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_mov_b64", k_mov_b64},
    {"v_readfirstlane_b32", k_readfirstlane}, {"v_readlane_b32", k_readlane}, {"v_bcnt_u32_b32", k_bcnt}, {"v_not_b32", k_not},
    {"v_ffbh_u32", k_ffbh}, {"v_bfm_b32", k_bfm}, {"v_max3_u32", k_max3_u32}, {"v_sub_u32_dpp row_shr", k_sub_dpp},
    {"v_add_u32_sdwa BYTE_1", k_add_sdwa}, {"v_or3_b32", k_or3}, {"v_bitop3_b32", k_bitop3}, {"v_alignbit_b32", k_alignbit},
    {"v_bfe_u32", k_bfe_u32}, {"v_dot2_i32_i16", k_dot2_i16}, {"v_dot8_i32_i4", k_dot8_i4}, {"v_pk_maximum3_f16", k_pk_max3_f16},
    {"v_pk_max_f16", k_pk_max_f16}, {"v_max3_f16", k_max3_f16}};
  for (int wps : {1, 2, 4, 8}) {
    int blocks = cus * wps; printf("--- waves/SIMD = %d\n", wps);
    for (auto& e : ks) {
      double s = run(e.k, blocks, 2048, dout);
      double winstr = (double)blocks * 4 * 2048 * 64;
      printf("%-28s %8.3f ms  %.2f clk/instr\n", e.name, s * 1e3, (cus * 4.0) * clk * s / winstr);
    }
  }
  return 0;
}
