// Instruction-issue microbenchmark for gfx950 integer VALU ops used by the SW kernels.
// Each kernel runs ITERS x 64 copies of one instruction over 8 independent registers.
// Reports wave-instructions per clock per SIMD (peak expected: 0.5 = one wave64 op / 2 clk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define REP8(S, a,b,c) \
  S(0,a,b,c) S(1,a,b,c) S(2,a,b,c) S(3,a,b,c) S(4,a,b,c) S(5,a,b,c) S(6,a,b,c) S(7,a,b,c)

#define DEFK(NAME, ASM3)                                                            \
__global__ void __launch_bounds__(256) NAME(int* out, int iters, int seed) {        \
  int r0 = threadIdx.x ^ seed, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3,               \
      r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                           \
  int a = seed * 3 + 1, b = seed | 0x01010101;                                      \
  for (int it = 0; it < iters; ++it) {                                              \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                 \
      asm volatile(ASM3("%0") "\n\t" ASM3("%1") "\n\t" ASM3("%2") "\n\t" ASM3("%3") "\n\t" \
                   ASM3("%4") "\n\t" ASM3("%5") "\n\t" ASM3("%6") "\n\t" ASM3("%7")        \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                   : "v"(a), "v"(b));                                               \
    }                                                                               \
  }                                                                                 \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; \
}

#define B_MOV(R) "v_mov_b32 " R ", %8"
#define B_MAX_U32(R) "v_max_u32 " R ", " R ", %8"
#define B_MIN_I32(R) "v_min_i32 " R ", " R ", %8"
#define B_MAX_F32(R) "v_max_f32 " R ", " R ", %8"
#define B_ADD_F32(R) "v_add_f32 " R ", " R ", %8"
#define B_MUL_F32(R) "v_mul_f32 " R ", " R ", %8"
#define B_MAX3_F32(R) "v_max3_f32 " R ", " R ", %8, %9"
#define B_MAX_U16(R) "v_max_u16 " R ", " R ", %8"
#define B_MIN_I16(R) "v_min_i16 " R ", " R ", %8"
#define B_ADD_U16(R) "v_add_u16 " R ", " R ", %8"
#define B_SUB_U16(R) "v_sub_u16 " R ", " R ", %8"
#define B_SUB_U16_CLAMP(R) "v_sub_u16_e64 " R ", " R ", %8 clamp"
#define B_ADD_I16_VOP3(R) "v_add_i16 " R ", " R ", %8"
#define B_MAX_I16_E64(R) "v_max_i16_e64 " R ", " R ", %8"
#define B_MAD_I16(R) "v_mad_i16 " R ", " R ", %8, %9"
#define B_MAD_LEGACY_U16(R) "v_mad_legacy_u16 " R ", " R ", %8, %9"
#define B_AND_B32(R) "v_and_b32 " R ", " R ", %8"
#define B_OR_B32(R) "v_or_b32 " R ", " R ", %8"
#define B_XOR_B32(R) "v_xor_b32 " R ", " R ", %8"
#define B_LSHLREV_B32(R) "v_lshlrev_b32 " R ", 1, " R ""
#define B_LSHLREV_B32_V(R) "v_lshlrev_b32 " R ", %8, " R ""
#define B_ASHRREV_I32(R) "v_ashrrev_i32 " R ", 1, " R ""
#define B_LSHRREV_B32_V(R) "v_lshrrev_b32 " R ", %8, " R ""
#define B_ADD_CO_U32(R) "v_add_co_u32 " R ", vcc, " R ", %8"
#define B_ADD_U32_E64(R) "v_add_u32_e64 " R ", " R ", %8"
#define B_SUB_I32_CLAMP(R) "v_sub_i32 " R ", " R ", %8 clamp"
#define B_ADD_I32_CLAMP(R) "v_add_i32 " R ", " R ", %8 clamp"
#define B_CNDMASK_SGPR(R) "v_cndmask_b32_e64 " R ", " R ", %8, s[10:11]"
#define B_MAX_F16(R) "v_max_f16 " R ", " R ", %8"
#define B_ADD_F16(R) "v_add_f16 " R ", " R ", %8"
#define B_FMAC_F32(R) "v_fmac_f32 " R ", %8, %9"
#define B_MUL_U32_U24(R) "v_mul_u32_u24 " R ", " R ", %8"
#define B_LSHL_OR_B32(R) "v_lshl_or_b32 " R ", " R ", 1, %8"
#define B_AND_OR_B32(R) "v_and_or_b32 " R ", " R ", %8, %9"
#define B_ALIGNBIT(R) "v_alignbit_b32 " R ", " R ", %8, 8"
#define B_SAD_U8(R) "v_sad_u8 " R ", " R ", %8, %9"
#define B_BFI(R) "v_bfi_b32 " R ", " R ", %8, %9"
#define B_MIN3_I32(R) "v_min3_i32 " R ", " R ", %8, %9"
#define B_MAX3_I16(R) "v_max3_i16 " R ", " R ", %8, %9"
#define B_MAX3_U16(R) "v_max3_u16 " R ", " R ", %8, %9"
#define B_MED3_I16(R) "v_med3_i16 " R ", " R ", %8, %9"
#define B_SUBREV_U32(R) "v_subrev_u32 " R ", " R ", %8"
#define B_XAD_U32(R) "v_xad_u32 " R ", " R ", %8, %9"
#define B_BITOP3(R) "v_bitop3_b32 " R ", " R ", %8, %9 bitop3:0x96"
DEFK(k_mov, B_MOV)
DEFK(k_max_u32, B_MAX_U32)
DEFK(k_min_i32, B_MIN_I32)
DEFK(k_max_f32, B_MAX_F32)
DEFK(k_add_f32, B_ADD_F32)
DEFK(k_mul_f32, B_MUL_F32)
DEFK(k_max3_f32, B_MAX3_F32)
DEFK(k_max_u16, B_MAX_U16)
DEFK(k_min_i16, B_MIN_I16)
DEFK(k_add_u16, B_ADD_U16)
DEFK(k_sub_u16, B_SUB_U16)
DEFK(k_sub_u16_clamp, B_SUB_U16_CLAMP)
DEFK(k_add_i16_vop3, B_ADD_I16_VOP3)
DEFK(k_max_i16_e64, B_MAX_I16_E64)
DEFK(k_mad_i16, B_MAD_I16)
DEFK(k_mad_legacy_u16, B_MAD_LEGACY_U16)
DEFK(k_and_b32, B_AND_B32)
DEFK(k_or_b32, B_OR_B32)
DEFK(k_xor_b32, B_XOR_B32)
DEFK(k_lshlrev_b32, B_LSHLREV_B32)
DEFK(k_lshlrev_b32_v, B_LSHLREV_B32_V)
DEFK(k_ashrrev_i32, B_ASHRREV_I32)
DEFK(k_lshrrev_b32_v, B_LSHRREV_B32_V)
DEFK(k_add_co_u32, B_ADD_CO_U32)
DEFK(k_add_u32_e64, B_ADD_U32_E64)
DEFK(k_sub_i32_clamp, B_SUB_I32_CLAMP)
DEFK(k_add_i32_clamp, B_ADD_I32_CLAMP)
DEFK(k_cndmask_sgpr, B_CNDMASK_SGPR)
DEFK(k_max_f16, B_MAX_F16)
DEFK(k_add_f16, B_ADD_F16)
DEFK(k_fmac_f32, B_FMAC_F32)
DEFK(k_mul_u32_u24, B_MUL_U32_U24)
DEFK(k_lshl_or_b32, B_LSHL_OR_B32)
DEFK(k_and_or_b32, B_AND_OR_B32)
DEFK(k_alignbit, B_ALIGNBIT)
DEFK(k_sad_u8, B_SAD_U8)
DEFK(k_bfi, B_BFI)
DEFK(k_min3_i32, B_MIN3_I32)
DEFK(k_max3_i16, B_MAX3_I16)
DEFK(k_max3_u16, B_MAX3_U16)
DEFK(k_med3_i16, B_MED3_I16)
DEFK(k_subrev_u32, B_SUBREV_U32)
DEFK(k_xad_u32, B_XAD_U32)
DEFK(k_bitop3, B_BITOP3)

typedef void (*kern_t)(int*, int, int);
static double run(kern_t k, int blocks, int threads, int iters, int* dout) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, iters / 8, 1);
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, dout, iters, 1);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  return best * 1e-3;
}
int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  int* dout; CHECK(hipMalloc(&dout, sizeof(int) * 256 * cus * 16));
  struct { const char* name; kern_t k; } ks[] = {
    {"mov", k_mov},
    {"max_u32", k_max_u32},
    {"min_i32", k_min_i32},
    {"max_f32", k_max_f32},
    {"add_f32", k_add_f32},
    {"mul_f32", k_mul_f32},
    {"max3_f32", k_max3_f32},
    {"max_u16", k_max_u16},
    {"min_i16", k_min_i16},
    {"add_u16", k_add_u16},
    {"sub_u16", k_sub_u16},
    {"sub_u16_clamp", k_sub_u16_clamp},
    {"add_i16_vop3", k_add_i16_vop3},
    {"max_i16_e64", k_max_i16_e64},
    {"mad_i16", k_mad_i16},
    {"mad_legacy_u16", k_mad_legacy_u16},
    {"and_b32", k_and_b32},
    {"or_b32", k_or_b32},
    {"xor_b32", k_xor_b32},
    {"lshlrev_b32", k_lshlrev_b32},
    {"lshlrev_b32_v", k_lshlrev_b32_v},
    {"ashrrev_i32", k_ashrrev_i32},
    {"lshrrev_b32_v", k_lshrrev_b32_v},
    {"add_co_u32", k_add_co_u32},
    {"add_u32_e64", k_add_u32_e64},
    {"sub_i32_clamp", k_sub_i32_clamp},
    {"add_i32_clamp", k_add_i32_clamp},
    {"cndmask_sgpr", k_cndmask_sgpr},
    {"max_f16", k_max_f16},
    {"add_f16", k_add_f16},
    {"fmac_f32", k_fmac_f32},
    {"mul_u32_u24", k_mul_u32_u24},
    {"lshl_or_b32", k_lshl_or_b32},
    {"and_or_b32", k_and_or_b32},
    {"alignbit", k_alignbit},
    {"sad_u8", k_sad_u8},
    {"bfi", k_bfi},
    {"min3_i32", k_min3_i32},
    {"max3_i16", k_max3_i16},
    {"max3_u16", k_max3_u16},
    {"med3_i16", k_med3_i16},
    {"subrev_u32", k_subrev_u32},
    {"xad_u32", k_xad_u32},
    {"bitop3", k_bitop3},
  };
  const int iters = 4096;
  for (int wps : {2, 8}) {
    int blocks = cus * wps;
    printf("--- waves/SIMD = %d\n", wps);
    for (auto& e : ks) {
      double s = run(e.k, blocks, 256, iters, dout);
      double winstr = (double)blocks * 4 * iters * 64;
      double r = winstr / s / (cus * 4.0) / clk;
      printf("%-18s %8.3f ms  %.2f clk/instr\n", e.name, s * 1e3, 1.0 / r);
    }
  }
  return 0;
}
