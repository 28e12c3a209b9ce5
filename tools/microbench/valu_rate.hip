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

#define A_MAX(R)      "v_max_i32 " R ", " R ", %8"
#define A_ADD(R)      "v_add_u32 " R ", " R ", %8"
#define A_MAX3(R)     "v_max3_i32 " R ", " R ", %8, %9"
#define A_SUBCL(R)    "v_sub_u32_e64 " R ", " R ", %8 clamp"
#define A_DOT4C(R)    "v_dot4c_i32_i8 " R ", %8, %9"
#define A_DOT4(R)     "v_dot4_i32_i8 " R ", %8, %9, " R
#define A_PERM(R)     "v_perm_b32 " R ", " R ", %8, %9"
#define A_BFE(R)      "v_bfe_i32 " R ", " R ", %8, 8"
#define A_ADD3(R)     "v_add3_u32 " R ", " R ", %8, %9"
#define A_PKADD(R)    "v_pk_add_u16 " R ", " R ", %8"
#define A_PKMAX(R)    "v_pk_max_i16 " R ", " R ", %8"
#define A_PKMAXU(R)   "v_pk_max_u16 " R ", " R ", %8"
#define A_PKSUBCL(R)  "v_pk_sub_u16 " R ", " R ", %8 clamp"
#define A_PKMAD(R)    "v_pk_mad_i16 " R ", " R ", %8, %9"
#define A_MOVDPP_R(R) "v_mov_b32_dpp " R ", " R " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define A_MOVDPP_W(R) "v_mov_b32_dpp " R ", " R " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define A_ANDDPP(R)   "v_and_b32_dpp " R ", " R ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define A_MAXDPP(R)   "v_max_i32_dpp " R ", " R ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define A_DOT4CDPP(R) "v_dot4c_i32_i8_dpp " R ", %8, %9 row_shr:1 row_mask:0xf bank_mask:0xf"
#define A_MAD24(R)    "v_mad_i32_i24 " R ", " R ", %8, %9"
#define A_CNDMASK(R)  "v_cndmask_b32 " R ", " R ", %8, vcc"
#define A_MED3(R)     "v_med3_i32 " R ", " R ", %8, %9"
#define A_LSHLADD(R)  "v_lshl_add_u32 " R ", " R ", 1, %8"
#define A_FMA(R)      "v_fma_f32 " R ", " R ", %8, %9"
#define A_PKFMA(R)    "v_pk_add_f16 " R ", " R ", %8"
#define A_ADDSDWA(R)  "v_add_u16_sdwa " R ", " R ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2"
#define A_MAXI16(R)   "v_max_i16 " R ", " R ", %8"

DEFK(k_max, A_MAX) DEFK(k_add, A_ADD) DEFK(k_max3, A_MAX3) DEFK(k_subcl, A_SUBCL)
DEFK(k_dot4c, A_DOT4C) DEFK(k_dot4, A_DOT4) DEFK(k_perm, A_PERM) DEFK(k_bfe, A_BFE)
DEFK(k_add3, A_ADD3) DEFK(k_pkadd, A_PKADD) DEFK(k_pkmax, A_PKMAX) DEFK(k_pkmaxu, A_PKMAXU)
DEFK(k_pksubcl, A_PKSUBCL) DEFK(k_pkmad, A_PKMAD) DEFK(k_movdpp_r, A_MOVDPP_R)
DEFK(k_movdpp_w, A_MOVDPP_W) DEFK(k_anddpp, A_ANDDPP) DEFK(k_maxdpp, A_MAXDPP)
DEFK(k_dot4cdpp, A_DOT4CDPP)
DEFK(k_mad24, A_MAD24) DEFK(k_cndmask, A_CNDMASK) DEFK(k_med3, A_MED3) DEFK(k_lshladd, A_LSHLADD)
DEFK(k_fma, A_FMA) DEFK(k_pkaddf16, A_PKFMA) DEFK(k_addsdwa, A_ADDSDWA) DEFK(k_maxi16, A_MAXI16)

// One SW-like dependent row chain: dot4c -> max3 -> sub clamp, R rows, to see latency effects vs occupancy.
template <int R>
__global__ void __launch_bounds__(256) k_chain(int* out, int iters, int seed) {
  int h[R], p[R];
  #pragma unroll
  for (int i = 0; i < R; ++i) { h[i] = (threadIdx.x + i) & 15; p[i] = 0x01020304 * (i + 1) + seed; }
  int oh = 1 << (8 * (threadIdx.x & 3)), gap = seed & 3, best = 0, upin = threadIdx.x & 7, dg = 0;
  for (int it = 0; it < iters; ++it) {
    int up = upin, d = dg;
    dg = upin;
    #pragma unroll
    for (int i = 0; i < R; ++i) {
      int t = __builtin_amdgcn_sdot4(p[i], oh, d, false);
      d = h[i];
      int x = max(max(h[i], up), t);
      best = max(best, x);
      h[i] = (int)__builtin_elementwise_sub_sat((unsigned)x, (unsigned)gap);
      up = h[i];
    }
    upin = __builtin_amdgcn_update_dpp(0, up, 0x111, 0xf, 0xf, true);
    oh = (oh << 8) | ((unsigned)oh >> 24);
  }
  int s = best;
  #pragma unroll
  for (int i = 0; i < R; ++i) s += h[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

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
  int cus = prop.multiProcessorCount;
  double clk = prop.clockRate * 1e3;  // Hz
  printf("device %s CUs %d clockRate %.0f MHz\n", prop.name, cus, clk / 1e6);
  int* dout; CHECK(hipMalloc(&dout, sizeof(int) * 256 * cus * 16));
  struct { const char* name; kern_t k; } ks[] = {
    {"v_max_i32", k_max}, {"v_add_u32", k_add}, {"v_max3_i32", k_max3}, {"v_sub_u32 clamp", k_subcl},
    {"v_dot4c_i32_i8", k_dot4c}, {"v_dot4_i32_i8", k_dot4}, {"v_perm_b32", k_perm}, {"v_bfe_i32", k_bfe},
    {"v_add3_u32", k_add3}, {"v_pk_add_u16", k_pkadd}, {"v_pk_max_i16", k_pkmax}, {"v_pk_max_u16", k_pkmaxu},
    {"v_pk_sub_u16 clamp", k_pksubcl}, {"v_pk_mad_i16", k_pkmad}, {"v_mov_dpp row_shr", k_movdpp_r},
    {"v_mov_dpp wave_shr", k_movdpp_w}, {"v_and_dpp row_shr", k_anddpp}, {"v_max_i32_dpp", k_maxdpp},
    {"v_dot4c_dpp row_shr", k_dot4cdpp},
    {"v_mad_i32_i24", k_mad24}, {"v_cndmask_b32", k_cndmask}, {"v_med3_i32", k_med3}, {"v_lshl_add_u32", k_lshladd},
    {"v_fma_f32", k_fma}, {"v_pk_add_f16", k_pkaddf16}, {"v_add_u16_sdwa", k_addsdwa}, {"v_max_i16", k_maxi16},
  };
  const int iters = 4096;
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
    int blocks = cus * wps;        // 256 threads = 4 waves = 1 wave per SIMD per block
    printf("--- waves/SIMD = %d (blocks %d x 256)\n", wps, blocks);
    for (auto& e : ks) {
      double s = run(e.k, blocks, 256, iters, dout);
      double winstr = (double)blocks * 4 * iters * 64;  // wave-instructions
      double per_simd_per_clk = winstr / s / (cus * 4.0) / clk;
      printf("%-22s %8.3f ms  %.3f winstr/clk/SIMD  (%.2f clk/instr @%.0fMHz)  %.2f Tlane-op/s\n", e.name, s * 1e3,
             per_simd_per_clk, 1.0 / per_simd_per_clk, clk / 1e6, winstr * 64 / s / 1e12);
    }
  }
  // dependent-chain SW body
  for (int wps : {1, 2, 3, 4, 6, 8}) {
    int blocks = cus * wps;
    {
      double s = run(k_chain<16>, blocks, 256, 2048, dout);
      double cells = (double)blocks * 256 * 2048 * 16;
      printf("chain R=16 wps=%d: %.3f ms  %.2f Tcell/s (lane-cells)\n", wps, s * 1e3, cells / s / 1e12);
    }
    if (wps <= 4) {
      double s = run(k_chain<32>, blocks, 256, 1024, dout);
      double cells = (double)blocks * 256 * 1024 * 32;
      printf("chain R=32 wps=%d: %.3f ms  %.2f Tcell/s (lane-cells)\n", wps, s * 1e3, cells / s / 1e12);
    }
  }
  return 0;
}
