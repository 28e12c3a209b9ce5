#!/usr/bin/env python3
"""Generates cell_v3.hip: candidate instruction streams for the packed cell of round 3, with their real register
dependencies, R = 32 rows per step (what L = 4 runs), two steps per loop iteration.  Measures SIMD cycles per row at 1..4
wavefronts per SIMD.  Streams:
  cur_bias   round 2's bias form (pk_two_rows<true>): two copies of the column, T and A as 32-bit v_add_u32, s_nops
  cur_q0     round 2's Q = 0 form (pk_two_rows<false>)
  q0_t64     Q = 0 form with the diagonal adds of two rows as ONE v_lshl_add_u64 on an aligned register pair
  bias_t64   bias form with the same pairing of T (A stays on the dependency chain, 32-bit)
  vert       "vertical offset" bias form: row r works in domain g * (r + 1), so `up` is the row above's max3 result as it is
             (chain = max3 -> max3), the saturating subtraction and the re-biasing add leave the chain, and both adds pair
Run: python gen_cell_v3.py > cell_v3.hip && make cell_v3 && ./cell_v3"""
R = 32
LO = 32                  # v32..v63: the column (h / hq / L), even-aligned pairs
RSEL = 64                # v64..v95: per-row selectors
HU = 96                  # v96..v127: second copy (round 2's bias form only)


def v(n):
    return "v%d" % n


def pair(n):
    return "v[%d:%d]" % (n, n + 1)


def stream(name):
    o = []
    P = lambda i, dst: "v_perm_b32 %s, v2, v1, %s" % (dst, v(RSEL + i))
    if name in ("cur_bias", "cur_q0"):
        bias = name == "cur_bias"
        sc = lambda i: v(5 + i % 3)
        t = lambda i: v(3 + i % 2)
        x = lambda i: v(8 + i % 2)
        src = HU if bias else LO
        T = lambda i: "v_add_u32 %s, %s, %s" % (t(i), v(src + i - 1) if i else "v11", sc(i))
        M = lambda i: "v_pk_maximum3_f16 %s, %s, %s, %s" % (x(i), v(LO + i), v(LO + i - 1) if i else "v12", t(i))
        S = lambda i: "v_pk_sub_u16 %s, %s, s40 clamp" % (v(src + i), x(i))
        A = lambda i: "v_add_u32 %s, s41, %s" % (v(LO + i), v(HU + i))
        B = lambda i: "v_pk_maximum3_f16 v10, v10, %s, %s" % (x(i - 1), x(i))
        n = "s_nop 0"
        o += [P(0, sc(0)), P(1, sc(1)), T(0)]
        for i in range(0, R, 2):
            last = i + 2 >= R
            if bias:
                o += [M(i), T(i + 1), n, S(i), n if last else P(i + 2, sc(i + 2)), A(i), n, M(i + 1)] + ([n] if last else [T(i + 2), n])
                o += [S(i + 1)] + ([] if last else [P(i + 3, sc(i + 3))]) + [n, A(i + 1), B(i + 1)]
            else:
                o += [M(i), T(i + 1), n, S(i), n if last else P(i + 2, sc(i + 2)), n, M(i + 1)] + ([n] if last else [T(i + 2), n])
                o += [S(i + 1)] + ([] if last else [P(i + 3, sc(i + 3))]) + [B(i + 1)]
        return o
    # paired forms.  SC_k = (sc_{2k+1}, sc_{2k+2}) in v[20:21] / v[22:23] alternating; TT = (t_{2k+1}, t_{2k+2}) in v[24:25];
    # t_0 in v3; x in v8 / v9 alternating by row; best v10; diag v11; up v12
    SC = lambda k: 20 + 2 * (k % 2)
    TT = 24
    x = lambda i: v(8 + i % 2)
    sc_reg = lambda i: v(SC((i - 1) // 2) + (i - 1) % 2) if i else "v5"          # where P_i puts its result
    t_reg = lambda i: v(TT + (i - 1) % 2) if i else "v3"
    T64 = lambda k, src: "v_lshl_add_u64 %s, %s, 0, %s" % (pair(TT), pair(src + 2 * k), pair(SC(k)))
    Pn = lambda i: P(i, sc_reg(i)) if i < R else "s_nop 0"
    if name == "q0_t64":
        M = lambda i: "v_pk_maximum3_f16 %s, %s, %s, %s" % (x(i), v(LO + i), v(LO + i - 1) if i else "v12", t_reg(i))
        S = lambda i: "v_pk_sub_u16 %s, %s, s40 clamp" % (v(LO + i), x(i))
        B = lambda i: "v_pk_maximum3_f16 v10, v10, %s, %s" % (x(i - 1), x(i))
        o += [P(0, "v5"), "v_add_u32 v3, v11, v5", Pn(1), Pn(2)]
        for k in range(R // 2):
            a, b = 2 * k, 2 * k + 1
            o += [M(a), T64(k, LO) if b + 1 < R else "v_add_u32 %s, %s, %s" % (v(TT), v(LO + a), v(SC(k))), S(a), Pn(b + 2), M(b), Pn(b + 3), S(b), B(b)]
        return o
    if name == "bias_t64":
        M = lambda i: "v_pk_maximum3_f16 %s, %s, %s, %s" % (x(i), v(LO + i), v(LO + i - 1) if i else "v12", t_reg(i))
        S = lambda i: "v_pk_sub_u16 %s, %s, s40 clamp" % (v(HU + i), x(i))
        A = lambda i: "v_add_u32 %s, s41, %s" % (v(LO + i), v(HU + i))
        B = lambda i: "v_pk_maximum3_f16 v10, v10, %s, %s" % (x(i - 1), x(i))
        o += [P(0, "v5"), "v_add_u32 v3, v11, v5", Pn(1), Pn(2)]
        for k in range(R // 2):
            a, b = 2 * k, 2 * k + 1
            o += [M(a), T64(k, HU) if b + 1 < R else "v_add_u32 %s, %s, %s" % (v(TT), v(HU + a), v(SC(k))), S(a), Pn(b + 2), A(a), "s_nop 0", M(b), Pn(b + 3), S(b), B(b), A(b), "s_nop 0"]
        return o
    if name == "vert":
        # L in v32..v63; hu pairs v[16:17] / v[18:19] alternating by block; K_r (g + D_r) in s42.. (one SGPR stands in for the
        # 32 per-row constants: same cost), DD_k in s[44:45]
        HUP = lambda k: 16 + 2 * (k % 2)
        M = lambda i: "v_pk_maximum3_f16 %s, %s, %s, %s" % (x(i), v(LO + i), x(i - 1) if i else "v12", t_reg(i))
        S = lambda i: "v_pk_sub_u16 %s, %s, s42 clamp" % (v(HUP(i // 2) + i % 2), x(i))
        A64 = lambda k: "v_lshl_add_u64 %s, %s, 0, s[44:45]" % (pair(LO + 2 * k), pair(HUP(k)))
        B = lambda k: "v_pk_maximum3_f16 v10, v10, %s, %s" % (v(HUP(k)), v(HUP(k) + 1))
        o += [P(0, "v5"), "v_add_u32 v3, v11, v5", Pn(1), Pn(2)]
        for k in range(R // 2):
            a, b = 2 * k, 2 * k + 1
            o += [M(a)] + ([A64(k - 1)] if k else ["s_nop 0"]) + [S(a), T64(k, LO) if b + 1 < R else "v_add_u32 %s, %s, %s" % (v(TT), v(LO + a), v(SC(k)))]
            o += ([B(k - 1)] if k else []) + [Pn(b + 2), M(b), Pn(b + 3), S(b)]
        o += ["s_nop 0", A64(R // 2 - 1), B(R // 2 - 1)]
        return o
    raise SystemExit(name)


VARIANTS = ["cur_bias", "cur_q0", "q0_t64", "bias_t64", "vert"]

print('''// GENERATED by gen_cell_v3.py -- do not edit.  Candidate instruction streams for the packed cell (DESIGN.md 5a, round 3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
''')
clob = ", ".join('"v%d"' % k for k in range(1, 128)) + ', "s40", "s41", "s42", "s44", "s45"'
for name in VARIANTS:
    body = stream(name) + stream(name)
    print("__global__ void __launch_bounds__(256) k_%s(int *out, int iters, int seed) {" % name)
    print('  asm volatile("s_mov_b32 s40, 0x000f000f\\n\\t s_mov_b32 s41, 0x00030003\\n\\t s_mov_b32 s42, 0x000f000f\\n\\t s_mov_b32 s44, 0x00010001\\n\\t s_mov_b32 s45, 0x00020002\\n\\t"')
    for k in range(1, 128):
        print('    "v_mov_b32 v%d, %s\\n\\t"' % (k, "0x0c040c00" if RSEL <= k < RSEL + R else "0x00050007"))
    print('    ::: %s);' % clob)
    print("  for (int it = 0; it < iters; ++it) {")
    print('    asm volatile(')
    for ins in body:
        print('      "%s\\n\\t"' % ins)
    print('      ::: %s);' % clob)
    print("  }")
    print('  int r; asm volatile("v_add_u32 %0, v10, v47" : "=v"(r));')
    print("  out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;\n}")
    nv = sum(1 for i in body if not i.startswith("s_nop"))
    print("static const int n_%s = %d, valu_%s = %d;" % (name, len(body), name, nv))
print('''
typedef void (*kern_t)(int *, int, int);
static double run(kern_t k, int blocks, int iters, int *dout) {
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
  int *dout; CHECK(hipMalloc(&dout, sizeof(int) * 256 * cus * 16));
  struct { const char *name; kern_t k; int n, valu; } ks[] = {''')
print(",\n".join('    {"%s", k_%s, n_%s, valu_%s}' % (n, n, n, n) for n in VARIANTS))
print('''  };
  const int iters = 256;
  printf("packed cell candidates, %d rows per step; clk = SIMD cycles per row of one wavefront at the device's nominal clock\\n");
  for (int wps : {1, 2, 3, 4}) {
    int blocks = cus * wps; printf("--- wavefronts per SIMD = %%d\\n", wps);
    for (auto &e : ks) {
      double s = run(e.k, blocks, iters, dout);
      double rows = (double)iters * %d;
      double clk_per_row = clk * s / rows / wps;
      printf("%%-12s %%8.3f ms  %%6.2f clk per row  (%%d instructions, %%d VALU per %%d rows)\\n", e.name, s * 1e3, clk_per_row, e.n, e.valu, %d);
    }
  }
  return 0;
}''' % (R, 2 * R, 2 * R))
