// sw_kernels.hip -- hand-written gfx950 (CDNA4, MI355X) kernels for the fixed-shape Smith-Waterman scorer.
//
// Semantics: the reference scalar recurrence (source.cpp:49-53)
//     H(i,j) = max(0, H(i-1,j-1) + sm[seq1[i-1]*4 + seq2[j-1]], H(i-1,j) - gap, H(i,j-1) - gap)
//     score  = max over the 128 x 128 cells
// This is NOT a translation of the reference's AVX2 parallelogram (source.cpp:462-571). The design is
// driven by what the gfx950 VALU issues at full / half rate (tools/microbench/*.hip):
//
//   * A 64-lane wavefront walks the anti-diagonals of 64/L alignments at once: L lanes per alignment,
//     lane j of a group owns the R = 128/L consecutive rows j*R .. j*R+R-1 and, at step t, computes the
//     R cells of column c = t - j (top to bottom).  L = 64 is literally "one wavefront per alignment";
//     smaller L trades lanes-in-flight for a shorter pipeline fill (L-1 idle steps of 128+L-1) and fewer
//     cross-lane moves per cell.  Default L = 4.
//   * The vertical/diagonal dependency between neighbouring lanes is one DPP lane shift per step
//     (v_and_b32_dpp / v_mov_b32_dpp row_shr:1 or wave_shr:1 -- what __shfl_up(v,1) should be on gfx9;
//     the compiler lowers __shfl_up to ds_bpermute_b32, which costs an LDS round trip).
//   * The query profile lives in LDS: for every column of seq2 a one-hot dword (1 << 8*base).  A lane reads
//     the profile entry of its current column with one ds_read_b32 per step; out-of-range columns hit zero pads.
//   * The score lookup AND the diagonal add are ONE instruction: v_dot4_i32_i8 (acc = H(i-1,j-1), a = the row's
//     4 int8 scores, b = the column's one-hot).
//   * The three-way max is one v_max3_i32, "- gap, floor 0" one v_sub_u32 ... clamp; half of the running-maximum
//     updates go to the otherwise idle LDS unit as ds_max_i32.
//
// Cell bodies, identical results (template flags):
//   folded  : rows carry s + gap (host folds when every s + gap fits int8):
//             x = max3(left, up, dot4(row', onehot, diag)); h = x -sat gap; the maximum is tracked on x
//   general : any int8 matrix: h = max3(gleft, gup, dot4(row, onehot, diag)); g = h -sat gap (h and g kept per row)
//   16-bit  : v_max_i16 formulation, kept for A/B only
//
// Why nothing is masked during pipeline fill and drain (lane j works on column t - j, which is < 0 for the first j
// steps and >= 128 for the last L-1-j):
//   before the first column: every profile entry read is 0, so dot4 adds nothing; all inputs of the cell (left, up,
//     diag) are still 0, hence x = 0 (folded) / h = 0 (general) and the state stays all-zero until column 0 arrives;
//   after the last column: the entries are 0 again, the cell computes x = max(left, up, diag) -- every value it can
//     produce is bounded by a value some real cell already held, which the running maximum has seen; such cells only
//     feed cells that are themselves past the last column (a lane's neighbour j+1 is one column behind it), so neither
//     the score nor any real cell can change.  This needs gap >= 0, which the C ABI enforces.
#include "swmi_internal.h"

#include <type_traits>
#include <utility>

namespace swmi {
namespace {

constexpr int kSeqLen = 128;
constexpr int kWavesPerBlock = 4;

// ---- small device helpers -----------------------------------------------------------------------

__device__ __forceinline__ int max_i16(int a, int b)
{
    // v_max_i16: full rate on gfx950 (v_max_i32 / v_max3_i32 are half rate).  Inputs are read as their low
    // 16 bits, the result is zero-extended (gfx9 16-bit VALU ops clear dst[31:16]).  Expressed through the
    // compiler (not inline asm) so that the hazard recogniser keeps the wait states a VALU read of a
    // v_dot4 result needs -- an asm v_max_i16 placed right behind v_dot4c reads a stale register.
    const short r = __builtin_elementwise_max((short)a, (short)b);
    return (int)(unsigned short)r;
}

template <bool I16>
__device__ __forceinline__ int vmax(int a, int b)
{
    if constexpr (I16) return max_i16(a, b);
    else return a > b ? a : b;
}

template <bool I16>
__device__ __forceinline__ int sat_sub(int a, int b)   // max(a - b, 0) for 0 <= a, b < 2^15 : v_sub_u16 / v_sub_u32 ... clamp (full rate)
{
    if constexpr (I16) return (int)(unsigned short)__builtin_elementwise_sub_sat((unsigned short)a, (unsigned short)b);
    else return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b);
}

// Opaque to the optimiser, free at run time: stops hipcc from re-associating two full-rate v_max_i16 into one
// quarter-rate v_max3_i16 (8 clk per wave64 on gfx950 against 2 x 2.25).
__device__ __forceinline__ int keep(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Value of the same register in lane-1 of the group (0 for the first lane of a group).
template <int L>
__device__ __forceinline__ int from_prev_lane(int v, int group_mask)
{
    if constexpr (L == 64) {
#ifdef SWMI_AB_SHFL_UP      // A/B build only (make ab_shfl_up, tools/shfl_up_ab.py; never shipped): north_star's literal
        const int s = __shfl_up(v, 1);                       // __shfl_up(h, 1), which hipcc lowers to ds_bpermute_b32
        return (threadIdx.x & 63) ? s : 0;
#else
        return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
#endif
    } else if constexpr (L == 32) {
        return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true) & group_mask;
    } else if constexpr (L == 16) {
        return __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
    } else {
        return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true) & group_mask;
    }
}

// R one-byte bases -> NW dwords, 4 bases per dword
template <int R>
__device__ __forceinline__ void load_bases(const uint8_t *p, uint32_t (&w)[(R + 3) / 4])
{
    if constexpr (R == 2) {
        w[0] = *reinterpret_cast<const uint16_t *>(p);
    } else if constexpr (R == 4) {
        w[0] = *reinterpret_cast<const uint32_t *>(p);
    } else if constexpr (R == 8) {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        w[0] = v.x; w[1] = v.y;
    } else {
#pragma unroll
        for (int k = 0; k < R / 16; ++k) {
            const uint4 v = *reinterpret_cast<const uint4 *>(p + 16 * k);
            w[4 * k + 0] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
    }
}

// 8 bits holding four 2-bit bases (LSB first, source.cpp:1581) -> dword with one base per byte
__device__ __forceinline__ uint32_t spread4(uint32_t v)
{
    return (v | (v << 6) | (v << 12) | (v << 18)) & 0x03030303u;
}

// R two-bit bases (R/4 bytes at p) -> NW dwords, one base per byte
template <int R>
__device__ __forceinline__ void load_bases_packed(const uint8_t *p, uint32_t (&w)[(R + 3) / 4])
{
    if constexpr (R == 2) {
        // half a nibble: lane j of 64 owns bases 2j, 2j+1 -> byte j/2, bit offset 4*(j&1); caller passes p = byte address
        w[0] = 0;  // handled by caller (needs lane parity); see sw128_kernel
    } else if constexpr (R == 4) {
        w[0] = spread4(p[0]);
    } else if constexpr (R == 8) {
        const uint32_t v = *reinterpret_cast<const uint16_t *>(p);
        w[0] = spread4(v & 0xff); w[1] = spread4(v >> 8);
    } else if constexpr (R == 16) {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(p);
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = spread4((v >> (8 * k)) & 0xff);
    } else {
#pragma unroll
        for (int q = 0; q < R / 32; ++q) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p + 8 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                w[8 * q + k] = spread4((v.x >> (8 * k)) & 0xff);
                w[8 * q + 4 + k] = spread4((v.y >> (8 * k)) & 0xff);
            }
        }
    }
}

// ---- the scoring kernel -------------------------------------------------------------------------
//
// MODE 0: pair k = 128 B at seq1s + 128k / seq2s + 128k   (swmi_score_batch*)
// MODE 1: 2-bit packed, pair k = 32 B at + 32k            (swmi_score_batch_packed*)
// MODE 2: one-vs-many: seq1 k at seq1s + 128k, ONE seq2    (swmi_score_one_vs_many)
template <int L, bool FOLD, bool I16, int MODE>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
sw128_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, int32_t *__restrict__ scores,
             uint32_t n, SmRows rows, int gap)
{
    constexpr int R = kSeqLen / L;          // rows per lane
    constexpr int A = 64 / L;               // alignments per wavefront
    constexpr int PAD = L + 2;              // zero profile entries either side (the 2-step loop may run one step past T
                                            // and reads one entry ahead)
    constexpr int S = kSeqLen + 2 * PAD;    // profile entries per alignment
    constexpr int T2 = (kSeqLen + L) / 2;   // pairs of anti-diagonal steps: covers T = 128 + L - 1 steps (+1 harmless)
    constexpr int NW = (R + 3) / 4;

    // One-hot query profile of seq2, per wave: the one-hot dword itself (1 << 8*base, 0 for a pad column).
    // (A 16-bit encoding (1 << 8) | 8*base, decoded per step with two shifts, halves the LDS and lifts L = 4 from 4 to 5
    // workgroups per CU -- measured slower: the kernel is issue bound from 3 waves per SIMD up, the two shifts are not free.)
    constexpr bool kWideProfile = true;
    using profile_t = std::conditional_t<kWideProfile, uint32_t, uint16_t>;
    __shared__ profile_t lds_profile[kWavesPerBlock][A * S];
    __shared__ uint32_t lds_rows[kWavesPerBlock][4];          // score-matrix rows, per wave
    // Part of the running maximum is kept by the LDS unit: rows with (i % kBestDen) < kBestNum send their value to a
    // per-lane LDS word with ds_max_i32 (no return value, wavefront scope) instead of spending VALU issue cycles on it.
    // The LDS pipe is otherwise idle in this kernel; one such atomic costs it ~4.3 cycles per wavefront, so about half
    // of the rows is what it can absorb before it becomes the bottleneck itself.
#ifndef SWMI_LDS_BEST_NUM
#define SWMI_LDS_BEST_NUM (R >= 64 ? 0 : R >= 32 ? 1 : R >= 16 ? 2 : R >= 8 ? 1 : 0)   // measured per L, 1M pairs
#define SWMI_LDS_BEST_DEN (R >= 64 ? 1 : R >= 32 ? 2 : R >= 16 ? 3 : R >= 8 ? 2 : 1)
#endif
    constexpr int kBestNum = I16 ? 0 : SWMI_LDS_BEST_NUM, kBestDen = SWMI_LDS_BEST_DEN;
    __shared__ int lds_best[kWavesPerBlock][64];
    lds_best[threadIdx.x >> 6][threadIdx.x & 63] = 0;         // own word, read back by the same lane: no barrier needed

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane / L;                 // alignment slot inside the wave
    const int j = lane % L;                 // lane inside the group
    const uint32_t first_pair = (blockIdx.x * kWavesPerBlock + wv) * A;
    if (first_pair >= n) return;            // wave-uniform: only wave-level synchronisation below
    uint32_t pair = first_pair + g;
    const bool live = pair < n;
    if (!live) pair = n - 1;                // ragged tail: recompute the last pair, do not store

    // ---- load this lane's R rows of seq1 and its R columns of seq2 (coalesced: lane * R bytes) ----
    uint32_t a_w[NW], b_w[NW];
    if constexpr (MODE == 1) {
        if constexpr (R == 2) {
            const uint32_t sh = 4 * (j & 1);
            a_w[0] = spread4((seq1s[(size_t)pair * 32 + (j >> 1)] >> sh) & 0xf);
            b_w[0] = spread4((seq2s[(size_t)pair * 32 + (j >> 1)] >> sh) & 0xf);
        } else {
            load_bases_packed<R>(seq1s + (size_t)pair * 32 + j * (R / 4), a_w);
            load_bases_packed<R>(seq2s + (size_t)pair * 32 + j * (R / 4), b_w);
        }
    } else {
        load_bases<R>(seq1s + (size_t)pair * kSeqLen + j * R, a_w);
        load_bases<R>(seq2s + (MODE == 2 ? (size_t)0 : (size_t)pair * kSeqLen) + j * R, b_w);
    }

    // ---- stage the score-matrix rows and the one-hot query profile in LDS ------------------------
    if (lane < 4) lds_rows[wv][lane] = rows.r[lane];
    profile_t *prof = &lds_profile[wv][g * S];
    for (int k = j; k < PAD; k += L) {
        prof[k] = 0;
        prof[PAD + kSeqLen + k] = 0;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const uint32_t b = (b_w[i / 4] >> (8 * (i % 4))) & 3u;
        prof[PAD + j * R + i] = kWideProfile ? (profile_t)(1u << (8u * b)) : (profile_t)(0x100u | (8u * b));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    int row_scores[R];                      // 4 x int8 (sm[a_i][0..3], + gap when FOLD) per owned row
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const uint32_t a = (a_w[i / 4] >> (8 * (i % 4))) & 3u;
        row_scores[i] = (int)lds_rows[wv][a];
    }

    // from here on the wavefront only issues VALU work: let it win arbitration against wavefronts that are still in their
    // (memory-bound) prologue (+0.4 %)
    __builtin_amdgcn_s_setprio(2);

    // ---- anti-diagonal sweep, two steps per iteration ---------------------------------------------------
    // Three cell bodies, identical scores:
    //   FOLD            rows hold s + gap:  x = max3(left, up, dot4(row', onehot, diag)); h = x -sat gap; best tracks x
    //   !FOLD, !I16     any int8 matrix:    h = max3(gleft, gup, dot4(row, onehot, diag)); g = h -sat gap  (g kept per row:
    //                                       one more register per row, same 3.5 instructions per cell as FOLD)
    //   !FOLD, I16      16-bit max form:    h = max(max(left, up) -sat gap, dot4(...)) on v_max_i16 (A/B only)
    constexpr bool kKeepG = !FOLD && !I16;
    int h[R];                               // h[i] = H(row i, previous column)
    int hg[kKeepG ? R : 1];                 // hg[i] = max(h[i] - gap, 0)
#pragma unroll
    for (int i = 0; i < R; ++i) h[i] = 0;
#pragma unroll
    for (int i = 0; i < (kKeepG ? R : 1); ++i) hg[i] = 0;
    int best = 0;
    int u0 = 0, u1 = 0;                     // H(last row of lane j-1): this step's column / the previous one, alternating
    // opaque so that hipcc emits one v_and_b32_dpp per step instead of v_mov_b32_dpp + v_cndmask_b32
    const int group_mask = keep(j == 0 ? 0 : -1);
    const profile_t *col = prof + PAD - j;  // col[t] = profile entry of column t - j

    // one anti-diagonal step: `up` = H(last row of lane j-1, this column), `diag` = same, previous column
    auto step = [&](uint32_t entry, int up, int diag) {
        const int onehot = kWideProfile ? (int)entry : (int)((entry >> 8) << (entry & 31u));
        // Diagonal terms t[i] = H(i-1, c-1) + s(i, c) (+ gap): one v_dot4_i32_i8 each (lookup and add fused).
        // clamp = true selects the 3-address VOP3P form (the 2-address v_dot4c_i32_i8 costs a v_mov per cell); the
        // saturation can never trigger (|t| < 2^15).
        int dprev = diag;
        if constexpr (kKeepG) up = sat_sub<false>(up, gap);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int left = h[i];
            const int tsum = __builtin_amdgcn_sdot4(row_scores[i], onehot, dprev, true);
            int hn;
            if constexpr (FOLD) {
                const int lu = left > up ? left : up;
                const int x = lu > tsum ? lu : tsum;                        // v_max3_i32 (half rate, two maxes)
                if constexpr (I16) best = keep(max_i16(best, x));
                else if (i % kBestDen < kBestNum) __hip_atomic_fetch_max(&lds_best[wv][lane], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                else best = vmax<false>(best, x);
                hn = sat_sub<I16>(x, gap);
                up = hn;
            } else if constexpr (I16) {
                const int m = keep(max_i16(left, up));
                hn = keep(max_i16(sat_sub<true>(m, gap), tsum));
                best = keep(max_i16(best, hn));
                up = hn;
            } else {
                const int gl = hg[i];
                const int lu = gl > up ? gl : up;
                hn = lu > tsum ? lu : tsum;                                 // v_max3_i32; >= 0 because gl >= 0
                if (i % kBestDen < kBestNum) __hip_atomic_fetch_max(&lds_best[wv][lane], hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                else best = vmax<false>(best, hn);
                const int gn = sat_sub<false>(hn, gap);
                hg[i] = gn;
                up = gn;
            }
            h[i] = hn;
            dprev = left;
        }
        return from_prev_lane<L>(h[R - 1], group_mask);
    };

    uint32_t e0 = col[0];
    for (int t2 = 0; t2 < T2; ++t2) {
        const uint32_t e1 = col[2 * t2 + 1];
        const uint32_t e2 = col[2 * t2 + 2];
        u1 = step(e0, u0, u1);              // u1: from lane j-1 for the next step; u0 becomes its diagonal
        u0 = step(e1, u1, u0);
        e0 = e2;
    }

    // ---- reduce over the L lanes of the group, one int32 per alignment ------------------------------
    if constexpr (kBestNum > 0) {
        const int from_lds = lds_best[wv][lane];
        best = best > from_lds ? best : from_lds;
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
        const int other = __shfl_xor(best, o);
        best = best > other ? best : other;
    }
    if constexpr (FOLD) best = best > gap ? best - gap : 0;
    if (j == 0 && live) scores[pair] = best;
}


// ---- packed kernel: two alignments per register, any parameters (L = 4, 8, 16) ----------------------------------------
//
// gfx950's new v_pk_maximum3_f16 is, for NON-NEGATIVE 16-bit integers below 0x7C00, a packed THREE-INPUT INTEGER MAX:
// such integers order exactly like the IEEE half-precision numbers with the same bit patterns (denormals included; hipcc
// kernels keep f16 denormals; tools/experiments/pk_max3_probe.hip checks the claim on 1M random triples).  Two cells' max3
// in one half-rate instruction (v_max3_i32 does one) is what makes 16-bit packing pay on this VALU -- with v_pk_max_i16
// alone it only ties with the int32 cell.  Every value of the gap-folded recurrence is kept
// non-negative and below 16 768 by working on values SHIFTED by a bias Q:
//     sc = s + gap + Q >= 0            (Q = max(0, -(min s + gap)); one byte: s + gap + Q <= 255 for every int8 matrix)
//     t  = H(i-1,j-1) + sc             (the folded diagonal term + Q: a plain add, never negative)
//     x  = max3(left + Q, up + Q, t)   (= the unshifted x + Q)
//     H  = x -sat (gap + Q)            (and H + Q for the neighbours)
// The previous column is held twice, as H (what t starts from) and as H + Q (what `left` and `up` are): both additions are
// 32-bit FULL-RATE v_add_u32 on the packed pair -- no half ever carries into the other -- instead of a second saturating
// packed subtraction for t.  When every s + gap is already >= 0 -- (1,-1,1), the parameters of SmithWaterman_8bit111simd /
// _8b111x32 (source.cpp:1105-1522), are the model case -- Q = 0, the two copies are one and the "+ Q" disappears
// (VARIANT kPkQ0).
//   * one VGPR holds the SAME row of TWO alignments (low / high half), so there is no dependency inside a pair and the
//     two alignments share every instruction; a lane group of L lanes walks two alignments (L = 4: a wavefront 32);
//   * score lookup for both halves = ONE v_perm_b32: its 8 source bytes are the score tables of the CURRENT COLUMN of
//     alignment X and of alignment Y (4 bytes each: the column's base against the four row bases, fetched per step from a
//     5-entry table in LDS through the per-column offsets staged there), its selector is a per-ROW register that picks
//     table[X][base of X's row] into byte 0 and table[Y][base of Y's row] into byte 2 (zero bytes between; pad columns
//     fetch the all-zero table).  A row costs ONE register for the lookup (round 2's first version kept the two row tables
//     in registers, 2 per row, with the selector per column), which is what makes room for the second copy of H;
//   * per row and alignment pair: v_perm_b32, v_add_u32, v_pk_maximum3_f16, v_pk_sub_u16 clamp, (v_add_u32,) and half a
//     v_pk_maximum3_f16 for the running best: 16 (Q = 0) / 18 nominal issue cycles per 2 cells;
//   * issued in a hand-chosen order (pk_two_rows below).
// Round 3: on this VALU every instruction of the cell costs one ~4.3-cycle issue slot whatever its class
// (profiles/r03_microbench_cell_v3.txt), so what counts is the NUMBER of instructions, and two cell bodies with fewer of them
// take over wherever the parameters allow (generated code, gen_pk_sweeps.py has the derivations):
//   kPkQ0    every s + gap >= 0 (no bias needed): the diagonal adds of rows r, r + 1 are ONE v_lshl_add_u64 on an aligned
//            register pair -- 8 instructions per two rows (round 2: 9);
//   kPkVert  every s + 2 gap >= 0, e.g. (10,-30,15): row r works in its own domain D_r = gap (r + 1), which makes `up` the
//            row above's max3 result as it is (chain max3 -> max3), takes the saturating subtraction and the re-biasing add
//            off the chain and lets both adds pair -- 9 instructions per two rows (the bias form below: 11);
//   kPkBias  everything else: the form described above.
// Same anti-diagonal pipeline, DPP hand-over and pad-column argument as sw128_kernel (header of this file).
__device__ __forceinline__ uint32_t pk_max3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The integers below 1024 are half-precision DENORMALS as bit patterns: make sure the wavefront keeps them (MODE.FP_DENORM
// bits 7:6 = f16 / f64 input and output denormals allowed -- hipcc's default, set here so that no compile flag can undo it).
// Every kernel that uses v_pk_maximum3_f16 as an integer max calls this first, pk_max3_selftest_kernel included.
__device__ __forceinline__ void keep_f16_denormals()
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 3");
}

// Self-test of the premise above (swmi_selftest_pk_max3; tests/test_gpu_pk_kernels.py runs it under pytest -m gpu): for
// EVERY pair (a, b) of 16-bit integers in [0, 0x7C00) -- 31744^2 = 1.0e9 pairs -- the packed max3 of registers built from
// a, b and a pseudo-random third value r equals the integer max3 of each half, with the third operand one of the two, the
// random one, and in every operand position.  Thread = one a, kBPerThread consecutive b.
constexpr uint32_t kPkLimit = 0x7C00u;      // first f16 bit pattern that is not a finite number (+Inf)
constexpr uint32_t kBPerThread = 1024;
__global__ void __launch_bounds__(256)
pk_max3_selftest_kernel(unsigned long long *__restrict__ counts /* [0] checked, [1] mismatches */)
{
    keep_f16_denormals();
    const uint32_t a = blockIdx.x * 256 + threadIdx.x;
    const uint32_t b0 = blockIdx.y * kBPerThread;
    if (a >= kPkLimit) return;
    auto imax3 = [](uint32_t p, uint32_t q, uint32_t r) { const uint32_t m = p > q ? p : q; return m > r ? m : r; };
    // (the instruction as the scoring kernel issues it, followed by the wait states pk_two_rows keeps between a packed result
    // and its consumer -- here the consumer is whatever hipcc schedules next)
    auto pk_max3 = [](uint32_t p, uint32_t q, uint32_t r) {
        uint32_t o;
        asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3\n\ts_nop 1" : "=v"(o) : "v"(p), "v"(q), "v"(r));
        return o;
    };
    uint32_t bad = 0, done = 0;
    for (uint32_t b = b0; b < b0 + kBPerThread && b < kPkLimit; ++b) {
        uint32_t h = (a * 0x9E3779B1u) ^ (b * 0x85EBCA77u);
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        const uint32_t r = h % kPkLimit, r2 = (h >> 7) % kPkLimit;
        const uint32_t x = a | (b << 16), y = b | (r << 16), z = r2 | (a << 16);
        // each line: the packed result against the integer max3 of the low halves | the high halves
        bad += pk_max3(x, y, z) != (imax3(a, b, r2) | (imax3(b, r, a) << 16));
        bad += pk_max3(z, x, y) != (imax3(r2, a, b) | (imax3(a, b, r) << 16));
        bad += pk_max3(y, z, x) != (imax3(b, r2, a) | (imax3(r, a, b) << 16));
        bad += pk_max3(x, y, x) != (imax3(a, b, a) | (imax3(b, r, b) << 16));      // third operand = one of the two
        bad += pk_max3(x, y, y) != (imax3(a, b, b) | (imax3(b, r, r) << 16));
        bad += pk_max3(x, x, y) != (imax3(a, a, b) | (imax3(b, b, r) << 16));
        done += 6;
    }
    atomicAdd(&counts[0], (unsigned long long)done);
    if (bad) atomicAdd(&counts[1], (unsigned long long)bad);
}
// TWO ROWS of the packed cell as one block of assembly, in a hand-chosen order of issue:
//   * a wavefront issues in order and a dependent instruction waits for its producer, so an independent instruction sits
//     between any two dependent ones: the lookup (P, v_perm_b32) runs two rows ahead and the diagonal add (T) one row ahead
//     of the chain  M (max3) -> S (saturating subtract) [-> A (add Q)] -> M of the next row.  That also keeps every
//     consumer of a packed (VOP3P) result one instruction away from its producer -- the wait state hipcc inserts an s_nop
//     for when it schedules such code itself;
//   * an s_nop follows each FULL-RATE add that is followed by a half-rate instruction.  A full-rate instruction fills half
//     of a 4-cycle issue slot and a second wavefront's full-rate instruction can take the other half; with
//     its own next instruction ready at once, the wavefront claims the following slot instead and the pairing is lost.
//     Measured on 1M pairs (bias form): no s_nop 1.254 ms, this placement 1.147 ms, an s_nop after every instruction
//     1.288 ms; Q = 0 form: 1.046 / 1.002 ms.  (profiles/r02_pk_cell_order_search.txt)
// Rows i (suffix 0) and i+1 (suffix 1).  in: t = diagonal term of row i, sc = looked-up scores of row i+1, up = H + Q of
// the row above;  out: t = diagonal term of row i+2, sc = scores of row i+3; hq1 is the next block's `up`.
// LAST: rows R-2, R-1 -- nothing to look ahead to; s_nop keeps the spacing.
template <bool LAST>
__device__ __forceinline__ void pk_two_rows(uint32_t &hq0, uint32_t &hq1, uint32_t &hu0, uint32_t &hu1, uint32_t &best, uint32_t &t,
                                            uint32_t &sc, uint32_t up, uint32_t sel2, uint32_t sel3, uint32_t cx, uint32_t cy,
                                            uint32_t gq2, uint32_t q2)
{
    uint32_t x0, x1, s2;
    if constexpr (!LAST) {
        asm volatile("v_pk_maximum3_f16 %[x0], %[hq0], %[up], %[t]\n\t"          // M0   x + Q
                     "v_add_u32 %[t], %[hu0], %[sc]\n\t"                          // T1   H(row i, previous column) + score
                     "s_nop 0\n\t"
                     "v_pk_sub_u16 %[hu0], %[x0], %[gq] clamp\n\t"                // S0   H = (x + Q) -sat (gap + Q)
                     "v_perm_b32 %[s2], %[cy], %[cx], %[sel2]\n\t"                // P2
                     "v_add_u32 %[hq0], %[q], %[hu0]\n\t"                         // A0   H + Q
                     "s_nop 0\n\t"
                     "v_pk_maximum3_f16 %[x1], %[hq1], %[hq0], %[t]\n\t"          // M1
                     "v_add_u32 %[t], %[hu1], %[s2]\n\t"                          // T2
                     "s_nop 0\n\t"
                     "v_pk_sub_u16 %[hu1], %[x1], %[gq] clamp\n\t"                // S1
                     "v_perm_b32 %[sc], %[cy], %[cx], %[sel3]\n\t"                // P3
                     "s_nop 0\n\t"
                     "v_add_u32 %[hq1], %[q], %[hu1]\n\t"                         // A1
                     "v_pk_maximum3_f16 %[best], %[best], %[x0], %[x1]"           // B
                     : [hq0] "+v"(hq0), [hq1] "+v"(hq1), [hu0] "+v"(hu0), [hu1] "+v"(hu1), [best] "+v"(best), [t] "+v"(t),
                       [sc] "+v"(sc), [x0] "=&v"(x0), [x1] "=&v"(x1), [s2] "=&v"(s2)
                     : [up] "v"(up), [sel2] "v"(sel2), [sel3] "v"(sel3), [cx] "v"(cx), [cy] "v"(cy), [gq] "s"(gq2), [q] "s"(q2));
    } else {
        asm volatile("v_pk_maximum3_f16 %[x0], %[hq0], %[up], %[t]\n\t"
                     "v_add_u32 %[t], %[hu0], %[sc]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_sub_u16 %[hu0], %[x0], %[gq] clamp\n\t"
                     "s_nop 0\n\t"
                     "v_add_u32 %[hq0], %[q], %[hu0]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_maximum3_f16 %[x1], %[hq1], %[hq0], %[t]\n\t"
                     "s_nop 0\n\t"
                     "v_pk_sub_u16 %[hu1], %[x1], %[gq] clamp\n\t"
                     "v_pk_maximum3_f16 %[best], %[best], %[x0], %[x1]\n\t"
                     "v_add_u32 %[hq1], %[q], %[hu1]"
                     : [hq0] "+v"(hq0), [hq1] "+v"(hq1), [hu0] "+v"(hu0), [hu1] "+v"(hu1), [best] "+v"(best), [t] "+v"(t),
                       [x0] "=&v"(x0), [x1] "=&v"(x1)
                     : [up] "v"(up), [sc] "v"(sc), [gq] "s"(gq2), [q] "s"(q2));
    }
}

// L = 4 (32 rows per lane, 32 alignments per wavefront) is what large batches run; L = 8 and 16 are the same kernel with
// 16 / 8 rows per lane and 16 / 8 alignments per wavefront, for launches that do not fill the chip at L = 4.
// VARIANT: the cell body (the host picks it from the parameters, swmi_api.cpp make_config)
//   kPkQ0    every s + gap >= 0: no bias; the two diagonal adds of a row pair are ONE v_lshl_add_u64 (generated sweep)
//   kPkBias  any parameters: values shifted by Q (round 2's form, pk_two_rows above)
//   kPkVert  every s + 2 gap >= 0 (and the row offsets fit): vertical-offset form (generated sweep, gen_pk_sweeps.py)
// (round 2's unbiased form, pk_two_rows<false>, ran 2-3 % slower than kPkQ0 and is gone: profiles/r03_pk_variants_timing.txt)
enum PkVariant : int { kPkQ0 = 0, kPkBias = 1, kPkVert = 2 };

template <int MODE, int VARIANT, int L>
__global__ void __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(4)))
sw128_pk_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, int32_t *__restrict__ scores,
                uint32_t n, SmRows rows /* s + gap + Q (kPkVert: s + 2 gap), every byte in [0, 255] */, int gap, int q)
{
    static_assert(L == 4 || L == 8 || L == 16, "lanes per pair of alignments");
    constexpr bool kGenerated = VARIANT == kPkQ0 || VARIANT == kPkVert;
    constexpr int R = kSeqLen / L;          // rows per lane (L = lanes per pair of alignments)
    constexpr int G = 64 / L;               // lane groups per wavefront, each walking TWO alignments
    constexpr int A = 2 * G;                // alignments per wavefront
    constexpr int PAD = L + 2;
    constexpr int S = kSeqLen + 2 * PAD;
    constexpr int T2 = (kSeqLen + L - 1) / 2;               // T = 128 + L - 1 steps (odd): T2 pairs of steps and one more
    constexpr int NW = R / 4;
    constexpr uint32_t kPadCode = 16u | (16u << 16);        // both halves point at the all-zero table

    // per column of an alignment pair: the LDS byte offsets (into lds_tab) of the column's score table for X (low half) and
    // for Y (high half)
    __shared__ uint32_t lds_col[kWavesPerBlock][G * S];
    // lds_tab[b], b < 4: the scores of column base b against the four row bases, one per byte (the transposed matrix);
    // lds_tab[4] = 0 for pad columns.  The same for every wavefront of the block, which all write it (identical values).
    __shared__ uint32_t lds_tab[8];

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane / L, j = lane % L;
    const uint32_t first = (blockIdx.x * kWavesPerBlock + wv) * A;
    if (first >= n) return;                 // wave-uniform
    uint32_t px = first + 2 * g, py = px + 1;                // the lane group's two alignments
    const bool live_x = px < n, live_y = py < n;
    if (!live_x) px = n - 1;                // ragged tail: recompute the last pair, do not store
    if (!live_y) py = n - 1;

    uint32_t ax[NW], bx[NW], ay[NW], by[NW];
    if constexpr (MODE == 1) {
        load_bases_packed<R>(seq1s + (size_t)px * 32 + j * (R / 4), ax);
        load_bases_packed<R>(seq2s + (size_t)px * 32 + j * (R / 4), bx);
        load_bases_packed<R>(seq1s + (size_t)py * 32 + j * (R / 4), ay);
        load_bases_packed<R>(seq2s + (size_t)py * 32 + j * (R / 4), by);
    } else {
        load_bases<R>(seq1s + (size_t)px * kSeqLen + j * R, ax);
        load_bases<R>(seq2s + (MODE == 2 ? (size_t)0 : (size_t)px * kSeqLen) + j * R, bx);
        load_bases<R>(seq1s + (size_t)py * kSeqLen + j * R, ay);
        load_bases<R>(seq2s + (MODE == 2 ? (size_t)0 : (size_t)py * kSeqLen) + j * R, by);
    }

    if (lane < 5) {
        uint32_t c = 0;
        if (lane < 4) {
#pragma unroll
            for (int a = 0; a < 4; ++a) c |= ((rows.r[a] >> (8 * lane)) & 0xFFu) << (8 * a);
        }
        lds_tab[lane] = c;
    }
    uint32_t *prof = &lds_col[wv][g * S];
    for (int k = j; k < PAD; k += L) {
        prof[k] = kPadCode;
        prof[PAD + kSeqLen + k] = kPadCode;
    }
    // Column codes and row selectors, four at a time: the bases of four rows / columns sit in the four bytes of a dword, so
    // one mask (and one shift / or) per DWORD prepares all four, and one v_perm_b32 per entry then puts X's byte into byte 0
    // and Y's byte into byte 2 (round 2 extracted, shifted and merged every entry on its own: ~5 instructions each).
    //   column code  = (4 * base of X's column) | (4 * base of Y's column) << 16   (byte offsets into lds_tab)
    //   row selector = byte 0 <- byte (base of X's row) of X's column table, byte 2 <- byte (base of Y's row) of Y's column
    //                  table (source bytes 4..7), bytes 1 and 3 <- 0x00 (selector 0x0C):  ra | (4 + rb) << 16 | 0x0C000C00
    uint32_t rsel[R];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t cx4 = (bx[w] & 0x03030303u) << 2, cy4 = (by[w] & 0x03030303u) << 2;
        const uint32_t ra4 = ax[w] & 0x03030303u, rb4 = (ay[w] & 0x03030303u) | 0x04040404u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t pick = (uint32_t)k | 0x0C000C00u | ((uint32_t)(4 + k) << 16);     // byte k of the first, byte k of the second
            prof[PAD + j * R + 4 * w + k] = __builtin_amdgcn_perm(cy4, cx4, pick);
            rsel[4 * w + k] = __builtin_amdgcn_perm(rb4, ra4, pick) | 0x0C000C00u;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_s_setprio(2);

    keep_f16_denormals();

    // Two copies of the previous column, X in the low half and Y in the high half of every register:
    //   hq[i] = H(row i) + Q   what `left` and `up` are (the max3 runs on values shifted by Q, which keeps the diagonal term
    //                          t = H(i-1,j-1) + (s + gap + Q) non-negative without a second saturating subtraction)
    //   hu[i] = H(row i)       what the diagonal term starts from
    // Adding Q is a plain 32-bit add on the packed pair: every half stays in [0, 0x7C00), so nothing carries from the low half
    // into the high one.  (Q = 0 is VARIANT kPkQ0: one copy, the generated sweep.)
    const uint32_t gq2 = (uint32_t)(gap + q) | ((uint32_t)(gap + q) << 16);
    const uint32_t q2 = (uint32_t)q | ((uint32_t)q << 16);
    uint32_t best = 0;
    const int group_mask = keep(j == 0 ? 0 : -1);
    const uint32_t *col = prof + PAD - j;   // col[t] = table offsets of column t - j
    auto tables = [&](uint32_t code, uint32_t &cx, uint32_t &cy) {
        cx = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds_tab) + (code & 0xFFFFu));
        cy = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds_tab) + (code >> 16));
    };
    // the same for column t, with the two halves of the code fetched as two 16-bit LDS reads: no VALU work per step
    auto tables_at = [&](int t, uint32_t &cx, uint32_t &cy) {
        const uint16_t *half = reinterpret_cast<const uint16_t *>(col + t);
        cx = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds_tab) + half[0]);
        cy = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(lds_tab) + half[1]);
    };

    if constexpr (kGenerated) {
        // The sweep as generated code (gen_pk_sweeps.py -> pk_sweeps_gen.inc): the column in explicit register variables so
        // that two rows share a 64-bit add; one asm statement per instruction, issued in the order written.
#define pk_best best
#define pk_dval(r) ((uint32_t)(gap * ((r) + 1)) * 0x10001u)      /* D_r = gap (r + 1) in both halves (kPkVert) */
        if constexpr (VARIANT == kPkQ0 && R == 32) {
#define PK_SECTION 32
#include "pk_sweeps_gen.inc"
#undef PK_SECTION
        } else if constexpr (VARIANT == kPkQ0 && R == 16) {
#define PK_SECTION 16
#include "pk_sweeps_gen.inc"
#undef PK_SECTION
        } else if constexpr (VARIANT == kPkQ0 && R == 8) {
#define PK_SECTION 8
#include "pk_sweeps_gen.inc"
#undef PK_SECTION
        } else if constexpr (VARIANT == kPkVert && R == 32) {
#define PK_SECTION 232
#include "pk_sweeps_gen.inc"
#undef PK_SECTION
        } else if constexpr (VARIANT == kPkVert && R == 16) {
#define PK_SECTION 216
#include "pk_sweeps_gen.inc"
#undef PK_SECTION
        } else {
#define PK_SECTION 208
#include "pk_sweeps_gen.inc"
#undef PK_SECTION
        }
#undef pk_best
#undef pk_dval
    } else {
    uint32_t hq[R], hu_[R];
#pragma unroll
    for (int i = 0; i < R; ++i) { hq[i] = q2; hu_[i] = 0; }
    uint32_t u0 = q2, u1 = q2;
    const uint32_t edge = (uint32_t)keep((int)(j == 0 ? q2 : 0u));   // above the first row of an alignment: H = 0

    auto step = [&](uint32_t cx, uint32_t cy, uint32_t up, uint32_t diag) {
        const uint32_t d0 = diag - q2;                                          // H of the diagonal neighbour of row 0
        uint32_t t = d0 + __builtin_amdgcn_perm(cy, cx, rsel[0]);               // diagonal term of row 0
        uint32_t sc = __builtin_amdgcn_perm(cy, cx, rsel[1]);                   // scores of row 1
#pragma unroll
        for (int i = 0; i < R; i += 2) {
            if (i + 2 < R) pk_two_rows<false>(hq[i], hq[i + 1], hu_[i], hu_[i + 1], best, t, sc, up, rsel[i + 2], rsel[i + 3], cx, cy, gq2, q2);
            else           pk_two_rows<true>(hq[i], hq[i + 1], hu_[i], hu_[i + 1], best, t, sc, up, 0, 0, cx, cy, gq2, q2);
            up = hq[i + 1];
        }
        asm volatile("s_nop 1");            // the DPP below reads a register the asm above wrote: 2 wait states, by hand
        uint32_t out = (uint32_t)from_prev_lane<L>((int)hq[R - 1], group_mask);
        return out | edge;
    };
    uint32_t c1 = col[1], x0, y0;
    tables(col[0], x0, y0);
    for (int t2 = 0; t2 < T2; ++t2) {
        const uint32_t c2 = col[2 * t2 + 2], c3 = col[2 * t2 + 3];
        uint32_t x1, y1, x2, y2;
        tables(c1, x1, y1);
        u1 = step(x0, y0, u0, u1);
        tables(c2, x2, y2);
        u0 = step(x1, y1, u1, u0);
        x0 = x2; y0 = y2; c1 = c3;
    }
    step(x0, y0, u0, u1);                   // step 128 + L - 2, the last lane's last column
    }

    // reduce over the L lanes of the group (packed), then unfold: the maximum was tracked on x + Q = H + gap + Q (kPkVert: on H)
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
        const uint32_t other = (uint32_t)__shfl_xor((int)best, o);
        best = pk_max3(best, other, other);
    }
    const int bx_score = (int)(best & 0xFFFFu), by_score = (int)(best >> 16), gq = VARIANT == kPkVert ? 0 : gap + q;
    if (j == 0) {
        if (live_x) scores[px] = bx_score > gq ? bx_score - gq : 0;
        if (live_y) scores[py] = by_score > gq ? by_score - gq : 0;
    }
}

// ---- LDS score fetch for the lookup variant ---------------------------------------------------------
// ds_read_i8 returns the sign-extended byte, so the consumer is a plain full-rate v_add_u32.  (Left to itself hipcc
// turns half of these into ds_read_u8 + v_add_u32_sdwa, a half-rate add.)  The loads are inline asm, so the compiler
// does not count them: land_scores() is the matching wait -- one s_waitcnt, then an empty asm per register that makes
// every consumer depend on it (cdna_hip_programming.md 5.7, form ii).
template <int ROW>
__device__ __forceinline__ void fetch_score(int &dst, uint32_t lds_addr)
{
    asm volatile("ds_read_i8 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(ROW * 256));
}
template <int R, int... I>
__device__ __forceinline__ void fetch_scores_impl(int (&dst)[R], uint32_t lds_addr, std::integer_sequence<int, I...>)
{
    (fetch_score<I>(dst[I], lds_addr), ...);
}
template <int R>
__device__ __forceinline__ void fetch_scores(int (&dst)[R], uint32_t lds_addr)
{
    fetch_scores_impl<R>(dst, lds_addr, std::make_integer_sequence<int, R>{});
}
template <int R>
__device__ __forceinline__ void land_scores(int (&s)[R])
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < R; ++i) asm volatile("" : "+v"(s[i]));
}

// ---- LDS-lookup variant ------------------------------------------------------------------------------
//
// Same recurrence (gap-folded body), but the score lookup leaves the VALU: every lane keeps the 4 int8 scores of each
// of its rows in an LDS table, and ONE ds_read_i8 per cell -- address = lane slot + column base, row selected by the
// instruction's immediate offset -- returns the sign-extended score.  The diagonal term is then a full-rate v_add_u32
// instead of a half-rate v_dot4 (4 -> 2 issue cycles per cell); the LDS pipe is otherwise idle in this kernel.
// Out-of-range columns (pipeline fill/drain) read a shared table of -128 bytes, which keeps not-yet-started rows at 0.
// Scores for step t+1 are fetched while step t computes (two register sets, loop unrolled by two).
template <int L, int MODE>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
sw128_lut_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, int32_t *__restrict__ scores,
                 uint32_t n, SmRows rows, int gap)
{
    constexpr int R = kSeqLen / L;
    constexpr int A = 64 / L;
    constexpr int PAD = L + 4;              // the unrolled loop may run one step past T and prefetches two ahead
    constexpr int S = kSeqLen + 2 * PAD;
    constexpr int T2 = (kSeqLen + L - 1 + 1) / 2;   // pairs of steps
    constexpr int NW = (R + 3) / 4;

    __shared__ int lds_colofs[kWavesPerBlock][A * S];            // per column: byte offset into the lane's score slot
    __shared__ uint32_t lds_q[kWavesPerBlock][R * 64];           // [row][lane] -> 4 x int8 scores (+ gap)
    __shared__ uint32_t lds_neg[R * 64];                         // all bytes -128: what pad columns read
    __shared__ uint32_t lds_rows[kWavesPerBlock][4];

    for (int k = threadIdx.x; k < R * 64; k += 64 * kWavesPerBlock) lds_neg[k] = 0x80808080u;
    __syncthreads();                        // before any wave may leave

    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int g = lane / L;
    const int j = lane % L;
    const uint32_t first_pair = (blockIdx.x * kWavesPerBlock + wv) * A;
    if (first_pair >= n) return;
    uint32_t pair = first_pair + g;
    const bool live = pair < n;
    if (!live) pair = n - 1;

    uint32_t a_w[NW], b_w[NW];
    if constexpr (MODE == 1) {
        if constexpr (R == 2) {
            const uint32_t sh = 4 * (j & 1);
            a_w[0] = spread4((seq1s[(size_t)pair * 32 + (j >> 1)] >> sh) & 0xf);
            b_w[0] = spread4((seq2s[(size_t)pair * 32 + (j >> 1)] >> sh) & 0xf);
        } else {
            load_bases_packed<R>(seq1s + (size_t)pair * 32 + j * (R / 4), a_w);
            load_bases_packed<R>(seq2s + (size_t)pair * 32 + j * (R / 4), b_w);
        }
    } else {
        load_bases<R>(seq1s + (size_t)pair * kSeqLen + j * R, a_w);
        load_bases<R>(seq2s + (MODE == 2 ? (size_t)0 : (size_t)pair * kSeqLen) + j * R, b_w);
    }

    if (lane < 4) lds_rows[wv][lane] = rows.r[lane];
    // byte distance from this wave's score table to the shared -128 table (same [row][lane] indexing)
    const int neg_ofs = (int)((const char *)lds_neg - (const char *)lds_q[wv]);
    int *prof = &lds_colofs[wv][g * S];
    for (int k = j; k < PAD; k += L) {
        prof[k] = neg_ofs;
        prof[PAD + kSeqLen + k] = neg_ofs;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) prof[PAD + j * R + i] = (int)((b_w[i / 4] >> (8 * (i % 4))) & 3u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const uint32_t a = (a_w[i / 4] >> (8 * (i % 4))) & 3u;
        lds_q[wv][i * 64 + lane] = lds_rows[wv][a];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // LDS byte address of this lane's slot in row 0 of the score table; + row * 256 (immediate) + column offset
    const uint32_t qlane = (uint32_t)(uintptr_t)(&lds_q[wv][lane]);
    const int *col = prof + PAD - j;

    int h[R];
#pragma unroll
    for (int i = 0; i < R; ++i) h[i] = 0;
    int best = 0;
    int u0 = 0, u1 = 0;                     // alternating: value from lane j-1 for this step / the step before
    const int group_mask = keep(j == 0 ? 0 : -1);

    int sA[R], sB[R];
    fetch_scores<R>(sA, qlane + (uint32_t)col[0]);
    int ofs_next = col[1];

    auto step = [&](const int (&sc)[R], int up, int diag) {
        int dprev = diag;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int left = h[i];
            const int tsum = dprev + sc[i];
            const int lu = left > up ? left : up;
            const int x = lu > tsum ? lu : tsum;
            best = best > x ? best : x;
            const int hn = sat_sub<false>(x, gap);
            h[i] = hn;
            up = hn;
            dprev = left;
        }
        return from_prev_lane<L>(up, group_mask);
    };

    for (int t2 = 0; t2 < T2; ++t2) {
        {   // step 2*t2: scores in sA (fetched one step ago); start fetching sB for step 2*t2+1
            land_scores<R>(sA);
            fetch_scores<R>(sB, qlane + (uint32_t)ofs_next);
            ofs_next = col[2 * t2 + 2];
            u1 = step(sA, u0, u1);          // u1 now: from lane j-1 for step 2*t2+1; u0 is its diagonal
        }
        {   // step 2*t2+1
            land_scores<R>(sB);
            fetch_scores<R>(sA, qlane + (uint32_t)ofs_next);
            ofs_next = col[2 * t2 + 3];
            u0 = step(sB, u1, u0);
        }
    }
    land_scores<R>(sA);                     // drain the last prefetch before the registers are reused

#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) {
        const int other = __shfl_xor(best, o);
        best = best > other ? best : other;
    }
    best = best > gap ? best - gap : 0;
    if (j == 0 && live) scores[pair] = best;
}

// ---- banded affine-gap scorer (BASELINE.json configs[4]: 1024 x 1024, band 128) --------------------------
//
// An extension without a reference counterpart (the reference has linear gaps only); semantics = oracle/sw_oracle.c
// sw_oracle_banded_affine():  E/F/H Gotoh recurrences on the 128 diagonals -64 <= j - i <= 63, local (floor 0).
//
// One wavefront walks ONE alignment's anti-diagonals: lane m owns the two adjacent diagonals 2m and 2m+1 of the band and
// alternates between them, so every lane computes exactly one in-band cell per step (a lane-per-diagonal layout would
// idle half the lanes on every anti-diagonal).  Step 2u computes cell (i, j) on diagonal 2m, step 2u+1 cell (i, j+1) on
// diagonal 2m+1, with i = 33 - m + u, j = u + m - 31 (1-based).  Neighbours:
//   even step: left = lane m-1's odd diagonal (DPP wave_shr:1), up = the lane's own odd diagonal
//   odd  step: left = the lane's own even diagonal,             up = lane m+1's even diagonal (DPP wave_shl:1)
// The row scores (4 x int8 per base of seq1) and the column one-hots (seq2) are staged per alignment in LDS, with pad
// entries either side; every lane reads the entry of its own row / column, one of each per two cells.
// E and F are kept saturated at 0 (v_sub_u32 clamp): max(0, E) is all H ever needs, and it makes every out-of-band or
// out-of-matrix neighbour (which the DPP shifts and the pad entries deliver as 0) behave as -infinity.  Cells past the end
// of either sequence keep computing; their values are bounded by the running maximum (open, ext >= 0), so nothing is masked.
// Value of lane m-1 / m+1 minus `ext`, NOT saturated: the DPP shift rides on the subtraction (v_sub*_dpp, one half-rate
// instruction instead of v_mov_b32_dpp + v_sub_u32 clamp).  An invalid source lane reads 0, so the band's edge lanes get
// -ext: a negative E / F acts as the "-infinity" the saturated form expressed as 0 (every consumer takes a max with a
// value >= 0), which is why the result may stay signed.
__device__ __forceinline__ int below_minus(int v, int ext)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true) - ext;
}
__device__ __forceinline__ int above_minus(int v, int ext)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true) - ext;
}

// kI16: every H of the alignment stays below 2^15 (len * max score < 32768, decided on the host), so the two maxes that
// form the values handed to the neighbours run as full-rate v_max_i16 (v_max_i32 is half rate on gfx950).
// s_nop after the full-rate instruction that is followed by a half-rate one (the packed kernel's finding):
// bit 0 = after the plain saturating subtraction, bit 3 = after the second v_max_i16.  1.578 -> 1.538 ms at 65 536 x 1024.
#ifndef SWMI_BA_NOPS
#define SWMI_BA_NOPS 9
#endif
#define BA_NOP(bit, v) do { if constexpr ((SWMI_BA_NOPS >> (bit)) & 1) asm volatile("s_nop 0" : "+v"(v)); } while (0)
template <bool kOpenGeExt, bool kI16>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
sw_banded_affine_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, int32_t *__restrict__ scores,
                        uint32_t n, int len, SmRows rows, int gap_open, int gap_ext)
{
    extern __shared__ uint32_t lds_dyn[];       // per wave: row scores [len + 72] then column one-hots [len + 72], 32 pads in front
    __shared__ uint32_t lds_rows[kWavesPerBlock][4];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t pair = blockIdx.x * kWavesPerBlock + wv;
    if (pair >= n) return;                      // wave-uniform; only wave-level synchronisation below
    const int stride = len + 72;
    uint32_t *arow = lds_dyn + (size_t)wv * 2 * stride;
    uint32_t *boh = arow + stride;

    if (lane < 4) lds_rows[wv][lane] = rows.r[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t *s1 = seq1s + (size_t)pair * (size_t)len;
    const uint8_t *s2 = seq2s + (size_t)pair * (size_t)len;
    for (int kp = lane; kp < stride; kp += 64) {                // entry kp holds position kp - 32 of the sequence
        const int k = kp - 32;
        const bool in = (unsigned)k < (unsigned)len;
        const uint32_t a = in ? (s1[k] & 3u) : 0u;
        const uint32_t b = in ? (s2[k] & 3u) : 0u;
        arow[kp] = in ? lds_rows[wv][a] : 0x80808080u;    // pad rows score -128 against everything
        boh[kp] = in ? (1u << (8u * b)) : 0u;             // pad columns select nothing
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // In iteration u lane m sits at row i = 33 - m + u (both steps) and column j = u + m - 31 (even step) / j + 1 (odd
    // step), 1-based.  Each lane reads its own row scores and one-hots from LDS, one entry of each per iteration and one
    // iteration ahead (the LDS pipe is otherwise idle; streaming them through the lanes cost one DPP move per cell).
    const uint32_t *pa = arow + (64 - lane);    // pa[u] = row scores of row i
    const uint32_t *pb = boh + lane;            // pb[u] = one-hot of column j, pb[u + 1] of column j + 1
    int a_cur = (int)pa[0];
    int b_cur = (int)pb[0];
    int best = 0;

    // Both gap recurrences are FOLDED so that a cell hands each neighbour ONE value and only one of the two crosses lanes.
    //   open >= ext (the usual case): with hm = max(H - (open - ext), 0)
    //       E = max(E_left, hm_left) - ext,   F = max(F_up, hm_up) - ext
    //     (saturating subtraction distributes over max, and (H -sat (open-ext)) -sat ext = H -sat open): the cell hands
    //     me = max(E, hm) to the right and mf = max(F, hm) downwards; per cell dot4 + max3 + sub_dpp + 2 sub + 2 max.
    //   open < ext (a gap's first position is the cheap one): with em = E - (ext - open), fm = F - (ext - open)
    //       E = max(em_left, H_left) - open,   F = max(fm_up, H_up) - open
    //     the cell hands me = max(em, H) and mf = max(fm, H); one more subtraction per cell (round 2 ran this case unfolded:
    //     two lane-crossing moves, four saturating subtractions, two maxes).
    // The crossing value is subtracted without saturation by a v_sub_dpp (the lane shift is free on it); it may come out
    // negative, which is harmless: it only feeds max3(t, f, e) beside a saturated partner (>= 0) and a max beside a value
    // >= 0 (hm resp. H).  em / fm are plain differences for the same reason.
    {
        const int step_cost = kOpenGeExt ? gap_ext : gap_open;           // what the value handed over loses on arrival
        const int fold = kOpenGeExt ? gap_open - gap_ext : gap_ext - gap_open;
        // gfx9 DPP encodings take no SGPR operand: the subtrahend has to sit in a VGPR for hipcc to fold the lane shift into
        // the subtraction (v_subrev_u32_dpp); from an SGPR it emits v_mov_b32_dpp + v_subrev_u32 instead
        const int cost_v = keep(step_cost);
        int h0 = 0, me0 = 0, mf0 = 0;           // last cell on the even diagonal 2m
        int h1 = 0, me1 = 0, mf1 = 0;           // last cell on the odd diagonal 2m+1
        auto hand_over = [&](int h, int e, int f, int &me, int &mf) {
            if constexpr (kOpenGeExt) {
                int hm = sat_sub<false>(h, fold);
                BA_NOP(1, hm);
                me = vmax<kI16>(e, hm);
                BA_NOP(2, me);
                mf = vmax<kI16>(f, hm);
                BA_NOP(3, mf);
            } else {
                const int em = e - fold, fm = f - fold;                  // may be negative: they only meet h >= 0 in a max
                me = vmax<kI16>(em, h);
                mf = vmax<kI16>(fm, h);
                BA_NOP(3, mf);
            }
        };
        auto pair_of_steps = [&](int b_next) {
            {   // even step: diagonal 2m, cell (i, j); left = lane m-1's odd diagonal, up = own odd diagonal
                const int e = below_minus(me1, cost_v);
                int f = sat_sub<false>(mf1, step_cost);
                BA_NOP(0, f);
                const int t = __builtin_amdgcn_sdot4(a_cur, b_cur, h0, true);
                const int tf = t > f ? t : f;
                h0 = tf > e ? tf : e;           // v_max3_i32; >= 0 because f >= 0
                hand_over(h0, e, f, me0, mf0);
            }
            b_cur = b_next;
            {   // odd step: diagonal 2m+1, cell (i, j+1); left = own even diagonal, up = lane m+1's even diagonal
                int e = sat_sub<false>(me0, step_cost);
                BA_NOP(0, e);
                const int f = above_minus(mf0, cost_v);
                const int t = __builtin_amdgcn_sdot4(a_cur, b_cur, h1, true);
                const int te = t > e ? t : e;
                h1 = te > f ? te : f;           // >= 0 because e >= 0
                hand_over(h1, e, f, me1, mf1);
            }
            const int hb = h0 > h1 ? h0 : h1;
            best = best > hb ? best : hb;
        };
        int u = 0;
#pragma unroll 1
        for (; u + 4 <= len; u += 4) {          // four iterations per trip: LDS offsets become immediates, no register rotation
            const int b1 = (int)pb[u + 1], b2 = (int)pb[u + 2], b3 = (int)pb[u + 3], b4 = (int)pb[u + 4];
            const int a1 = (int)pa[u + 1], a2 = (int)pa[u + 2], a3 = (int)pa[u + 3], a4 = (int)pa[u + 4];
            pair_of_steps(b1); a_cur = a1;
            pair_of_steps(b2); a_cur = a2;
            pair_of_steps(b3); a_cur = a3;
            pair_of_steps(b4); a_cur = a4;
        }
        for (; u < len; ++u) {
            const int b_next = (int)pb[u + 1];
            const int a_next = (int)pa[u + 1];
            pair_of_steps(b_next);
            a_cur = a_next;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(best, o);
        best = best > other ? best : other;
    }
    if (lane == 0) scores[pair] = best;
}

// ---- banded affine, TWO alignments per wavefront (round 4) ---------------------------------------------------------
//
// The 128 x 128 scorer's packing (section 5a of DESIGN.md) applied to the band: one register holds the SAME cell of two
// alignments X (low half) and Y (high half), so the two share every instruction, and v_pk_maximum3_f16 -- a packed 3-way
// INTEGER max on halves below 0x7C00 -- does per instruction what two v_max3_i32 did.  Same mapping as the int32 kernel above
// (lane m owns diagonals 2m, 2m + 1 and alternates), same folding of the gap recurrences; what changes is the arithmetic:
//   * every half is unsigned.  With B = max(0, -min s) the lookup yields s + B >= 0 and E, F travel as E + B, F + B:
//         M = max3(H_diag + (s + B), E + B, F + B),   H = M -sat B                      (the zero floor of local alignment)
//     A half that saturates at 0 stands for the true value -B or less; every non-positive E / F / hm is equivalent (H floors
//     at 0 and E' = max(E, hm) - ext is monotone), so the clamps change nothing -- the argument of the int32 kernel, shifted by B.
//   * lookup of both halves = ONE v_perm_b32: its eight source bytes are the score rows of X's and Y's ROW base (s + B, one
//     byte per column base; staged in LDS per row), its selector comes from LDS per COLUMN (byte 0 <- X's column base, byte 2
//     <- 4 + Y's, the other bytes constant zero).  Pads select nothing / hold zero rows: s + B = 0, the lowest score.
//   * the two diagonal terms of an iteration (even and odd step) are ONE v_lshl_add_u64 over two aligned register pairs.
//   * the value handed to a neighbour is floored by the max3's spare operand:  me = max3(E, hm, C)  with C >= the cost the
//     receiver subtracts, so that subtraction is a plain 32-bit one (no half borrows) -- v_sub_u32, and for the value that
//     crosses lanes v_sub_u32_dpp with the lane shift riding on it (VOP3P has no DPP form; a v_pk_sub_u16 clamp would need a
//     v_mov_b32_dpp in front).  The band's edge lanes, whose DPP source does not exist and reads 0, subtract 0.
//       open >= ext:  hm = M -sat (open - ext);  me = max3(E, hm, ext);             E' = me - ext
//       open <  ext:  em = E -sat (ext - open);  me = max3(em, M, max(B, open));    E' = me - open     (M stands in for H + B:
//                     the floor max(B, open) >= B repairs M < B, and it is <= B + open, i.e. E' <= 0 in true terms)
//   Per cell and PAIR of alignments: perm, 1/2 paired add, max3, 2 sat-sub, 2 max3, 2 sub, 1/2 max3 (running best) = 9
//   instructions (10 for open < ext) = 4.5 per alignment-cell against the int32 kernel's 8.25.
// Domain (decided on the host, launch_banded_affine): len * max(s, 0) + 2 B + open + ext + 64 < 0x7C00; else the int32 kernel.
__device__ __forceinline__ unsigned ba_pk_max3(unsigned a, unsigned b, unsigned c)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(h2, a), __builtin_bit_cast(h2, b)), __builtin_bit_cast(h2, c));
    return __builtin_bit_cast(unsigned, m);
}
__device__ __forceinline__ unsigned ba_pk_sub_sat(unsigned v, unsigned d)
{
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, v), __builtin_bit_cast(us2, d)));
}

struct BandedPkParams {
    uint32_t rows[4];          // rows[a]: s(a, b) + B in byte b
    uint32_t bias2;            // B in both halves
    uint32_t fold2;            // |open - ext| in both halves
    uint32_t cost2;            // what a handed-over value loses on arrival (ext resp. open) in both halves
    uint32_t delta2;           // B - cost per half as ONE 32-bit addend (see the kernel): what a value gains crossing an iteration
    uint32_t best_step;        // B in both halves, less the carry the low dword of the paired add always produces when B < cost
};

// The bias GROWS: a cell computed in iteration u of a trip carries the bias beta_u = (u + 1) B, so that
//     M = max3(H_diag' + (s + B), E', F')        IS the next diagonal term's H' -- no "M -sat B" per cell:
// H_diag' has the bias of the iteration before, the lookup adds B, and E', F' arrive with beta_u because
//   * a value handed over inside an iteration loses `cost` (v_sub_u32 / v_sub_u32_dpp, as before),
//   * a value handed over ACROSS an iteration gains B - cost: one addend per dword, d | d << 16 for d >= 0 and
//     (-|d| & 0xFFFF) | ((-|d| - 1) & 0xFFFF) << 16 for d < 0 -- each half is >= |d| (the floors), so the low half always
//     carries into the high one and the -1 absorbs it; the own-lane one of the two is the low dword of a v_lshl_add_u64
//     whose high dword moves the running best along by B (best_step: B less that ever-present carry);
//   * the max3 that forms a hand-over value floors it at beta_u + cost (an SGPR that steps by B per iteration), i.e. at the
//     true value `cost`: after the receiver's subtraction E', F' >= beta_u -- true zero -- and since every cell takes one of
//     E', F' from its own lane, M >= beta_u: the zero floor of local alignment without an instruction.
// Every kTrip iterations the seven state registers go back by kTrip * B (plain subtractions: all of them are >= that).
// Per cell and pair of alignments: perm, 1/2 paired add, max3, sat-sub (hm), 2 max3, 1/2 + 1/2 (paired) + 1 add / sub,
// 1/2 max3 (best) = 8 (9 for open < ext), + 7 / (2 kTrip) for the trip's rebase.
constexpr int kBandedTrip = 16;
template <bool kOpenGeExt>
__global__ void __launch_bounds__(64 * kWavesPerBlock)
sw_banded_affine_pk_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, int32_t *__restrict__ scores,
                           uint32_t n, int len, BandedPkParams prm)
{
    // per wave: row entries [len + 72] x 2 dwords (X's, Y's score row) then column selectors [len + 72], 32 pads in front
    extern __shared__ uint32_t lds_dyn[];
    __shared__ uint32_t lds_rows[kWavesPerBlock][4];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const uint32_t wave_id = blockIdx.x * kWavesPerBlock + wv;
    const uint32_t pair_x = 2 * wave_id;
    if (pair_x >= n) return;                    // wave-uniform; only wave-level synchronisation below
    const uint32_t pair_y = pair_x + 1 < n ? pair_x + 1 : pair_x;          // an odd batch's last wavefront scores its pair twice
    const int stride = len + 72;
    uint2 *arow = reinterpret_cast<uint2 *>(lds_dyn + (size_t)wv * 3 * stride);
    uint32_t *bsel = lds_dyn + (size_t)wv * 3 * stride + 2 * stride;

    keep_f16_denormals();
    if (lane < 4) lds_rows[wv][lane] = prm.rows[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint8_t *x1 = seq1s + (size_t)pair_x * (size_t)len, *x2 = seq2s + (size_t)pair_x * (size_t)len;
    const uint8_t *y1 = seq1s + (size_t)pair_y * (size_t)len, *y2 = seq2s + (size_t)pair_y * (size_t)len;
    for (int kp = lane; kp < stride; kp += 64) {                // entry kp holds position kp - 32 of the sequences
        const int k = kp - 32;
        const bool in = (unsigned)k < (unsigned)len;
        const uint32_t ax = in ? (x1[k] & 3u) : 0u, ay = in ? (y1[k] & 3u) : 0u;
        const uint32_t bx = in ? (x2[k] & 3u) : 0u, by = in ? (y2[k] & 3u) : 0u;
        arow[kp] = in ? make_uint2(lds_rows[wv][ax], lds_rows[wv][ay]) : make_uint2(0u, 0u);      // pad rows: s + B = 0
        bsel[kp] = in ? (0x0C000C00u | bx | ((4u + by) << 16)) : 0x0C0C0C0Cu;                      // pad columns select zero
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // iteration u: lane m at row i = 33 - m + u, columns j = u + m - 31 (even step) and j + 1 (odd step), 1-based
    const uint2 *pa = arow + (64 - lane);       // pa[u] = score rows of row i
    const uint32_t *pb = bsel + lane;           // pb[u] = selector of column j, pb[u + 1] of column j + 1
    uint2 a_cur = pa[0];
    uint32_t b_cur = pb[0];
    const unsigned bias2 = prm.bias2, fold2 = prm.fold2, cost2 = prm.cost2;
    // the addend / subtrahend of the lane-crossing operations sits in a VGPR (gfx9 DPP takes no SGPR operand) and is 0 in the
    // lane whose source lane does not exist (that lane's DPP read yields 0, and 0 stands for "nothing": every cell takes
    // its other gap operand from its own lane)
    const unsigned delta_left = lane == 0 ? 0u : prm.delta2, cost_up = lane == 63 ? 0u : cost2;
    const unsigned long long own_step = (unsigned long long)prm.delta2 | ((unsigned long long)prm.best_step << 32);
    unsigned best = 0;
    unsigned h0 = 0, h1 = 0;                    // H' of the last cell on diagonals 2m / 2m + 1 (X low, Y high)
    unsigned me0 = cost2, mf0 = cost2, me1 = cost2, mf1 = cost2;           // hand-over values at their floor (true value: cost)
    auto hand_over = [&](unsigned m, unsigned e, unsigned f, unsigned floor2, unsigned &me, unsigned &mf) {
        if constexpr (kOpenGeExt) {
            const unsigned hm = ba_pk_sub_sat(m, fold2);
            me = ba_pk_max3(e, hm, floor2);
            mf = ba_pk_max3(f, hm, floor2);
        } else {
            const unsigned em = ba_pk_sub_sat(e, fold2), fm = ba_pk_sub_sat(f, fold2);
            me = ba_pk_max3(em, m, floor2);
            mf = ba_pk_max3(fm, m, floor2);
        }
    };
    // one iteration = two anti-diagonal steps; floor2 = beta + cost of THIS iteration in both halves
    auto pair_of_steps = [&](uint32_t b_next, unsigned floor2) {
        // both diagonal terms at once: (h0, h1) + (lookup of (i, j), lookup of (i, j + 1)).  (Written as a C addition hipcc
        // splits it again, so the instruction is named; no half ever carries: every sum stays below 0x7C00.)
        const unsigned p0 = __builtin_amdgcn_perm(a_cur.y, a_cur.x, b_cur), p1 = __builtin_amdgcn_perm(a_cur.y, a_cur.x, b_next);
        unsigned long long t, fb;
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"((unsigned long long)h0 | ((unsigned long long)h1 << 32)),
                                                        "v"((unsigned long long)p0 | ((unsigned long long)p1 << 32)));
        const unsigned t0 = (unsigned)t, t1 = (unsigned)(t >> 32);
        {   // even step: diagonal 2m, cell (i, j); left = lane m-1's odd diagonal, up = own odd diagonal -- both from the
            // iteration before: + (B - cost); the own one shares its instruction with the running best's + B
            const unsigned e = (unsigned)__builtin_amdgcn_update_dpp(0, (int)me1, 0x138 /* wave_shr:1 */, 0xf, 0xf, true) + delta_left;
            asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(fb) : "v"((unsigned long long)mf1 | ((unsigned long long)best << 32)), "s"(own_step));
            const unsigned f = (unsigned)fb;
            best = (unsigned)(fb >> 32);
            h0 = ba_pk_max3(t0, e, f);
            hand_over(h0, e, f, floor2, me0, mf0);
        }
        {   // odd step: diagonal 2m+1, cell (i, j+1); left = own even diagonal, up = lane m+1's even diagonal (this iteration's)
            const unsigned e = me0 - cost2;
            const unsigned f = (unsigned)__builtin_amdgcn_update_dpp(0, (int)mf0, 0x130 /* wave_shl:1 */, 0xf, 0xf, true) - cost_up;
            h1 = ba_pk_max3(t1, e, f);
            hand_over(h1, e, f, floor2, me1, mf1);
        }
        best = ba_pk_max3(best, h0, h1);
        b_cur = b_next;
    };
    auto rebase = [&](unsigned back2) {         // every state register goes back by the trip's bias growth (all are >= it)
        h0 -= back2; h1 -= back2; me0 -= back2; mf0 -= back2; me1 -= back2; mf1 -= back2; best -= back2;
    };
    int u = 0;
#pragma unroll 1
    for (; u + kBandedTrip <= len; u += kBandedTrip) {
#pragma unroll
        for (int k = 0; k < kBandedTrip; ++k) {
            const uint32_t b_next = pb[u + k + 1];
            const uint2 a_next = pa[u + k + 1];
            pair_of_steps(b_next, cost2 + (unsigned)(k + 1) * bias2);
            a_cur = a_next;
        }
        rebase((unsigned)kBandedTrip * bias2);
    }
    for (; u < len; ++u) {                      // the last len % kTrip iterations: trips of one
        const uint32_t b_next = pb[u + 1];
        const uint2 a_next = pa[u + 1];
        pair_of_steps(b_next, cost2 + bias2);
        a_cur = a_next;
        rebase(bias2);
    }
    int bx = (int)(best & 0xFFFFu), by = (int)(best >> 16);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int ox = __shfl_xor(bx, o), oy = __shfl_xor(by, o);
        bx = bx > ox ? bx : ox;
        by = by > oy ? by : oy;
    }
    if (lane == 0) {
        scores[pair_x] = bx;
        if (pair_x + 1 < n) scores[pair_x + 1] = by;
    }
}

// ---- synthetic input generator (specification in include/swmi.h) -----------------------------------

__device__ __forceinline__ uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one thread = one 64-bit word = 32 bases = 32 output bytes
__global__ void __launch_bounds__(256)
generate_kernel(uint8_t *__restrict__ seq1s, uint8_t *__restrict__ seq2s, uint64_t n_words, uint64_t seed,
                uint64_t first_pair)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_words; id += stride) {
        const uint64_t w = id & 3, s = (id >> 2) & 1, k = id >> 3;
        const uint64_t ctr = ((first_pair + k) * 2 + s) * 4 + w;
        const uint64_t x = splitmix64(seed ^ (ctr * 0x9E3779B97F4A7C15ull));
        uint4 lo, hi;
        lo.x = spread4((uint32_t)(x >> 0) & 0xff);  lo.y = spread4((uint32_t)(x >> 8) & 0xff);
        lo.z = spread4((uint32_t)(x >> 16) & 0xff); lo.w = spread4((uint32_t)(x >> 24) & 0xff);
        hi.x = spread4((uint32_t)(x >> 32) & 0xff); hi.y = spread4((uint32_t)(x >> 40) & 0xff);
        hi.z = spread4((uint32_t)(x >> 48) & 0xff); hi.w = spread4((uint32_t)(x >> 56) & 0xff);
        uint4 *dst = reinterpret_cast<uint4 *>((s ? seq2s : seq1s) + k * kSeqLen + w * 32);
        dst[0] = lo;
        dst[1] = hi;
    }
}

// unpack() of source.cpp:1580-1583 for n sequences: one thread = 4 packed bytes -> 16 output bytes
__global__ void __launch_bounds__(256)
unpack_kernel(const uint8_t *__restrict__ packed, uint8_t *__restrict__ unpacked, uint64_t n_words)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_words; id += stride) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(packed)[id];
        uint4 o;
        o.x = spread4(v & 0xff); o.y = spread4((v >> 8) & 0xff);
        o.z = spread4((v >> 16) & 0xff); o.w = spread4(v >> 24);
        reinterpret_cast<uint4 *>(unpacked)[id] = o;
    }
}

template <int L, int MODE>
hipError_t launch_L(const LaunchConfig &cfg, const uint8_t *s1, const uint8_t *s2, int32_t *out, size_t n,
                    const SmRows &rows, int gap, hipStream_t stream)
{
    constexpr int A = 64 / L;
    const size_t waves = (n + A - 1) / A;
    const size_t blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffull || n > 0xffffffffull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
    const uint32_t n32 = (uint32_t)n;
    if constexpr (L == 4 || L == 8 || L == 16) {
        if (cfg.use_pk) {                   // two alignments per register (rows = s + gap + pk_bias, one byte each)
            const size_t waves_pk = (n + 2 * A - 1) / (2 * A);
            const dim3 grid_pk((unsigned)((waves_pk + kWavesPerBlock - 1) / kWavesPerBlock));
            switch (cfg.pk_variant) {
            case kPkQ0:    hipLaunchKernelGGL((sw128_pk_kernel<MODE, kPkQ0, L>), grid_pk, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap, 0); break;
            case kPkVert:  hipLaunchKernelGGL((sw128_pk_kernel<MODE, kPkVert, L>), grid_pk, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap, 0); break;
            default:       hipLaunchKernelGGL((sw128_pk_kernel<MODE, kPkBias, L>), grid_pk, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap, cfg.pk_bias); break;
            }
            return hipGetLastError();
        }
    }
    if constexpr (L <= 16 && L >= 4) {
        if (cfg.fold_gap && cfg.use_lut) {
            hipLaunchKernelGGL((sw128_lut_kernel<L, MODE>), grid, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap);
            return hipGetLastError();
        }
    }
    if (cfg.fold_gap) {
        if (cfg.use_i16) hipLaunchKernelGGL((sw128_kernel<L, true, true, MODE>), grid, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap);
        else             hipLaunchKernelGGL((sw128_kernel<L, true, false, MODE>), grid, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap);
    } else {
        if (cfg.use_i16) hipLaunchKernelGGL((sw128_kernel<L, false, true, MODE>), grid, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap);
        else             hipLaunchKernelGGL((sw128_kernel<L, false, false, MODE>), grid, block, cfg.extra_lds_bytes, stream, s1, s2, out, n32, rows, gap);
    }
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_mode(const LaunchConfig &cfg, const uint8_t *s1, const uint8_t *s2, int32_t *out, size_t n,
                       const SmRows &rows, int gap, hipStream_t stream)
{
    switch (cfg.lanes_per_alignment) {
    case 64: return launch_L<64, MODE>(cfg, s1, s2, out, n, rows, gap, stream);
    case 32: return launch_L<32, MODE>(cfg, s1, s2, out, n, rows, gap, stream);
    case 16: return launch_L<16, MODE>(cfg, s1, s2, out, n, rows, gap, stream);
    case 8:  return launch_L<8, MODE>(cfg, s1, s2, out, n, rows, gap, stream);
    case 4:  return launch_L<4, MODE>(cfg, s1, s2, out, n, rows, gap, stream);
    case 2:  return launch_L<2, MODE>(cfg, s1, s2, out, n, rows, gap, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool schedule_supported(int L)
{
    return L == 64 || L == 32 || L == 16 || L == 8 || L == 4 || L == 2;
}

hipError_t launch_score(const LaunchConfig &cfg, const uint8_t *d_seq1s, const uint8_t *d_seq2s, int32_t *d_scores,
                        size_t n, const SmRows &rows, int gap, bool packed, hipStream_t stream)
{
    // a launch covers at most 2^31 pairs; larger batches are split by the caller
    return packed ? launch_mode<1>(cfg, d_seq1s, d_seq2s, d_scores, n, rows, gap, stream)
                  : launch_mode<0>(cfg, d_seq1s, d_seq2s, d_scores, n, rows, gap, stream);
}

hipError_t launch_score_one_vs_many(const LaunchConfig &cfg, const uint8_t *d_seq1s, const uint8_t *d_seq2,
                                    int32_t *d_scores, size_t n_seq1, const SmRows &rows, int gap, hipStream_t stream)
{
    return launch_mode<2>(cfg, d_seq1s, d_seq2, d_scores, n_seq1, rows, gap, stream);
}

// Which kernel a banded-affine launch runs: 2 = packed (two alignments per wavefront), 1 = int32 cell with 16-bit maxes,
// 0 = int32 cell.  The packed kernel needs every half to stay a finite half-precision pattern.
int banded_affine_kernel_choice(int len, const SmRows &rows, int gap_open, int gap_ext, bool allow_i16, bool allow_pk)
{
    int top = 0, low = 0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const int v = (int)(int8_t)(rows.r[a] >> (8 * b));
            top = v > top ? v : top;
            low = v < low ? v : low;
        }
    const long long bias = -low;
    if (allow_pk && (long long)len * top + (kBandedTrip + 2) * bias + gap_open + gap_ext + 64 < 0x7C00) return 2;
    // 16-bit maxes are exact while no H can reach 2^15: H <= len * (largest score, at least 0)
    return (long long)len * top < 32768 && allow_i16 ? 1 : 0;
}

hipError_t launch_banded_affine(const uint8_t *d_seq1s, const uint8_t *d_seq2s, int32_t *d_scores, size_t n, int len,
                                const SmRows &rows, int gap_open, int gap_ext, hipStream_t stream, bool allow_i16, bool allow_pk)
{
    if (n == 0) return hipSuccess;
    const int choice = banded_affine_kernel_choice(len, rows, gap_open, gap_ext, allow_i16, allow_pk);
    const dim3 block(64 * kWavesPerBlock);
    if (choice == 2) {
        const size_t waves = (n + 1) / 2, blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
        if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
        const size_t lds = (size_t)kWavesPerBlock * 3 * (size_t)(len + 72) * sizeof(uint32_t);
        int low = 0;
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                const int v = (int)(int8_t)(rows.r[a] >> (8 * b));
                low = v < low ? v : low;
            }
        const unsigned bias = (unsigned)-low;
        BandedPkParams prm;
        for (int a = 0; a < 4; ++a) {
            uint32_t r = 0;
            for (int b = 0; b < 4; ++b) r |= (uint32_t)((int)(int8_t)(rows.r[a] >> (8 * b)) + (int)bias) << (8 * b);
            prm.rows[a] = r;
        }
        const bool oge = gap_open >= gap_ext;
        const unsigned fold = (unsigned)(oge ? gap_open - gap_ext : gap_ext - gap_open), cost = (unsigned)(oge ? gap_ext : gap_open);
        prm.bias2 = bias * 0x10001u;
        prm.fold2 = fold * 0x10001u;
        prm.cost2 = cost * 0x10001u;
        if (bias >= cost) {                     // B - cost >= 0: a plain per-half addend
            prm.delta2 = (bias - cost) * 0x10001u;
            prm.best_step = prm.bias2;
        } else {                                // negative: two's complement per half, the high half less the low half's carry
            const unsigned d = cost - bias;
            prm.delta2 = ((0x10000u - d) & 0xFFFFu) | (((0x10000u - d - 1u) & 0xFFFFu) << 16);
            prm.best_step = prm.bias2 - 1u;     // ... which also reaches the running best in the paired add
        }
        const dim3 grid((unsigned)blocks);
        if (oge) hipLaunchKernelGGL((sw_banded_affine_pk_kernel<true>), grid, block, lds, stream, d_seq1s, d_seq2s, d_scores, (uint32_t)n, len, prm);
        else     hipLaunchKernelGGL((sw_banded_affine_pk_kernel<false>), grid, block, lds, stream, d_seq1s, d_seq2s, d_scores, (uint32_t)n, len, prm);
        return hipGetLastError();
    }
    const size_t blocks = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    const size_t lds = (size_t)kWavesPerBlock * 2 * (size_t)(len + 72) * sizeof(uint32_t);
    const bool i16 = choice == 1;
    const dim3 grid((unsigned)blocks);
    if (gap_open >= gap_ext) {
        if (i16) hipLaunchKernelGGL((sw_banded_affine_kernel<true, true>), grid, block, lds, stream, d_seq1s, d_seq2s, d_scores, (uint32_t)n, len, rows, gap_open, gap_ext);
        else     hipLaunchKernelGGL((sw_banded_affine_kernel<true, false>), grid, block, lds, stream, d_seq1s, d_seq2s, d_scores, (uint32_t)n, len, rows, gap_open, gap_ext);
    } else {
        if (i16) hipLaunchKernelGGL((sw_banded_affine_kernel<false, true>), grid, block, lds, stream, d_seq1s, d_seq2s, d_scores, (uint32_t)n, len, rows, gap_open, gap_ext);
        else     hipLaunchKernelGGL((sw_banded_affine_kernel<false, false>), grid, block, lds, stream, d_seq1s, d_seq2s, d_scores, (uint32_t)n, len, rows, gap_open, gap_ext);
    }
    return hipGetLastError();
}

hipError_t launch_pk_max3_selftest(unsigned long long *d_counts, hipStream_t stream)
{
    const dim3 grid((kPkLimit + 255) / 256, (kPkLimit + kBPerThread - 1) / kBPerThread);
    hipLaunchKernelGGL(pk_max3_selftest_kernel, grid, dim3(256), 0, stream, d_counts);
    return hipGetLastError();
}

hipError_t launch_generate(uint8_t *d_seq1s, uint8_t *d_seq2s, size_t n, uint64_t seed, uint64_t first_pair,
                           hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const uint64_t n_words = (uint64_t)n * 8;
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(generate_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_seq1s, d_seq2s, n_words, seed,
                       first_pair);
    return hipGetLastError();
}

hipError_t launch_unpack(const uint8_t *d_packed, uint8_t *d_unpacked, size_t n_seqs, hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    const uint64_t n_words = (uint64_t)n_seqs * 8;   // 32 packed bytes = 8 dwords per sequence
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_packed, d_unpacked, n_words);
    return hipGetLastError();
}

}  // namespace swmi
