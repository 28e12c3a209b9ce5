// sg_kernels.hip -- gfx950 kernels for the reference's semi-global adaptive-band X-drop aligner
//     SemiGlobal_AdaptiveBanded_XDrop_111_32_70   (source.cpp:1836-1976; SIMD variants :1978-2725)
// SURVEY.md section 8f row N4.  Same results as the reference: (score, traceback from (0,0) to the best cell).
//
// Not a translation of the AVX2 variants (one 32-byte vector per anti-diagonal, u8 cells re-based every round,
// source.cpp:2099-2109).  Here:
//   * forward sweep: 32 lanes = the 32 cells of the band (BANDWIDTH, source.cpp:1848), two alignments per wavefront; cell values stay int32 with the
//     reference's +70 offset (0 = dropped).  The band's direction (right / down) is decided per alignment every round
//     from its two end lanes; the shifted neighbours come from cross-lane reads.  Instead of the reference's 4 MB table
//     of cell values per alignment (source.cpp:1876) the sweep stores, per round, only what the traceback needs:
//     a 2-bit predecessor code per lane (diag / up / left in the reference's own tie-break order :1962-1971) and the row
//     of the band's top lane -- 10 bytes per round instead of 128.
//   * traceback: one thread per alignment follows the codes back to (0,0) (twice: once to count, once to emit the
//     positions in ascending order, as the reference returns them).
#include "swmi_internal.h"

#include <cstdlib>

namespace swmi {
namespace {

constexpr int kLen = 16384;                 // std::array<uint8_t,16384>, source.cpp:1837-1838
constexpr int kXDrop = 70;                  // X_THRESHOLD, source.cpp:1848
constexpr int kMaxRound = 2 * (kLen + 1) - 1;   // MAX_ROUND, source.cpp:1875
// per-alignment row strides of the sweep's records, padded so that every alignment starts on a 64-byte line
constexpr int kCodeStride = (kMaxRound + 7) & ~7;        // uint2 entries (8 B): 262208 B per alignment
constexpr int kTopStride = (kMaxRound + 31) & ~31;       // uint16 entries: 65600 B per alignment
constexpr size_t kLaneSweepMinBatch = 32768;
constexpr size_t kLaneTracebackMinBatch = 32768; // from here on the lane-per-alignment sweep wins

// max over each row of 16 lanes, left in every lane of the row: four DPP butterflies (v_max_i32_dpp, no LDS crossbar)
__device__ __forceinline__ int row16_max(int v)
{
    int o;
    o = __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(0, v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true); v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(0, v, 0x141 /* row_half_mirror */, 0xf, 0xf, true);    v = v > o ? v : o;
    o = __builtin_amdgcn_update_dpp(0, v, 0x140 /* row_mirror */, 0xf, 0xf, true);         v = v > o ? v : o;
    return v;
}

__device__ __forceinline__ int sat_dec(int v)             // max(v - 1, 0) for v >= 0: v_sub_u32 ... clamp
{
    return (int)__builtin_elementwise_sub_sat((unsigned)v, 1u);
}

__device__ __forceinline__ int keep_opaque(int v)         // stops hipcc from turning `x & mask` into a v_cndmask
{
    asm volatile("" : "+v"(v));
    return v;
}

// codes[(a * kMaxRound + r) * 2 + {0,1}]: bit k of word 0 / word 1 = low / high bit of lane k's predecessor code
// (0 none or dropped, 1 diagonal, 2 up, 3 left); top_y[a * kMaxRound + r] = row of lane 31 in round r;
// summary[a] = {score, best_round, best_lane, rounds stored}
//
// A round is one long dependency chain and the kernel is bound by how many instructions it issues per round (8 waves
// per SIMD keep the issue port busy), so the body is branch-free and every cross-lane step is a DPP move or a v_readlane:
// an LDS-crossbar shuffle (__shfl*, ds_bpermute_b32) costs more than all of a round's arithmetic.  The predecessor codes
// fall out of three v_cmp masks combined on the scalar unit (the masks ARE the ballots).
__global__ void __launch_bounds__(256)
sg_forward_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, uint32_t n,
                  uint32_t *__restrict__ codes, uint16_t *__restrict__ top_y, int4 *__restrict__ summary)
{
    const int lane = threadIdx.x & 63;
    const int k = lane & 31;                              // lane of the band, as the reference numbers them
    const bool second = lane >= 32;                       // which of the wavefront's two alignments
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t a = wave * 2 + (second ? 1u : 0u);
    if (wave * 2 >= n) return;
    const bool real = a < n;
    if (!real) a = n - 1;                                 // odd tail: shadow the last alignment, store nothing
    const uint32_t seq_base = a * (uint32_t)kLen;         // n <= 2^18 alignments per launch: fits 32 bits
    uint2 *my_codes = reinterpret_cast<uint2 *>(codes) + (size_t)a * kCodeStride;
    uint16_t *my_top = top_y + (size_t)a * kTopStride;
    const int not_first = keep_opaque(k == 0 ? 0 : -1), not_last = keep_opaque(k == 31 ? 0 : -1);
    const bool writer = real && k == 0;

    int cur = k == 31 ? kXDrop : 0, hor = 0, ver = 0, dia = 0;
    int pos_x = 31;                                       // the reference's now_pos_x (31 leading pads); now_pos_y = round - (pos_x - 31)
    int best = kXDrop, best_round = 0, best_lane = 31;
    bool alive = true;
    int rounds = 1;
    if (writer) { my_codes[0] = make_uint2(0, 0); my_top[0] = 0; }

    for (int round = 1; round < kMaxRound; ++round) {
        if (!__any(alive)) break;
        // direction of each alignment's band: lane 0 against lane 31 (source.cpp:1895)
        const bool right_a = __builtin_amdgcn_readlane(cur, 0) < __builtin_amdgcn_readlane(cur, 31);
        const bool right_b = __builtin_amdgcn_readlane(cur, 32) < __builtin_amdgcn_readlane(cur, 63);
        const bool right = second ? right_b : right_a;
        const int from_above = __builtin_amdgcn_update_dpp(0, cur, 0x130 /* wave_shl:1 */, 0xf, 0xf, true) & not_last;   // cur[k+1]
        const int from_below = __builtin_amdgcn_update_dpp(0, cur, 0x138 /* wave_shr:1 */, 0xf, 0xf, true) & not_first;  // cur[k-1]
        dia = right ? ver : hor;                          // :1897 / :1908
        const int nh = right ? cur : from_below;          // :1898 / :1910-1911
        const int nv = right ? from_above : cur;          // :1899-1900 / :1909
        hor = nh;
        ver = nv;
        pos_x += right ? 1 : 0;
        const int pos_y = round - (pos_x - 31);
        const bool inside = pos_x <= 32 + kLen + 31 && pos_y <= 1 + kLen;      // :1903, :1913: checked before the round is stored
        alive = alive && inside;
        const int i1 = pos_y + 30 - k;                    // 0-based index into seq1 of this lane's row y = pos_y + 31 - k
        const int i2 = pos_x - 63 + k;                    // 0-based index into seq2 of this lane's column x = pos_x - 62 + k
        const int l1 = seq1s[seq_base + (uint32_t)min(max(i1, 0), kLen - 1)];
        const int l2 = seq2s[seq_base + (uint32_t)min(max(i2, 0), kLen - 1)];
        const int c1 = (unsigned)i1 < (unsigned)kLen ? l1 : 0xF0;             // pads, source.cpp:1861-1873
        const int c2 = (unsigned)i2 < (unsigned)kLen ? l2 : 0xF1;
        const int s = (c1 == c2 && c1 < 4) ? 1 : -1;      // :1918-1920 (a pad never equals anything)
        const int vd = dia != 0 ? dia + s : 0;            // :1922
        const int vu = sat_dec(ver);                      // :1924, 0 stays 0
        const int vl = sat_dec(hor);                      // :1923
        const int m1 = vd > vu ? vd : vu;
        const int v0 = m1 > vl ? m1 : vl;                 // >= 0
        const int rm = row16_max(v0);
        const int best_a = max(__builtin_amdgcn_readlane(rm, 0), __builtin_amdgcn_readlane(rm, 16));
        const int best_b = max(__builtin_amdgcn_readlane(rm, 32), __builtin_amdgcn_readlane(rm, 48));
        const int round_best = second ? best_b : best_a;
        const int gain = alive ? round_best : 0;
        const bool improved = gain > best;                // :1933-1936
        const unsigned long long hit = __ballot(v0 == round_best);
        const unsigned mine = second ? (unsigned)(hit >> 32) : (unsigned)hit;
        best = improved ? gain : best;
        best_round = improved ? round : best_round;
        best_lane = improved ? 31 - __builtin_clz(mine) : best_lane;          // the search of :1957-1958 walks down from lane 31
        const int v = v0 < best - kXDrop ? 0 : v0;        // :1938-1941
        // predecessor code in the reference's tie-break order (diag, up, left; :1962-1971): 1 / 2 / 3, 0 for a dropped cell.
        // v != 0 && vd == v implies dia != 0 (and likewise for up), so three compare masks are enough.
        const unsigned long long m_nz = __ballot(v != 0), m_d = __ballot(vd == v), m_u = __ballot(vu == v);
        const unsigned long long bit0 = m_nz & (m_d | ~m_u), bit1 = m_nz & ~m_d;
        if (alive && writer) {
            my_codes[round] = second ? make_uint2((unsigned)(bit0 >> 32), (unsigned)(bit1 >> 32))
                                     : make_uint2((unsigned)bit0, (unsigned)bit1);
            my_top[round] = (uint16_t)pos_y;
        }
        cur = alive ? v : cur;
        rounds = alive ? round + 1 : rounds;
        alive = alive && round_best != 0;                 // :1943-1946
    }
    if (writer) summary[a] = make_int4(best - kXDrop, best_round, best_lane, rounds);
}

// ---- sweep, one LANE per alignment (large batches) -----------------------------------------------------------
//
// Same results as sg_forward_kernel, different mapping: every lane sweeps its own alignment with the 32 band cells in
// registers, so a round needs no cross-lane instruction at all and a wavefront advances 64 alignments per ~16
// instructions per cell (the band-per-half-wave kernel above spends ~117 instructions per round on two alignments, most
// of them cross-lane plumbing).  It needs >= 64 alignments per wavefront to pay, i.e. large batches; the launcher picks.
//   * dropped cells hold kNeg instead of 0: the != 0 guards of source.cpp:1922-1924 then fall out of max3, and the
//     X-drop test "v < max(best - 70, 1)" resets every dropped cell to exactly kNeg each round;
//   * the sequences ride along as two 64-bit windows of 2-bit fields (cell k <-> field k), shifted by one field per
//     move; one XOR gives the match/mismatch field of all 32 cells;
//   * the band maximum and the lane that holds it come from one max over keys (value << 5 | lane);
//   * predecessor codes are collected as three 32-bit words per round (vd == v, vd == v || vu != v, live).
constexpr int kNeg = -(1 << 24);

__global__ void __launch_bounds__(64)
sg_forward_lane_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, uint32_t n,
                       uint32_t *__restrict__ codes, uint16_t *__restrict__ top_y, int4 *__restrict__ summary)
{
    // Per-lane records are staged in LDS and leave as whole 64-byte lines: every 8 rounds (codes) / 32 rounds (band rows)
    // the wavefront writes, per instruction, the lines of 16 alignments (4 lanes x 16 B each).  Storing 8 + 2 bytes per
    // lane per round directly would issue 128 partial-line requests per round and leave the sweep waiting on the TA.
    __shared__ uint2 stage_codes[64][8];                  // [alignment of the block][round & 7]
    __shared__ uint16_t stage_top[64][32];                // [alignment of the block][round & 31]
    const int lane = threadIdx.x;
    const uint32_t block_first = blockIdx.x * 64;
    const uint32_t a0 = block_first + threadIdx.x;
    const bool real = a0 < n;
    const uint32_t a = real ? a0 : n - 1;
    const uint8_t *s1 = seq1s + (size_t)a * kLen;
    const uint8_t *s2 = seq2s + (size_t)a * kLen;
    // group-of-8-rounds `g8` (rounds 8*g8 .. 8*g8+7) of all 64 alignments -> global, 4 x 16 lines
    auto flush_codes = [&](int g8) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int al = q * 16 + (lane >> 2), part = lane & 3;
            const uint4 v = *reinterpret_cast<const uint4 *>(&stage_codes[al][2 * part]);
            if (block_first + al < n)
                *reinterpret_cast<uint4 *>(reinterpret_cast<uint2 *>(codes) + (size_t)(block_first + al) * kCodeStride + 8 * g8 + 2 * part) = v;
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto flush_top = [&](int g32) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int al = q * 16 + (lane >> 2), part = lane & 3;
            const uint4 v = *reinterpret_cast<const uint4 *>(&stage_top[al][8 * part]);
            if (block_first + al < n)
                *reinterpret_cast<uint4 *>(top_y + (size_t)(block_first + al) * kTopStride + 32 * g32 + 8 * part) = v;
        }
        __builtin_amdgcn_wave_barrier();
    };

    int cur[32], hor[32], ver[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) { cur[k] = kNeg; hor[k] = kNeg; ver[k] = kNeg; }
    cur[31] = kXDrop;
    // round 0: pos_y = 0, pos_x = 31 -> cell k sits at row 31 - k (valid for k <= 30), column k - 31 (never valid)
    unsigned long long aw = 0, av = 0, bw = 0, bv = 0;
#pragma unroll
    for (int k = 0; k <= 30; ++k) {
        aw |= (unsigned long long)(s1[30 - k] & 3u) << (2 * k);
        av |= 3ull << (2 * k);
    }
    int pos_x = 31;
    int best = kXDrop, best_round = 0, best_lane = 31, rounds = 1, last_round = 0;
    bool alive = true;
    stage_codes[lane][0] = make_uint2(0, 0);
    stage_top[lane][0] = 0;
    // sequence characters arrive 16 at a time, one 16-byte load per lane per 16 consumed characters, fetched one
    // buffer ahead (the load has 16 moves to land)
    uint4 abuf = *reinterpret_cast<const uint4 *>(s1 + 16);    // seq1[16..31]: the next step down consumes seq1[31]
    uint4 anext = *reinterpret_cast<const uint4 *>(s1 + 32);
    uint4 bbuf = *reinterpret_cast<const uint4 *>(s2);         // seq2[0..15]:  the next step right consumes seq2[0]
    uint4 bnext = *reinterpret_cast<const uint4 *>(s2 + 16);
    auto pick = [](const uint4 &buf, int idx) -> unsigned {    // character idx & 15 of the buffered 16
        const unsigned w = (idx & 8) ? ((idx & 4) ? buf.w : buf.z) : ((idx & 4) ? buf.y : buf.x);
        return (w >> (8 * (idx & 3))) & 3u;
    };

    for (int round = 1; round < kMaxRound; ++round) {
        if (!__any(alive)) break;
        const bool right = cur[0] < cur[31];              // source.cpp:1895
        pos_x += right ? 1 : 0;
        const int pos_y = round - (pos_x - 31);
        alive = alive && pos_x <= 32 + kLen + 31 && pos_y <= 1 + kLen;        // :1903, :1913
        // sequence windows follow the band
        {
            const int ia = pos_y + 30, ib = pos_x - 32;   // 0-based index of the character that has just entered
            const bool va = (unsigned)ia < (unsigned)kLen, vb = (unsigned)ib < (unsigned)kLen;
            const unsigned cand_a = pick(abuf, ia), cand_b = pick(bbuf, ib);
            const unsigned long long aw_d = (aw << 2) | cand_a, av_d = (av << 2) | (va ? 3ull : 0ull);
            const unsigned long long bw_r = (bw >> 2) | ((unsigned long long)cand_b << 62), bv_r = (bv >> 2) | (vb ? 3ull << 62 : 0ull);
            aw = right ? aw : aw_d;  av = right ? av : av_d;
            bw = right ? bw_r : bw;  bv = right ? bv_r : bv;
            // a buffer whose last character (index 15 mod 16) has just been consumed is replaced by the prefetched one
            if (!right && (ia & 15) == 15) {
                abuf = anext;
                if (ia + 17 < kLen) anext = *reinterpret_cast<const uint4 *>(s1 + ia + 17);
            }
            if (right && (ib & 15) == 15) {
                bbuf = bnext;
                if (ib + 17 < kLen) bnext = *reinterpret_cast<const uint4 *>(s2 + ib + 17);
            }
        }
        const unsigned long long x = aw ^ bw;
        const unsigned long long same = ~(x | (x >> 1)) & 0x5555555555555555ull & av & bv;   // bit 2k: cell k is a match
        const unsigned long long m2 = same << 1;                                            // field k = 2 (match) or 0
        const unsigned m2lo = (unsigned)m2, m2hi = (unsigned)(m2 >> 32);

        int kmax = kNeg;
        unsigned w_nd = 0, w_nu = 0;                      // bit k: vd != v0 / vu != v0, shifted in as the sign of the difference
        int pending = 0;                                  // new value of cell k+1, written once cell k no longer needs the old one
#pragma unroll
        for (int k = 31; k >= 0; --k) {
            const int c_lo = k > 0 ? cur[k - 1] : kNeg, c_mid = cur[k], c_hi = k < 31 ? cur[k + 1] : kNeg;
            const int dia = right ? ver[k] : hor[k];      // :1897 / :1908
            const int nh = right ? c_mid : c_lo;          // :1898 / :1910-1911
            const int nv = right ? c_hi : c_mid;          // :1899-1900 / :1909
            hor[k] = nh;
            ver[k] = nv;
            const int f = (int)(((k < 16 ? m2lo : m2hi) >> (2 * (k & 15))) & 3u);
            const int vd = dia + f - 1;                   // +1 / -1, :1918-1922
            const int vu = nv - 1, vl = nh - 1;           // :1923-1924
            const int m1 = vd > vu ? vd : vu;
            const int v0 = m1 > vl ? m1 : vl;
            const int key = (int)(((unsigned)v0 << 5) | (unsigned)k);
            kmax = kmax > key ? kmax : key;
            w_nd = __builtin_amdgcn_alignbit(w_nd, (unsigned)(vd - v0), 31);   // (w << 1) | sign(vd - v0); v0 >= vd always
            w_nu = __builtin_amdgcn_alignbit(w_nu, (unsigned)(vu - v0), 31);
            if (k < 31) cur[k + 1] = pending;
            pending = v0;
        }
        cur[0] = pending;
        const int band_best = kmax >> 5;                  // arithmetic shift: the value part of the winning key
        const int round_best = band_best > 0 ? band_best : 0;
        const bool improved = alive && round_best > best; // :1933-1936
        best = improved ? round_best : best;
        best_round = improved ? round : best_round;
        best_lane = improved ? (kmax & 31) : best_lane;   // highest lane among equals: where the search of :1957 stops
        const int thr = best - kXDrop > 1 ? best - kXDrop : 1;                // :1938-1941, and "0 means dropped"
        unsigned w_drop = 0;
#pragma unroll
        for (int k = 31; k >= 0; --k) {
            const int d = cur[k] - thr;
            w_drop = __builtin_amdgcn_alignbit(w_drop, (unsigned)d, 31);
            cur[k] = d < 0 ? kNeg : cur[k];
        }
        const unsigned nz = ~w_drop, wd = ~w_nd, wx = ~w_nd | w_nu;   // live; came by the diagonal; diagonal or not up
        // codes 1 / 2 / 3 = diag / up / left (:1962-1971).  Rounds past a lane's end are staged and flushed too: they
        // land beyond `rounds` of that alignment, which the traceback never reads.
        stage_codes[lane][round & 7] = make_uint2(nz & wx, nz & ~wd);
        stage_top[lane][round & 31] = (uint16_t)pos_y;
        if ((round & 7) == 7) flush_codes(round >> 3);
        if ((round & 31) == 31) flush_top(round >> 5);
        rounds = alive ? round + 1 : rounds;
        alive = alive && round_best != 0;                 // :1943-1946
        last_round = round;
    }
    // the partial groups of the last executed round (uniform across the wavefront)
    if ((last_round & 7) != 7) flush_codes(last_round >> 3);
    if ((last_round & 31) != 31) flush_top(last_round >> 5);
    if (real) summary[a] = make_int4(best - kXDrop, best_round, best_lane, rounds);
}

// Traceback: one wavefront per alignment.  The walk itself is scalar (y, x and the round live in SGPRs); the lanes hold
// 64 consecutive rounds of (code words, band row) each, fetched with coalesced loads one block ahead of the walker, and
// the walker picks its round with v_readlane.  Two walks: the first counts the steps, the second writes the positions
// at their final (ascending) index, 64 at a time.
__global__ void __launch_bounds__(64)
sg_traceback_kernel(uint32_t n, const uint32_t *__restrict__ codes, const uint16_t *__restrict__ top_y,
                    const int4 *__restrict__ summary, int32_t *__restrict__ scores, int32_t *__restrict__ tracebacks,
                    uint32_t cap, uint32_t *__restrict__ lengths)
{
    const uint32_t a = blockIdx.x;
    const int lane = threadIdx.x;
    const uint2 *my_codes = reinterpret_cast<const uint2 *>(codes) + (size_t)a * kCodeStride;
    const uint16_t *my_top = top_y + (size_t)a * kTopStride;
    const int4 sum = summary[a];
    const int y0 = (int)my_top[sum.y] + 31 - sum.z;
    const int x0 = sum.y - y0;
    int2 *out = reinterpret_cast<int2 *>(tracebacks) + (size_t)a * cap;

    // With room for the longest possible path the positions are written once, in walking (descending) order, and the
    // wavefront reverses them in place afterwards; a smaller `cap` needs the count first (two walks).
    const bool one_walk = cap >= (uint32_t)kMaxRound;
    uint32_t total = 0;
    for (int pass = one_walk ? 1 : 0; pass < 2; ++pass) {
        int y = y0, x = x0;
        int base = ((y + x) >> 6) << 6;                   // lanes hold rounds base .. base+63 (cur) and base-64 .. base-1 (nxt)
        auto fetch = [&](int b, uint2 &cw, int &tw) {
            const int r = b + lane;
            const bool ok = b >= 0 && r < kMaxRound;
            cw = ok ? my_codes[r] : make_uint2(0, 0);
            tw = ok ? (int)my_top[r] : 0;
        };
        uint2 cw, nw;
        int tw, nt;
        fetch(base, cw, tw);
        fetch(base - 64, nw, nt);
        uint32_t count = 0;                               // positions emitted so far (descending order)
        int by = 0, bx = 0;                               // this lane's slot of the 64-position output buffer
        bool more = true;
        while (more) {
            // record the current position in slot count % 64
            const int slot = (int)(count & 63u);
            if (lane == slot) { by = y; bx = x; }
            ++count;
            more = (y | x) != 0;
            if (more) {
                const int r = y + x;
                if (r < base) {                           // walked off the block: take the prefetched one, prefetch the next
                    base -= 64;
                    cw = nw; tw = nt;
                    fetch(base - 64, nw, nt);
                }
                const int idx = __builtin_amdgcn_readfirstlane(r - base);
                const unsigned lo = __builtin_amdgcn_readlane(cw.x, idx), hi = __builtin_amdgcn_readlane(cw.y, idx);
                const int top = __builtin_amdgcn_readlane(tw, idx);
                const int bl = 31 - (y - top);
                const int code = (int)((lo >> bl) & 1u) | (int)(((hi >> bl) & 1u) << 1);
                if (code == 1) { --y; --x; }
                else if (code == 2) { --y; }
                else if (code == 3) { --x; }
                else more = false;                        // cannot happen for a cell on a live path
            }
            if (pass == 1 && (slot == 63 || !more)) {     // flush 64 buffered positions with one coalesced store
                const uint32_t c = count - 1 - (uint32_t)slot + (uint32_t)lane;   // element number of this lane's slot
                if (lane <= slot) {
                    const uint32_t idx_out = one_walk ? c : total - 1 - c;
                    if (idx_out < cap) out[idx_out] = make_int2(by, bx);
                }
            }
        }
        total = count;
    }
    if (one_walk) {                                       // reverse out[0 .. total) in place
        __syncthreads();                                  // one wavefront per block: orders the stores above before the loads below
        for (uint32_t i = (uint32_t)lane; i < total / 2; i += 64) {
            const int2 lo_v = out[i], hi_v = out[total - 1 - i];
            out[i] = hi_v;
            out[total - 1 - i] = lo_v;
        }
    }
    if (lane == 0) { scores[a] = sum.x; lengths[a] = total; }
}

// Traceback, one LANE per alignment (large batches): the wave-per-alignment walker above is bound by the scalar unit
// (one walk per wavefront, ~40 scalar instructions per step); here 64 walks advance per vector instruction.  Each lane
// keeps the 64-byte line of codes (8 rounds) and of band rows (32 rounds) it is walking through in LDS.
// The 64 walks of a wavefront move in LOCKSTEP BY WINDOW of 8 rounds: all lanes consume window w (each at its own pace,
// 4..8 steps), then the whole wavefront swaps in the line of window w-1, which was requested before window w was
// walked.  (Refilling per lane, whenever a walk left its line, made almost every step wait for some lane's load: 64
// walks at random phases, one dependent HBM access per ~5 steps each.)  Two walks: count, then write each position at
// its final (ascending) index.
__global__ void __launch_bounds__(64)
sg_traceback_lane_kernel(uint32_t n, const uint32_t *__restrict__ codes, const uint16_t *__restrict__ top_y,
                         const int4 *__restrict__ summary, int32_t *__restrict__ scores, int32_t *__restrict__ tracebacks,
                         uint32_t cap, uint32_t *__restrict__ lengths)
{
    __shared__ uint4 line_codes[64][4 + 1];               // [lane][16-byte quarter of the line], padded
    __shared__ uint4 line_top[64][4 + 1];
    const int lane = threadIdx.x;
    const uint32_t a0 = blockIdx.x * 64 + threadIdx.x;
    const bool real = a0 < n;
    const uint32_t a = real ? a0 : n - 1;                 // tail lanes shadow the last alignment and store nothing
    const uint4 *my_codes = reinterpret_cast<const uint4 *>(reinterpret_cast<const uint2 *>(codes) + (size_t)a * kCodeStride);
    const uint4 *my_top = reinterpret_cast<const uint4 *>(top_y + (size_t)a * kTopStride);
    const int4 sum = summary[a];
    const int y0 = (int)top_y[(size_t)a * kTopStride + sum.y] + 31 - sum.z;
    const int x0 = sum.y - y0;                            // y0 + x0 = the round of the best cell
    int2 *out = reinterpret_cast<int2 *>(tracebacks) + (size_t)a * cap;
    // first window of the wavefront = the highest one any of its walks starts in
    int wmax = sum.y >> 3;
    wmax = row16_max(wmax);
    wmax = max(max(__builtin_amdgcn_readlane(wmax, 0), __builtin_amdgcn_readlane(wmax, 16)),
               max(__builtin_amdgcn_readlane(wmax, 32), __builtin_amdgcn_readlane(wmax, 48)));

    uint32_t total = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int y = y0, x = x0;
        bool walking = (y | x) != 0;
        uint32_t count = 1;
        uint32_t idx = total - 1;                         // pass 1: index of the current position in the ascending list
        if (pass == 1 && real && idx < cap) out[idx] = make_int2(y, x);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            line_codes[lane][q] = my_codes[4 * wmax + q];
            line_top[lane][q] = my_top[4 * (wmax >> 2) + q];
        }
        for (int w = wmax; w >= 0; --w) {
            // request the lines of window w - 1 now; they are needed only after window w has been walked
            // (window 0 re-requests itself: no branch around the loads, the values stay in registers)
            const int wp = w > 0 ? w - 1 : 0;
            const bool top_changes = (w & 3) == 0;        // window w - 1 lies in the previous 32-round line of band rows
            const uint4 nc0 = my_codes[4 * wp], nc1 = my_codes[4 * wp + 1], nc2 = my_codes[4 * wp + 2], nc3 = my_codes[4 * wp + 3];
            uint4 nt0 = make_uint4(0, 0, 0, 0), nt1 = nt0, nt2 = nt0, nt3 = nt0;
            if (top_changes) {
                const uint4 *tp = my_top + 4 * (wp >> 2);
                nt0 = tp[0]; nt1 = tp[1]; nt2 = tp[2]; nt3 = tp[3];
            }
            while (walking && ((y + x) >> 3) == w) {
                const int r = y + x;
                const uint2 cw = reinterpret_cast<const uint2 *>(&line_codes[lane][0])[r & 7];
                const int top = (int)reinterpret_cast<const uint16_t *>(&line_top[lane][0])[r & 31];
                const int bl = 31 - (y - top);
                const int code = (int)((cw.x >> bl) & 1u) | (int)(((cw.y >> bl) & 1u) << 1);
                y -= (code == 1 || code == 2) ? 1 : 0;    // 1 diag, 2 up: one row back
                x -= (code == 1 || code == 3) ? 1 : 0;    // 1 diag, 3 left: one column back
                walking = code != 0 && (y | x) != 0;      // code 0 cannot happen for a cell on a live path
                if (code != 0) {
                    ++count;
                    if (pass == 1) {
                        --idx;
                        if (real && idx < cap) out[idx] = make_int2(y, x);
                    }
                }
            }
            line_codes[lane][0] = nc0; line_codes[lane][1] = nc1; line_codes[lane][2] = nc2; line_codes[lane][3] = nc3;
            if (top_changes) {
                line_top[lane][0] = nt0; line_top[lane][1] = nt1; line_top[lane][2] = nt2; line_top[lane][3] = nt3;
            }
        }
        total = count;
    }
    if (real) {
        scores[a] = sum.x;
        lengths[a] = total;
    }
}

}  // namespace

namespace {
inline size_t round16(size_t v) { return (v + 15) & ~size_t(15); }
inline size_t codes_bytes(size_t n) { return round16(n * (size_t)kCodeStride * sizeof(uint2)); }
inline size_t top_bytes(size_t n) { return round16(n * (size_t)kTopStride * sizeof(uint16_t)); }
}  // namespace

size_t semiglobal_workspace_bytes(size_t n)
{
    return codes_bytes(n) + top_bytes(n) + round16(n * sizeof(int4));
}

hipError_t launch_semiglobal(const uint8_t *d_seq1s, const uint8_t *d_seq2s, size_t n, void *d_workspace,
                             int32_t *d_scores, int32_t *d_tracebacks, size_t cap, uint32_t *d_lengths, hipStream_t stream,
                             hipEvent_t between)
{
    if (n == 0) return hipSuccess;
    char *ws = static_cast<char *>(d_workspace);
    uint32_t *codes = reinterpret_cast<uint32_t *>(ws);
    uint16_t *top = reinterpret_cast<uint16_t *>(ws + codes_bytes(n));
    int4 *summary = reinterpret_cast<int4 *>(ws + codes_bytes(n) + top_bytes(n));
    // (Cutting the batch into sub-batches so that traceback k overlaps sweep k+1 was tried and is slower: below ~16k
    // alignments the sweep is latency bound, and four short sweeps in sequence cost four times one.)
    // two mappings of the sweep, same results: a band per half-wavefront (low latency, fills the chip from a few
    // thousand alignments) or an alignment per lane (far fewer instructions per alignment, needs a large batch)
    const char *force = getenv("SWMI_SG_SWEEP");
    const bool lane_sweep = force ? atoi(force) == 1 : n >= kLaneSweepMinBatch;
    if (lane_sweep) {
        hipLaunchKernelGGL(sg_forward_lane_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_seq1s, d_seq2s,
                           (uint32_t)n, codes, top, summary);
    } else {
        const unsigned waves = (unsigned)((n + 1) / 2);
        hipLaunchKernelGGL(sg_forward_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, d_seq1s, d_seq2s, (uint32_t)n, codes,
                           top, summary);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && between) e = hipEventRecord(between, stream);      // phase timing (swmi_semiglobal_time_device)
    if (e != hipSuccess) return e;
    const char *force_tb = getenv("SWMI_SG_TRACEBACK");
    const bool lane_tb = force_tb ? atoi(force_tb) == 1 : n >= kLaneTracebackMinBatch;
    if (lane_tb)
        hipLaunchKernelGGL(sg_traceback_lane_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, (uint32_t)n, codes, top,
                           summary, d_scores, d_tracebacks, (uint32_t)cap, d_lengths);
    else
        hipLaunchKernelGGL(sg_traceback_kernel, dim3((unsigned)n), dim3(64), 0, stream, (uint32_t)n, codes, top, summary, d_scores,
                           d_tracebacks, (uint32_t)cap, d_lengths);
    return hipGetLastError();
}

}  // namespace swmi
