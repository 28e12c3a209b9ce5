// sg_kernels.hip -- gfx950 kernels for the reference's semi-global adaptive-band X-drop aligner
//     SemiGlobal_AdaptiveBanded_XDrop_111_32_70   (source.cpp:1836-1976; SIMD variants :1978-2725)
// SURVEY.md section 8f row N4.  Same results as the reference: (score, traceback from (0,0) to the best cell).
//
// Not a translation of the AVX2 variants (one 32-byte vector per anti-diagonal, u8 cells re-based every round,
// source.cpp:2099-2109).  Here:
//   * forward sweep: the 32 cells of the band (BANDWIDTH, source.cpp:1848) live in 4 / 2 lanes with 8 / 16 cells each
//     (sg_forward_split_kernel) or in one lane (sg_forward_lane_kernel, the largest batches), two 16-bit cells per register.
//     The band's direction (right / down) is decided per alignment every round from its two end cells (:1895).  Instead
//     of the reference's 4 MB table of cell values per alignment (source.cpp:1876) the sweep stores, per round, only what
//     the traceback needs: a 2-bit predecessor TAG per cell (3 diagonal / 2 up / 1 left: the candidate that won the cell's
//     max, in the reference's own tie-break order :1962-1971) = 8 bytes, and ONE bit for the band's move (right / down);
//     the row of the band's top cell is rebuilt from the move bits (round - right moves so far = a popcount).
//     The X-drop rule (:1938-1941) is not asked in windows of 8 rounds in which no cell can reach the threshold ("CALM
//     WINDOWS" in the kernels: a margin argument, the same results) -- two loops per kernel, exact and calm.
//   * traceback: follows the codes from the best cell back to (0,0) and returns the positions in ascending order, as the
//     reference does (:1951-1975): one lane per walk that records its moves (sg_walk_lane_kernel) + a prefix-sum kernel
//     that expands them into positions (sg_expand_kernel).
#include "swmi_internal.h"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace swmi {
namespace {

constexpr int kLen = 16384;                 // std::array<uint8_t,16384>, source.cpp:1837-1838
constexpr int kXDrop = 70;                  // X_THRESHOLD, source.cpp:1848
constexpr int kMaxRound = 2 * (kLen + 1) - 1;   // MAX_ROUND, source.cpp:1875
// Predecessor records: 8 bytes per alignment and round (band cell k's 2-bit tag: below), WINDOW-MAJOR and in TWO HALVES:
//     centre half = cells 8 .. 23 (cell 8 + k at bits 2k of a 32-bit word), outer half = cells 0 .. 7 (low 16 bits) and
//     24 .. 31 (high 16 bits); 16 rounds of one alignment = one 64-byte piece per half, and the pieces of one window lie side
//     by side for all alignments of the batch:
//         centre word of (alignment a, round r) = uint32 index ((r / 16) * n + a) * 16 + (r % 16) of the centre array,
//         the outer word the same index of the outer array behind it.
// A walk that stays in cells 8 .. 23 -- where the band keeps the best path -- never needs the outer half: the walk kernel
// fetches the centre pieces only (half the bytes of the records) and the outer piece of a window when a walk of its
// wavefront leaves the centre in it.  (Round 4; before, a round's record was one 8-byte word and a window one 128-byte line.)
// Sweeps and walks move through the rounds in lockstep -- every alignment of a wavefront is in the same window at the same
// time -- so a sweep wavefront's flush (its 16 / 32 / 64 alignments) and a walk wavefront's fetch (its 64 walks) are
// contiguous blocks of 1 .. 4 KB per half, written and read with fully coalesced 16-byte accesses.  (Round 2 kept one array per alignment: a walk
// wavefront's fetch was 64 separate lines 262 KB apart, each requested in 16-byte pieces by one lane: 3.5 TB/s.)
constexpr int kCodeWindow = 16;                           // rounds per piece
constexpr int kCodeWindows = (kMaxRound + kCodeWindow - 1) / kCodeWindow;
constexpr int kHalfQuads = kCodeWindow / 4;               // uint4 per piece (64 bytes)
constexpr int kStagePitch = kCodeWindow + 4;              // a sweep's LDS staging row per alignment and half, in words: 80 bytes
__device__ __forceinline__ size_t half_quads(uint32_t n) { return (size_t)kCodeWindows * n * kHalfQuads; }   // uint4 per half array
// move bits: bit (r & 31) of word r >> 5 = 1 when the band stepped right in round r (source.cpp:1895); stored
// word-major, dirs[word * n + alignment], so that the writers and the readers of neighbouring alignments share lines
constexpr int kDirWords = kMaxRound / 32 + 1;
// Predecessor records.  Every sweep stores, per round and band cell, the 2-bit TAG of the candidate that won the cell's
// three-way max: 3 diagonal, 2 up, 1 left (0: round 0 / nothing) -- the reference's tie-break order (source.cpp:1962-1971)
// falls out of comparing equal values by tag -- band cell k at bits 2k..2k+1 of the round's 64 bits, which leave as two words (above).
// The walk hands the tags on as its moves; the expand kernel reads a row step off bit 1 and a column step off bit 0.
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

__device__ __forceinline__ int keep_opaque(int v)         // stops hipcc from turning `x & mask` into a v_cndmask
{
    asm volatile("" : "+v"(v));
    return v;
}

// A cell value of the sweeps travels as  stored * kScale + band_cell * 4 + tag  in 16 bits, stored = true + round - base (a gap
// or a mismatch step then adds 0 / 1 and a match 3: nothing negative is ever added to an unsigned half), re-based every 16
// rounds so that the X-drop threshold is kPkFloor again; kPkBase0 = the base before round 1 (threshold 1 + 1 - base = kPkFloor).
constexpr int kScale = 128;
constexpr int kPkFloor = 8;
constexpr int kPkBase0 = 2 - kPkFloor;

// ---- character streams for the split sweep -------------------------------------------------------------------------
//
// The band consumes seq1 top to bottom and seq2 left to right, one character per move, and only ever the next one.  A
// pre-pass rewrites both sequences of every alignment as streams of 4-bit fields, 16 per 64-bit word: the base (0..3), and
// past the end of the sequence the pad the reference appends (source.cpp:1861-1873) -- 4 for seq1, 5 for seq2, so that a
// pad never equals a base or the other pad.  The sweep then needs no index arithmetic, range checks or byte extraction:
// next character = low field of a 64-bit shift register, topped up every 16 rounds from prefetched words.
// Layout: the streams of the A alignments one sweep wavefront owns are interleaved word by word,
//     streams[(block * kStreamWords + word) * 2A + alignment_in_block * 2 + {0: seq1, 1: seq2}]
// because the lanes of a wavefront ask for (nearly) the same word index at the same time: their 8-byte loads then fall
// into the same few 64-byte lines (with one stream after the other per alignment, every 8-byte load pulled in a line of
// its own and each line was fetched up to eight times: 10 GB of fetches for 1.1 GB of streams at 65536 alignments).
constexpr unsigned kPadSeq1 = 4u, kPadSeq2 = 5u;          // three bits: the match test ORs three bits of the XOR
constexpr int kStreamWords = (kLen + 128) / 16;          // 16 fields per word; 128 fields of pad cover every read-ahead

// 4 bytes -> 4 fields (16 bits): a byte that is no base scores as a mismatch against anything (source.cpp:1918-1920), which
// is what a pad field does
__device__ __forceinline__ unsigned sg_squeeze4(unsigned v, unsigned pad)
{
    const unsigned high = v & 0xFCFCFCFCu;                                          // nonzero in a byte: no base
    const unsigned flag = (((high & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | high) & 0x80808080u;   // bit 7 of every such byte
    const unsigned mask = flag | (flag - (flag >> 7));                              // 0xFF in every such byte
    unsigned c = (v & 0x03030303u & ~mask) | ((pad * 0x01010101u) & mask);
    c = (c | (c >> 4)) & 0x00FF00FFu;
    return (c | (c >> 8)) & 0xFFFFu;
}

// One wavefront per TILE: 8 consecutive words (128 characters) of all 2A streams of one sweep block.  Eight lanes read one
// 128-byte line of a sequence, the tile crosses LDS, and it leaves as one contiguous block of 8 * 2A words in 16-byte pieces
// -- every line of the input is fetched once and every store instruction writes a contiguous kilobyte.  (One thread per output
// word, the earlier form: the 16-byte reads of neighbouring threads lay 16 KB apart and every input line was requested by eight
// different wavefronts; 1.3 ms at 65536 alignments with 128 streams interleaved.)
constexpr int kPackWords = 8;                             // words per tile
static_assert(kStreamWords % kPackWords == 0 && kLen / 16 % kPackWords == 0, "whole tiles; the pad words are a tile of their own");
__global__ void __launch_bounds__(256)
sg_pack_streams_kernel(const uint8_t *__restrict__ seq1s, const uint8_t *__restrict__ seq2s, uint32_t n,
                       unsigned long long *__restrict__ streams, uint32_t per_block /* A: alignments per sweep wavefront, 16 / 32 / 64 */)
{
    __shared__ unsigned long long tiles[4][kPackWords * (128 + 8)];          // [wavefront][word][slot], rows padded by 8 words
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t slots = 2 * per_block, pitch = slots + 8;                 // (the pitch keeps the 8-byte writes of a wavefront on distinct banks)
    constexpr uint32_t kTiles = kStreamWords / kPackWords;
    const size_t n_blocks = ((size_t)n + per_block - 1) / per_block;
    const size_t tile_id = (size_t)blockIdx.x * 4 + wv;
    if (tile_id >= n_blocks * kTiles) return;                                 // wave-uniform
    const uint32_t blk = (uint32_t)(tile_id / kTiles), tw = (uint32_t)(tile_id % kTiles);      // words 8 tw .. 8 tw + 7 of block blk
    unsigned long long *out = streams + ((size_t)blk * kStreamWords + (size_t)kPackWords * tw) * slots;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    if (tw >= (uint32_t)(kLen / 16 / kPackWords)) {                           // past the sequences: pads (slot parity = which sequence)
        const u64x2 pads = {kPadSeq1 * 0x1111111111111111ull, kPadSeq2 * 0x1111111111111111ull};
        for (uint32_t p = lane; p < kPackWords * slots / 2; p += 64) *reinterpret_cast<u64x2 *>(out + 2 * p) = pads;
        return;
    }
    unsigned long long *tile = tiles[wv];
    const uint32_t word = lane & 7, group = lane >> 3;
    for (uint32_t s0 = 0; s0 < slots; s0 += 8) {
        const uint32_t slot = s0 + group;
        const uint32_t a_full = blk * per_block + (slot >> 1);
        const uint32_t a = a_full < n ? a_full : n - 1;                       // a ragged last block shadows the last alignment
        const bool second = slot & 1u;                                        // 0: seq1, 1: seq2
        const uint4 b = *reinterpret_cast<const uint4 *>((second ? seq2s : seq1s) + (size_t)a * kLen + 16 * (kPackWords * tw + word));
        const unsigned pad = second ? kPadSeq2 : kPadSeq1;
        tile[word * pitch + slot] = (unsigned long long)(sg_squeeze4(b.x, pad) | (sg_squeeze4(b.y, pad) << 16)) |
                                    ((unsigned long long)(sg_squeeze4(b.z, pad) | (sg_squeeze4(b.w, pad) << 16)) << 32);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (uint32_t p = lane; p < kPackWords * slots / 2; p += 64) {            // 16-byte piece p = words 2p, 2p + 1 of the output block
        const uint32_t w = 2 * p / slots, sl = 2 * p % slots;
        const u64x2 v = {tile[w * pitch + sl], tile[w * pitch + sl + 1]};
        *reinterpret_cast<u64x2 *>(out + 2 * p) = v;
    }
}

// One sequence's character stream of one alignment (the format sg_pack_streams_kernel writes), read 16 characters ahead:
//   sreg   the next 16 characters (the next one at bit used4: a consume only advances used4), pend: the characters after
//   those (p_fill of them, low-aligned, zeros above), ahead: the word after pend (a load issued at the previous top-up),
//   w_next: index of the word after `ahead`.
// top_up() runs every 16 rounds -- a stream gives at most 16 characters in 16 rounds -- at the same place for every lane, so
// the round itself holds no load and no wait.  (With the refill inside the round hipcc kept the prefetched word in a register
// pair of its own and copied it every round, which put an s_waitcnt vmcnt(0) -- on the load AND on the record stores -- into
// every round.)
struct SgStream {
    const unsigned long long *words;                      // word w of the stream at words[w * stride]
    unsigned long long sreg, pend, ahead;
    int p_fill, w_next, used4;
    __device__ __forceinline__ void start(const unsigned long long *w, size_t stride, int first_char)
    {
        words = w;
        const int c0 = first_char & 15, w0i = first_char >> 4;
        const unsigned long long w0 = words[w0i * stride], w1 = words[(w0i + 1) * stride];
        sreg = c0 ? (w0 >> (4 * c0)) | (w1 << (64 - 4 * c0)) : w0;
        pend = w1 >> (4 * c0);
        p_fill = 16 - c0;
        ahead = words[(w0i + 2) * stride];
        w_next = w0i + 3;
        used4 = 0;
    }
    __device__ __forceinline__ unsigned next() const { return (unsigned)(sreg >> used4) & 15u; }
    __device__ __forceinline__ void top_up(size_t stride)
    {
        const int k = used4 >> 2;
        sreg = k < 16 ? sreg >> used4 : 0ull;
        used4 = 0;
        const int from_pend = k < p_fill ? k : p_fill, rem = k - from_pend;
        const unsigned long long add_p = k ? pend << (64 - 4 * k) : 0ull;           // pend's characters behind the 16 - k left
        const unsigned long long add_a = rem ? ahead << (64 - 4 * rem) : 0ull;      // ... then `rem` characters of the word after it
        sreg |= add_p | add_a;
        // pend is used up: `ahead` becomes pend and the next word is requested.  Written WITHOUT a branch around the load: every
        // lane loads (one that keeps its `ahead` asks for the same word again, a hit) straight into `ahead`, which nothing reads
        // before the next top-up.  A load under a per-lane condition goes through a temporary + copy, and the copy waits for
        // the load on the spot: a whole memory latency per 16 rounds with one wavefront on the SIMD.
        const bool take = rem > 0 || from_pend == p_fill;
        const unsigned long long pend_take = rem < 16 ? ahead >> (4 * rem) : 0ull;
        const unsigned long long pend_keep = pend >> (4 * (from_pend & 15));        // (from_pend = 16 only when `take`)
        pend = take ? pend_take : pend_keep;
        p_fill = take ? 16 - rem : p_fill - from_pend;
        w_next += take ? 1 : 0;
        const int w = w_next - 1 < kStreamWords ? w_next - 1 : kStreamWords - 1;     // a band that has left the matrix keeps stepping
        ahead = words[(size_t)w * stride];
    }
};

// ---- packed 16-bit helpers of the split and lane sweeps ----------------------------------------------------------
// Two band cells per register.  Every half is an integer in [0, 0x7C00): as bit patterns those are the non-negative finite
// half-precision numbers in increasing order, so v_pk_maximum3_f16 is a packed 3-way INTEGER max on them (the scorer's
// premise, checked exhaustively by swmi_selftest_pk_max3; the denormal patterns below 1024 need MODE.FP_DENORM kept).
// Written with the compiler's own vector builtins, not inline assembly: hipcc forms v_pk_maximum3_f16 from the nested
// maximum, knows the VOP3P result hazard and schedules around it.
typedef _Float16 sg_half2 __attribute__((ext_vector_type(2)));
typedef short sg_short2 __attribute__((ext_vector_type(2)));
typedef unsigned short sg_ushort2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sg_pk_max3(unsigned a, unsigned b, unsigned c)
{
    const sg_half2 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(sg_half2, a), __builtin_bit_cast(sg_half2, b)),
                                                     __builtin_bit_cast(sg_half2, c));
    return __builtin_bit_cast(unsigned, m);
}
__device__ __forceinline__ unsigned sg_pk_min3(unsigned a, unsigned b, unsigned c)
{
    const sg_half2 m = __builtin_elementwise_minimum(__builtin_elementwise_minimum(__builtin_bit_cast(sg_half2, a), __builtin_bit_cast(sg_half2, b)),
                                                     __builtin_bit_cast(sg_half2, c));
    return __builtin_bit_cast(unsigned, m);
}
__device__ __forceinline__ unsigned sg_pk_below(unsigned v, unsigned limit)      // 0xFFFF in every half with v < limit (both < 0x8000)
{
    const sg_short2 d = (__builtin_bit_cast(sg_short2, v) - __builtin_bit_cast(sg_short2, limit)) >> 15;
    return __builtin_bit_cast(unsigned, d);
}
__device__ __forceinline__ unsigned sg_pk_sub_sat(unsigned v, unsigned d)        // max(v - d, 0) in every half: v_pk_sub_u16 ... clamp
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(sg_ushort2, v), __builtin_bit_cast(sg_ushort2, d)));
}
__device__ __forceinline__ void sg_keep_f16_denormals()                          // MODE.FP_DENORM[7:6] = 3 (hipcc's default, pinned)
{
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 3");
}

// ---- sweep, G lanes per alignment (C = 32 / G band cells per lane) -------------------------------------------------
//
// Same results as sg_forward_kernel, different mapping: each lane keeps C cells of the band in registers, so most of a
// round is plain per-lane arithmetic.  What crosses lanes per round is small: the two cells next to the lane's slice, one
// sequence character in each direction, the band's two end cells (direction) and the band maximum -- all DPP moves inside
// a row of 16 lanes.
//
// The cell (round 3 form: TWO band cells per register; round 2 kept one int32 per cell and ran 13.5 instructions per cell
// where this runs 5.6 -- profiles/r03_sg_kernel_matrix.txt has both).  A cell value travels as ONE tagged 16-bit integer
//     V = stored * 128 + band_cell * 4 + tag                     (kScale = 128: 5 bits of cell index, 2 bits of tag)
//     stored = true value + round - base                          (below)
// register k of a lane holds cells k (low half) and k + C/2 (high half), and the three candidates of a cell carry the tags
// 3 (diagonal), 2 (up), 1 (left).  Then
//   * V0 = v_pk_maximum3_f16(VD, VU, VL) yields, for both cells, the value AND in its low two bits which predecessor won --
//     with the reference's tie-break order (diagonal, then up, then left; source.cpp:1962-1971) because equal values compare
//     by tag.  Every half stays below 0x7C00, where the half-precision order of the bit patterns is the integer order (the
//     scorer's premise, swmi_selftest_pk_max3);
//   * all three candidates of a cell carry the same cell index, so the index never influences the cell's own max, while
//     the band maximum max(V0 over cells and lanes) is decided by value, then by the HIGHEST cell index among equals --
//     where the reference's search stops (source.cpp:1957-1958); the tag never matters there (indices differ);
//   * nothing negative is ever added to an unsigned half: a stored value is  true + round - base,  so a gap step (true - 1,
//     one round later) adds 0, a mismatch (true - 1, two rounds later) adds 1 and a match adds 3.  The X-drop threshold
//     climbs by one or two per round in stored terms; every 16 rounds (where the records are flushed anyway) base moves so
//     that it is kPkFloor again: live values stay below 8 + 32 + 71 + 3 < 248 = 0x7C00 / 128;
//   * a dropped cell holds 0 (the reference stores 0 and guards with != 0, source.cpp:1922-1924): a candidate derived only
//     from dropped cells is <= 3 < kPkFloor - 2 <= every live one, is dropped again by the X-drop test and counts as "<= 0"
//     for the band maximum.  The test is  cur = V0 & ~(V0 < threshold) & clean_mask  per register: v_pk_sub_i16,
//     v_pk_ashrrev_i16, v_bitop3;
//   * with cells k and k + C/2 in one register the shifted views S[j] = right ? P[j] : P[j-1] (left = S[c], up = S[c+1]; last
//     round's view gives the diagonal) are whole registers again: S of register k = right ? register k : register k - 1, one
//     v_bitop3_b32 select per REGISTER with a per-lane all-ones / all-zeros mask (v_cndmask is half rate); only the two
//     registers at the slice's ends are assembled from halves (v_alignbit);
//   * index and tag of the two gap candidates of two registers are added by ONE v_lshl_add_u64 (no half ever carries);
//   * sequence characters are 4-bit fields (0..3, pads 4 / 5: a pad never matches) in one window per sequence, cell c <->
//     field c, shifted by one field per move (a variable 64-bit shift by 0 or 4); XOR + OR of three bits gives the match
//     bits of all cells, split into bytes of even and odd cells, and one v_perm_b32 per register puts its two cells' bytes
//     where they add 256 = 2 * kScale;
//   * the winners' tags: one v_perm_b32 gathers the low bytes of two registers, one mask, one shift-or per pair.
// Measured (65536 alignments, G = 2): 167 VALU instructions per wavefront-round (round 2: 273.5), 31.2 -> 21.7 ms.

template <int G, int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W)))
sg_forward_split_kernel(const unsigned long long *__restrict__ streams, uint32_t n,
                        uint32_t *__restrict__ codes, uint32_t *__restrict__ dirs, int4 *__restrict__ summary, int exact_only,
                        uint32_t *__restrict__ window_stats)
{
    constexpr int C = 32 / G;                             // band cells per lane
    constexpr int A = 64 / G;                             // alignments per wavefront
    static_assert(G == 2 || G == 4, "the cross-lane moves below are written for quads");
    using win_t = typename std::conditional<C == 8, uint32_t, unsigned long long>::type;   // C 4-bit fields
    // sixteen rounds of code records per alignment are staged here and leave as ONE 128-byte line per alignment: a
    // 64-byte piece of a 128-byte L2 line costs a read-for-ownership of the whole line on top of the write
    __shared__ __attribute__((aligned(16))) uint32_t stage_codes[2][A][kStagePitch];   // [centre / outer word][alignment of the block][round & 15] (+ 4: see the flush)
    const int lane = threadIdx.x;
    const int g = lane & (G - 1);                         // slice of the band: cells g*C .. g*C + C-1
    const int al = lane / G;                              // alignment of the block
    const bool is_first = g == 0, is_last = g == G - 1;
    const uint32_t block_first = blockIdx.x * A;
    const uint32_t a0 = block_first + al;
    const bool real = a0 < n;
    const uint32_t a = real ? a0 : n - 1;
    // word w of this alignment's seq1 / seq2 stream: stream_x[w * kStreamStride] (interleaved layout, see sg_pack_streams_kernel)
    constexpr size_t kStreamStride = 2 * A;
    const unsigned long long *stream_a = streams + (size_t)blockIdx.x * kStreamWords * kStreamStride + 2 * al;
    const unsigned long long *stream_b = stream_a + 1;
    // this slice's bytes of the round's 8-byte code record: cell k of the band at bits 2k, 2k+1
    // The lane's cells in 16-bit pieces of 8: piece j of the band is the low half of the outer word, the low and the high
    // half of the centre word (cells 8 .. 23), the high half of the outer word for j = 0 .. 3
    const int piece0 = g * C / 8;                         // first piece of this lane (C / 8 pieces: 1 or 2)
    auto piece_home = [&](int j) -> uint8_t * {
        return reinterpret_cast<uint8_t *>(&stage_codes[j == 1 || j == 2 ? 0 : 1][al][0]) + (j >= 2 ? 2 : 0);
    };
    uint8_t *const stage_a = piece_home(piece0), *const stage_b = piece_home(piece0 + 1);     // (stage_b: C = 16 only)
    uint32_t *my_dirs = dirs + a;                         // word w at my_dirs[w * n]
    auto flush_codes = [&](int g16, auto together) {      // rounds 16 * g16 .. 16 * g16 + 15 of every alignment of the block
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // NO branch between the stores: the rows of a ragged last block past
        // alignment n - 1 hold that alignment's records once more (such lanes shadow it: same streams, same arithmetic) and
        // are stored on top of its row -- the same bytes.  With a test per store the block was read, wait, store eight times
        // over; now the LDS reads go out together.
        // (the flush after the last round, once per wavefront, goes row by row: it has no registers to spare for the rows)
        // the centre words and the outer words are staged apart, 16 rounds per alignment in a row: four lanes take one
        // alignment's row, four rounds each -- 16 alignments x 64 B of centre words, then of outer words, per pair of store
        // instructions, eight neighbouring lanes per 128-byte line.  (Staged as (centre, outer) pairs the words had to be
        // parted in registers: 32 moves per flush.)
        const int fa16 = lane >> 2, part = lane & 3;
        uint4 vc[A / 16], vo[A / 16];
        auto fetch = [&](int q) {
            vc[q] = *reinterpret_cast<const uint4 *>(&stage_codes[0][q * 16 + fa16][4 * part]);
            vo[q] = *reinterpret_cast<const uint4 *>(&stage_codes[1][q * 16 + fa16][4 * part]);
        };
        if constexpr (decltype(together)::value) {
#pragma unroll
            for (int q = 0; q < A / 16; ++q) fetch(q);
        }
        uint4 *centre = reinterpret_cast<uint4 *>(codes), *outer = centre + half_quads(n);
#pragma unroll
        for (int q = 0; q < A / 16; ++q) {
            if constexpr (!decltype(together)::value) fetch(q);
            const uint32_t row = min(block_first + (uint32_t)(q * 16 + fa16), n - 1);
            const size_t at = ((size_t)g16 * n + row) * kHalfQuads + part;
            centre[at] = vc[q];
            outer[at] = vo[q];
        }
        __builtin_amdgcn_wave_barrier();
    };
    // value of lane g-1 / g+1 of the same alignment (garbage at the slice ends: the callers select it away)
    auto from_prev = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, true); };
    auto from_next = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101 /* row_shl:1 */, 0xf, 0xf, true); };
    auto group_first = [](int v) {                        // lane 0 of the group, in every lane of the group
        return __builtin_amdgcn_update_dpp(0, v, G == 4 ? 0x00 /* quad_perm:[0,0,0,0] */ : 0xA0 /* [0,0,2,2] */, 0xf, 0xf, true);
    };
    auto group_last = [](int v) {
        return __builtin_amdgcn_update_dpp(0, v, G == 4 ? 0xFF /* quad_perm:[3,3,3,3] */ : 0xF5 /* [1,1,3,3] */, 0xf, 0xf, true);
    };
    auto group_max = [](int v) {
        int o = __builtin_amdgcn_update_dpp(0, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, true);
        v = v > o ? v : o;
        if (G == 4) {
            o = __builtin_amdgcn_update_dpp(0, v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, true);
            v = v > o ? v : o;
        }
        return v;
    };
    // bitwise select with an all-ones / all-zeros mask: ONE full-rate v_bitop3_b32 (truth table 0xCA = a ? b : c).  Written
    // as (b & m) | (c & ~m) hipcc picks v_bfi_b32, written as ?: v_cndmask_b32 -- both half rate on gfx950.
    auto pick = [](int m, int if_set, int if_clear) { return (int)__builtin_amdgcn_bitop3_b32((unsigned)m, (unsigned)if_set, (unsigned)if_clear, 0xCA); };

    const int lane_base = (g * C) << 2;                   // this lane's share of the cell index: added to the lane's maximum only
    const int first_mask = keep_opaque(is_first ? -1 : 0), last_mask = keep_opaque(is_last ? -1 : 0);
    // Clean cells of the previous round, and its shifted view S in TWO register sets that swap roles every round (the
    // round loop is unrolled by two): a single set costs a register copy per cell per round.
    // NV = C / 2 registers, register k = cells k (low half) and k + NV (high half); 0 = dropped.
    constexpr int NV = C / 2;
    int cur[NV], sp_a[NV + 1], sp_b[NV + 1];
#pragma unroll
    for (int c = 0; c < NV; ++c) cur[c] = 0;
#pragma unroll
    for (int c = 0; c <= NV; ++c) sp_a[c] = sp_b[c] = 0;
    if (is_last) cur[NV - 1] = ((kXDrop - kPkBase0) * kScale) << 16;         // band cell 31
    int off = kPkBase0;                                   // true value = stored value + off (off = base - round)
    sg_keep_f16_denormals();
    // round 0: pos_y = 0, pos_x = 31 -> cell k sits at row 31 - k (valid for k <= 30), column k - 31 (never valid)
    win_t aw = 0, bw = 0;
    {
        const unsigned long long w0 = stream_a[0], w1 = stream_a[kStreamStride];   // seq1[0..31]
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int i1 = 30 - (g * C + c);                                  // differs per lane: shifts, not indexing
            const unsigned ch = (unsigned)(((i1 & 16) ? w1 : w0) >> (4 * (i1 & 15))) & 15u;
            aw |= (win_t)(i1 >= 0 ? ch : kPadSeq1) << (4 * c);
            bw |= (win_t)kPadSeq2 << (4 * c);
        }
    }
    int pos_x = 31;
    int best = kXDrop, best_round = 0, best_kmax = 31 << 2, best_px = 31, last_round = 0;
    int alive_m = -1;                                     // all ones while the alignment is alive
    unsigned dir_word = 0;                                // move bits of the current 32 rounds (round r ends at bit r & 31)
    if (is_first) stage_codes[0][al][0] = stage_codes[1][al][0] = 0u;
    // This lane's character stream: the first slice feeds on seq1 (consumed when the band steps down), the last slice on
    // seq2 (consumed when it steps right); slices in between run the seq1 stream along without using it.
    SgStream feed;
    feed.start(is_last ? stream_b : stream_a, kStreamStride, is_last ? 0 : 31);       // next character: seq2[0] / seq1[31]

    // one round: reads the previous round's view `sp`, leaves this round's view in `sp_next`
    // round = base + K of a window of 8 rounds that starts at base (1 or 9 mod 16): what happens every 16th round is then a
    // test of one bit of `base` in ONE of the eight unrolled rounds, not a copy of the block behind each
    auto one_round = [&](auto calm_tag, auto k_tag, const int base, const int (&sp)[NV + 1], int (&sp_next)[NV + 1]) {
        constexpr bool kCalm = decltype(calm_tag)::value;   // a round of a calm window (see the loop below): no cell can fall under the threshold
        constexpr int K = decltype(k_tag)::value;
        const int round = base + K;
        // source.cpp:1895: band cell 0 (first lane, register 0, low half) against band cell 31 (last lane, last register, high half)
        // All of a round's decisions are arithmetic masks (sign of a difference, shifted down): a compare writes a scalar
        // register pair and the select that reads it waits for it -- with one or two wavefronts on the SIMD nothing fills that wait
        const int rmask = keep_opaque((int)((unsigned)group_first(cur[0]) & 0xFFFFu) - (int)((unsigned)group_last(cur[NV - 1]) >> 16)) >> 31;
        pos_x -= rmask;                                   // += 1 when the band steps right
        dir_word = __builtin_amdgcn_alignbit((unsigned)rmask, dir_word, 1);   // (dir_word >> 1) | (right << 31)
        // :1903, :1913: both still inside.  (Not asked in a calm round: every character such a window takes from the streams
        // is a base, i.e. lies inside its sequence -- pos_y + 30 < kLen and pos_x - 32 < kLen after the step -- far from these limits.)
        if constexpr (!kCalm) {
            const int pos_y = round - (pos_x - 31);
            alive_m &= (keep_opaque(pos_x - (32 + kLen + 31 + 1)) & keep_opaque(pos_y - (1 + kLen + 1))) >> 31;
        }
        // neighbours of the slice in the previous round's band
        // register "-1" = cells (-1, NV - 1), register "NV" = cells (NV, C): the neighbour lane's end cell in one half, this
        // lane's own middle cell in the other (0 = dropped past the band's ends)
        const unsigned p_lo = (unsigned)from_prev(cur[NV - 1]) & ~(unsigned)first_mask;          // high half: cell -1
        const unsigned p_hi = (unsigned)from_next(cur[0]) & ~(unsigned)last_mask;                // low half: cell C
        const int lo_in = (int)__builtin_amdgcn_alignbit((unsigned)cur[NV - 1], p_lo, 16);       // (cell -1, cell NV - 1)
        const int hi_in = (int)__builtin_amdgcn_alignbit(p_hi, (unsigned)cur[0], 16);            // (cell NV, cell C)
        // sequence windows follow the band
        {
            const unsigned cand = feed.next();            // the character entering: seq1[pos_y + 30] or seq2[pos_x - 32], pads included
            const unsigned a_top = (unsigned)(aw >> (4 * C - 4)), b_low = (unsigned)bw & 15u;
            // (the DPP moves are evaluated by ALL lanes before the select: inside one arm of `?:` they would run with the
            // source lanes masked off)
            const unsigned a_nb = (unsigned)from_prev((int)a_top), b_nb = (unsigned)from_next((int)b_low);
            const unsigned a_in = (unsigned)pick(first_mask, (int)cand, (int)a_nb);
            const unsigned b_in = (unsigned)pick(last_mask, (int)cand, (int)b_nb);
            // aw moves up one field when the band steps down, bw moves down one field when it steps right: a shift by 0 or 4
            // (one variable 64-bit shift each) and the entering character masked in -- not two shifted copies and a select
            const unsigned shift_a = 4u & ~(unsigned)rmask, shift_b = 4u & (unsigned)rmask;
            aw = (aw << shift_a) | (win_t)(a_in & ~(unsigned)rmask);
            bw = (bw >> shift_b) | ((win_t)(b_in & (unsigned)rmask) << (4 * C - 4));
            const int cmask = keep_opaque(~(rmask ^ last_mask));              // consume = is_last ? right : !right
            feed.used4 += cmask & 4;
        }
        // a field of aw ^ bw is 0..7 (the codes have three bits): bit 0 of  z | z >> 1 | z >> 2  = "the characters differ"
        const win_t z = aw ^ bw;
        const win_t differ = z | (z >> 1) | (z >> 2);                         // bit 4c clear: cell c is a match
        const unsigned df_lo = (unsigned)differ, df_hi = (unsigned)((unsigned long long)differ >> 32);

        unsigned tags = 0;                                // two tag bits per cell, cell c at bits 2c of the lane's record
        // match bits as bytes: cells 0, 2, 4, .. in `even`, cells 1, 3, 5, .. in `odd` (bit 0 of byte j = cell 2j / 2j + 1);
        // ONE v_perm_b32 per register then puts the low cell's byte into byte 1 and the high cell's into byte 3:
        // 256 = 2 * kScale in the half of a matching cell
        constexpr unsigned kByteOnes = 0x01010101u;
        const unsigned even_lo = ~df_lo & kByteOnes, odd_lo = ~(df_lo >> 4) & kByteOnes;
        const unsigned even_hi = C == 16 ? ~df_hi & kByteOnes : even_lo, odd_hi = C == 16 ? ~(df_hi >> 4) & kByteOnes : odd_lo;
        unsigned v[NV];
        // S of cells (k, k + NV), k = 0 .. NV: left of register k is sv[k], up is sv[k + 1]
        unsigned sv[NV + 1];
        sv[0] = (unsigned)pick(rmask, cur[0], lo_in);
#pragma unroll
        for (int k = 0; k < NV; ++k) sv[k + 1] = (unsigned)pick(rmask, k + 1 < NV ? cur[k + 1] : hi_in, cur[k]);
        // index and tag of the gap candidates: TWO registers per v_lshl_add_u64 (no half ever carries: all stay < 0x7C00).
        // left of (2m, 2m + 1) and up of (2m - 1, 2m) start from the same register pair (sv[2m], sv[2m + 1]).
        auto both = [](int k) { return (unsigned)(k << 2) | ((unsigned)((k + NV) << 2) << 16); };   // the two cell indices
        unsigned vl[NV], vu[NV];
#pragma unroll
        for (int m = 0; m < NV / 2; ++m) {
            const unsigned long long pair = (unsigned long long)sv[2 * m] | ((unsigned long long)sv[2 * m + 1] << 32);
            const unsigned long long l = pair + (((unsigned long long)(both(2 * m + 1) + 0x00010001u) << 32) | (both(2 * m) + 0x00010001u));
            vl[2 * m] = (unsigned)l;
            vl[2 * m + 1] = (unsigned)(l >> 32);
            if (m > 0) {
                const unsigned long long u = pair + (((unsigned long long)(both(2 * m) + 0x00020002u) << 32) | (both(2 * m - 1) + 0x00020002u));
                vu[2 * m - 1] = (unsigned)u;
                vu[2 * m] = (unsigned)(u >> 32);
            }
        }
        vu[0] = sv[1] + (both(0) + 0x00020002u);
        vu[NV - 1] = sv[NV] + (both(NV - 1) + 0x00020002u);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int dsel = pick(rmask, sp[k + 1], sp[k]);                                   // diagonal
            // cell k = byte k / 2 of even / odd (low word), cell k + NV = the same byte of the high word (C = 16) or
            // byte k / 2 + 2 of the same word (C = 8)
            const unsigned j = (unsigned)k >> 1;
            const unsigned sel = 0x000C000Cu | (j << 8) | ((C == 16 ? 4u + j : j + 2u) << 24);
            const unsigned f = __builtin_amdgcn_perm(k & 1 ? odd_hi : even_hi, k & 1 ? odd_lo : even_lo, sel);
            const unsigned vd = (unsigned)dsel + f + (both(k) + 0x00030003u + (unsigned)kScale * 0x10001u);   // dia + 3 / dia + 1, tag 3
            v[k] = sg_pk_max3(vd, vu[k], vl[k]);                                              // up + 0, tag 2; left + 0, tag 1
            sp_next[k] = (int)sv[k];
        }
        sp_next[NV] = (int)sv[NV];
        // band maximum: a tree of 3-way maxima (depth 2), not a chain
        unsigned kmax2;
        if constexpr (NV == 8) kmax2 = sg_pk_max3(sg_pk_max3(v[0], v[1], v[2]), sg_pk_max3(v[3], v[4], v[5]), sg_pk_max3(v[6], v[7], v[7]));
        else                   kmax2 = sg_pk_max3(sg_pk_max3(v[0], v[1], v[2]), v[3], v[3]);
        // tags, cell c at bits 2c of the lane's record.  C = 16: one v_perm_b32 gathers the low bytes of registers k and
        // k + 4 as (cell k, cell k + 4, cell k + 8, cell k + 12), one mask keeps the four tags, four such words shifted
        // together give byte j = cells 4j .. 4j + 3
        if constexpr (NV == 8) {
#pragma unroll
            for (int k = 3; k >= 0; --k)
                tags = (tags << 2) | (__builtin_amdgcn_perm(v[k + 4], v[k], 0x06020400u) & 0x03030303u);
        } else {
#pragma unroll
            for (int k = NV - 1; k >= 0; --k) tags = (tags << 2) | (v[k] & 0x00030003u);      // cell k at bits 2k, cell k + NV at 16 + 2k
        }
        const unsigned k_lo = kmax2 & 0xFFFFu, k_hi = kmax2 >> 16;
        int kmax = (int)(k_lo > k_hi ? k_lo : k_hi) + lane_base;
#pragma unroll
        for (int k = 0; k < NV; ++k) cur[k] = (int)v[k];                  // (tagged until the X-drop pass below cleans it)
        kmax = group_max(kmax);
        // stored values are offset: true = stored + off.  A candidate derived from live cells is >= kPkFloor - 2 >= 4, one
        // derived only from dropped cells (0) is <= 3 and stands for "<= 0" (the reference's guard, source.cpp:1922-1924)
        --off;
        const int stored = kmax >> 7, band_best = stored + off;
        // (calm: the band maximum is a live cell, true value >= the threshold >= 1)
        const int round_best = kCalm ? band_best : max(band_best, 0) & (keep_opaque(3 - stored) >> 31);
        const int imask = alive_m & (keep_opaque(best - round_best) >> 31);  // improved = alive && round_best > best (:1933-1936)
        best = pick(imask, round_best, best);
        best_round = pick(imask, round, best_round);
        // where: the winning value with its cell index (the highest cell among equals: where the search of :1957 stops) and the
        // band's column -- the cell and the row are taken out of them once, after the last round
        best_kmax = pick(imask, kmax, best_kmax);
        best_px = pick(imask, pos_x, best_px);
        const int thr_true = best - kXDrop > 1 ? best - kXDrop : 1;           // :1938-1941, and "0 means dropped"
        if constexpr (kCalm) {
#pragma unroll
            for (int k = 0; k < NV; ++k) cur[k] &= (int)~((unsigned)(kScale - 1) * 0x10001u);     // every cell stays above the threshold
        } else {
            const unsigned thr2 = __umul24((unsigned)(thr_true - off), (unsigned)kScale * 0x10001u);     // v_mul_u32_u24, full rate
#pragma unroll
            for (int k = 0; k < NV; ++k)              // v_pk_sub_i16, v_pk_ashrrev_i16, v_bitop3 (0x20 = a & ~b & c): dropped -> 0
                cur[k] = (int)__builtin_amdgcn_bitop3_b32((unsigned)cur[k], sg_pk_below((unsigned)cur[k], thr2), ~((unsigned)(kScale - 1) * 0x10001u), 0x20);
        }
        if (C == 8) {                                     // cells k at bits 2k, cells k + 4 at bits 16 + 2k: one piece of 8 cells
            *reinterpret_cast<uint16_t *>(stage_a + 4 * (round & 15)) = (uint16_t)((tags & 0xFFu) | ((tags >> 8) & 0xFF00u));
        } else {                                          // cells 0 .. 7 of the lane in the low half, 8 .. 15 in the high half: two pieces
            *reinterpret_cast<uint16_t *>(stage_a + 4 * (round & 15)) = (uint16_t)tags;
            *reinterpret_cast<uint16_t *>(stage_b + 4 * (round & 15)) = (uint16_t)(tags >> 16);
        }
        if (K == 6 && (base & 8)) {                       // round = 15 (mod 16): same place for every lane of the wavefront
            flush_codes(round >> 4, std::true_type());
            if ((round & 31) == 31) {
                if (real && is_first) *my_dirs = dir_word;
                my_dirs += n;
            }
            // re-base: the threshold (which climbs by one or two a round in stored terms) goes back to kPkFloor; dropped
            // cells stay 0 (saturating), live ones -- this round's and the view of the round before -- are well above
            const int delta = (thr_true - off) - kPkFloor;
            const unsigned d2 = __umul24((unsigned)delta, (unsigned)kScale * 0x10001u);
#pragma unroll
            for (int k = 0; k < NV; ++k) cur[k] = (int)sg_pk_sub_sat((unsigned)cur[k], d2);
#pragma unroll
            for (int k = 0; k <= NV; ++k) sp_next[k] = (int)sg_pk_sub_sat((unsigned)sp_next[k], d2);
            off += delta;
        }
        if (K == 7 && (base & 8)) feed.top_up(kStreamStride);                // after round 0 (mod 16); a stream gives at most 16 characters in 16 rounds
        if constexpr (!kCalm) alive_m &= keep_opaque(-round_best) >> 31;      // alive && round_best != 0 (:1943-1946)
        last_round = round;
    };

    // CALM WINDOWS.  Every kCalmWindow rounds the wavefront asks whether any of its live alignments has a band cell closer
    // than kCalmMargin to the X-drop threshold.  If none has, no cell can be dropped in the next kCalmWindow rounds: in stored
    // terms a cell never falls below the lowest cell of the band before it (every cell has a gap parent inside the band, and
    // a gap step adds 0), while the threshold climbs by one per round plus the rise of `best`, which is at most one per two
    // rounds (a diagonal step takes two) -- 12 in 8 rounds.  Those rounds then run without the X-drop test (one v_and per
    // register instead of compare, mask, apply) and without the guards for a dropped band maximum: the same results by
    // construction.  A band next to its threshold, a dropped cell anywhere in it (the first rounds of every alignment), a pad
    // in reach or exact_only send the window down the exact path.  Two loops, one per kind of window, each running for as long as its
    // kind lasts: hipcc gives the two bodies different register assignments, and the moves between them are paid only where
    // the kind changes.  ("Has every alignment of the wavefront ended" is asked at the same place, not per round: a vector
    // compare feeding a scalar branch drains the wavefront's pipeline; the rounds a finished wavefront runs on change nothing
    // -- no alignment is alive to improve, and the records of rounds after an alignment's best round are never read.)
    constexpr int kCalmWindow = 8, kCalmMargin = kCalmWindow + kCalmWindow / 2 + 1;
    static_assert((kMaxRound - 1) % 16 == 0 && kCalmWindow == 8, "whole windows; one_round places the 16-round events by K and base");
    auto window_kind = [&]() -> int {                     // 1 calm, 0 exact, -1 every alignment of the wavefront has ended
        if (!__any(alive_m != 0)) return -1;
        unsigned low;                                     // the lowest of this lane's cells (the vote below covers the band's other lanes)
        if constexpr (NV == 8) low = sg_pk_min3(sg_pk_min3((unsigned)cur[0], (unsigned)cur[1], (unsigned)cur[2]), sg_pk_min3((unsigned)cur[3], (unsigned)cur[4], (unsigned)cur[5]),
                                                sg_pk_min3((unsigned)cur[6], (unsigned)cur[7], (unsigned)cur[7]));
        else                   low = sg_pk_min3(sg_pk_min3((unsigned)cur[0], (unsigned)cur[1], (unsigned)cur[2]), (unsigned)cur[3], (unsigned)cur[3]);
        const int low_stored = (int)min(low & 0xFFFFu, low >> 16) >> 7;
        const int thr_now = (best - kXDrop > 1 ? best - kXDrop : 1) - off;
        // ... and bases only, in the slice's windows and among the at most 8 characters the window takes from the stream (a
        // pad of the matrix's edges or a byte that was not 0..3 has bit 2 set): the position tests are then idle, see one_round
        const unsigned long long w64 = (unsigned long long)(aw | bw) | (feed.sreg >> feed.used4 & 0xFFFFFFFFull);
        const unsigned fields = (unsigned)w64 | (unsigned)(w64 >> 32);
        const bool edgy = alive_m != 0 && (low_stored < thr_now + kCalmMargin || (fields & 0x44444444u) != 0);
        return !exact_only && !__any(edgy) ? 1 : 0;
    };
    int round = 1, kind = window_kind();
    uint32_t calm_windows = 0;                            // (scalar: the loops are wavefront-uniform)
#define SWMI_SG_WINDOW(CALM)                                                                                      \
    one_round(CALM(), std::integral_constant<int, 0>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 1>(), round, sp_b, sp_a);                                        \
    one_round(CALM(), std::integral_constant<int, 2>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 3>(), round, sp_b, sp_a);                                        \
    one_round(CALM(), std::integral_constant<int, 4>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 5>(), round, sp_b, sp_a);                                        \
    one_round(CALM(), std::integral_constant<int, 6>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 7>(), round, sp_b, sp_a);                                        \
    round += kCalmWindow
    while (kind >= 0) {
        // vmcnt(0) (gfx9 encoding; the builtin, not inline assembly: hipcc's wait-count pass has to see it) in the block that
        // enters each loop.  The pass merges what is in flight where the loops meet: a stream word still on its way when one loop
        // hands over would put an s_waitcnt vmcnt into every trip of the other, where that loop first writes the register the
        // word was loaded into -- and that wait then catches the loop's OWN top-up load, one memory latency per 16 rounds.
        if (kind == 1) {
            __builtin_amdgcn_s_waitcnt(0x0070);           // (+ lgkmcnt(0): two identical waits hipcc hoists above the branch -- where they are no use)
            do {
                SWMI_SG_WINDOW(std::true_type);
                ++calm_windows;
                kind = round < kMaxRound ? window_kind() : -1;
            } while (kind == 1);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            do {
                SWMI_SG_WINDOW(std::false_type);
                kind = round < kMaxRound ? window_kind() : -1;
            } while (kind == 0);
        }
    }
#undef SWMI_SG_WINDOW
    if (lane == 0) {                                      // windows run / of them calm, summed over the launch (swmi_semiglobal_window_stats)
        atomicAdd(&window_stats[0], (uint32_t)(round - 1) / kCalmWindow);
        atomicAdd(&window_stats[1], calm_windows);
    }
    if ((last_round & 15) != 15) flush_codes(last_round >> 4, std::false_type());
    if (real && is_first) {
        if ((last_round & 31) != 31) *my_dirs = dir_word >> (31 - (last_round & 31));
        summary[a] = make_int4(best - kXDrop, best_round, (best_kmax >> 2) & 31, best_round - (best_px - 31));     // (pos_y = round - right steps)
    }
}

// ---- sweep, ONE lane per alignment (the whole band in 16 registers of two cells): the largest batches ---------------
//
// The cell of the split sweep above with G = 1: 64 alignments per wavefront and nothing crosses lanes -- the band's end
// cells, its neighbours, both character streams and the band maximum are the lane's own.  What a round costs besides the
// cells (direction, windows, match bits, best / threshold bookkeeping: ~80 instructions) is paid once per 64 alignments
// instead of once per 32, so a batch that still gives every SIMD two or more of these wavefronts runs a third faster.
// Register k = cells k (low half) and k + 16 (high half); the windows are 2 x 64 bits per sequence.

template <int W>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W)))
sg_forward_lane_kernel(const unsigned long long *__restrict__ streams, uint32_t n,
                       uint32_t *__restrict__ codes, uint32_t *__restrict__ dirs, int4 *__restrict__ summary, int exact_only,
                       uint32_t *__restrict__ window_stats)
{
    constexpr int NV = 16, A = 64;
    __shared__ __attribute__((aligned(16))) uint32_t stage_codes[2][A][kStagePitch];   // [centre / outer word][alignment of the block][round & 15] (+ 4: see the flush)
    const int lane = threadIdx.x;
    const uint32_t block_first = blockIdx.x * A;
    const uint32_t a0 = block_first + lane;
    const bool real = a0 < n;
    const uint32_t a = real ? a0 : n - 1;
    constexpr size_t kStreamStride = 2 * A;
    const unsigned long long *stream_a = streams + (size_t)blockIdx.x * kStreamWords * kStreamStride + 2 * lane;
    uint32_t *my_dirs = dirs + a;                         // word w at my_dirs[w * n]
    auto flush_codes = [&](int g16, auto together) {      // rounds 16 * g16 .. 16 * g16 + 15 of every alignment of the block
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // NO branch between the stores: the rows of a ragged last block past
        // alignment n - 1 hold that alignment's records once more (such lanes shadow it: same streams, same arithmetic) and
        // are stored on top of its row -- the same bytes.  With a test per store the block was read, wait, store eight times
        // over; now the LDS reads go out together.
        // (the flush after the last round, once per wavefront, goes row by row: it has no registers to spare for the rows)
        // the centre words and the outer words are staged apart, 16 rounds per alignment in a row: four lanes take one
        // alignment's row, four rounds each -- 16 alignments x 64 B of centre words, then of outer words, per pair of store
        // instructions, eight neighbouring lanes per 128-byte line.  (Staged as (centre, outer) pairs the words had to be
        // parted in registers: 32 moves per flush.)
        const int fa16 = lane >> 2, part = lane & 3;
        uint4 vc[A / 16], vo[A / 16];
        auto fetch = [&](int q) {
            vc[q] = *reinterpret_cast<const uint4 *>(&stage_codes[0][q * 16 + fa16][4 * part]);
            vo[q] = *reinterpret_cast<const uint4 *>(&stage_codes[1][q * 16 + fa16][4 * part]);
        };
        if constexpr (decltype(together)::value) {
#pragma unroll
            for (int q = 0; q < A / 16; ++q) fetch(q);
        }
        uint4 *centre = reinterpret_cast<uint4 *>(codes), *outer = centre + half_quads(n);
#pragma unroll
        for (int q = 0; q < A / 16; ++q) {
            if constexpr (!decltype(together)::value) fetch(q);
            const uint32_t row = min(block_first + (uint32_t)(q * 16 + fa16), n - 1);
            const size_t at = ((size_t)g16 * n + row) * kHalfQuads + part;
            centre[at] = vc[q];
            outer[at] = vo[q];
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto pick = [](int m, unsigned if_set, unsigned if_clear) { return __builtin_amdgcn_bitop3_b32((unsigned)m, if_set, if_clear, 0xCA); };

    sg_keep_f16_denormals();
    unsigned cur[NV], sp_a[NV + 1], sp_b[NV + 1];
#pragma unroll
    for (int k = 0; k < NV; ++k) cur[k] = 0;
#pragma unroll
    for (int k = 0; k <= NV; ++k) sp_a[k] = sp_b[k] = 0;
    cur[NV - 1] = (unsigned)((kXDrop - kPkBase0) * kScale) << 16;             // band cell 31
    int off = kPkBase0;                                   // true value = stored value + off (off = base - round)
    // round 0: cell c sits at row 31 - c (valid for c <= 30), column c - 31 (never valid); windows: cell c <-> field c & 15 of word c / 16
    unsigned long long aw0 = 0, aw1 = 0, bw0 = kPadSeq2 * 0x1111111111111111ull, bw1 = bw0;
    {
        const unsigned long long w0 = stream_a[0], w1 = stream_a[kStreamStride];   // seq1[0..31]
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int i1 = 30 - c;
            const unsigned long long ch = i1 >= 0 ? (((i1 & 16) ? w1 : w0) >> (4 * (i1 & 15))) & 15ull : (unsigned long long)kPadSeq1;
            if (c < 16) aw0 |= ch << (4 * c);
            else        aw1 |= ch << (4 * (c - 16));
        }
    }
    SgStream sa, sb;                                      // seq1 is consumed when the band steps down, seq2 when it steps right
    sa.start(stream_a, kStreamStride, 31);
    sb.start(stream_a + 1, kStreamStride, 0);
    int pos_x = 31;
    int best = kXDrop, best_round = 0, best_kmax = 31 << 2, best_px = 31, last_round = 0;
    int alive_m = -1;                                     // all ones while the alignment is alive
    unsigned dir_word = 0;                                // move bits of the current 32 rounds (round r ends at bit r & 31)
    stage_codes[0][lane][0] = stage_codes[1][lane][0] = 0u;

    // the calm loop's windows (below): 2 bits per cell, cell c at bits 2c -- bases only
    unsigned long long a2 = 0, b2 = 0;
    // round = base + K of a window of 8 rounds that starts at base (1 or 9 mod 16): what happens every 16th round is then a
    // test of one bit of `base` in ONE of the eight unrolled rounds, not a copy of the block behind each
    auto one_round = [&](auto calm_tag, auto k_tag, const int base, const unsigned (&sp)[NV + 1], unsigned (&sp_next)[NV + 1]) {
        constexpr bool kCalm = decltype(calm_tag)::value;   // a round of a calm window (below): no cell can fall under the threshold
        constexpr int K = decltype(k_tag)::value;
        const int round = base + K;
        // source.cpp:1895: band cell 0 against band cell 31.  All of a round's decisions are arithmetic masks (sign of a
        // difference, shifted down): a compare writes a scalar register pair and the select that reads it waits for it, and
        // with one wavefront on the SIMD nothing fills that wait
        const int rmask = keep_opaque((int)(cur[0] & 0xFFFFu) - (int)(cur[NV - 1] >> 16)) >> 31;
        const unsigned rm = (unsigned)rmask, dm = ~rm;
        pos_x -= rmask;
        dir_word = __builtin_amdgcn_alignbit(rm, dir_word, 1);
        // :1903, :1913: both still inside.  (Not asked in a calm round: every character such a window takes from the streams
        // is a base, i.e. lies inside its sequence -- pos_y + 30 < kLen and pos_x - 32 < kLen after the step -- far from these limits.)
        if constexpr (!kCalm) {
            const int pos_y = round - (pos_x - 31);
            alive_m &= (keep_opaque(pos_x - (32 + kLen + 31 + 1)) & keep_opaque(pos_y - (1 + kLen + 1))) >> 31;
        }
        // the windows follow the band: seq1's moves up one field on a step down, seq2's down one field on a step right
        constexpr unsigned kByteOnes = 0x01010101u;
        unsigned match_lo[4], match_hi[4];                // bit 0 of byte j: a cell of the low / high half of the band matches
        if constexpr (kCalm) {
            // 2-bit fields, one 64-bit window per sequence: one shift each; "the characters differ" = z | z >> 1
            const unsigned a_in = sa.next(), b_in = sb.next();
            a2 = (a2 << (2u & dm)) | (a_in & dm);
            b2 = (b2 >> (2u & rm)) | ((unsigned long long)(b_in & rm) << 62);
            sa.used4 += 4 & dm;
            sb.used4 += 4 & rm;
            const unsigned long long z = a2 ^ b2;
            const unsigned z_lo = (unsigned)z, z_hi = (unsigned)(z >> 32);
            const unsigned same_lo = __builtin_amdgcn_bitop3_b32(z_lo, z_lo >> 1, 0u, 0x03);      // ~(a | b): bit 2c set = cell c matches
            const unsigned same_hi = __builtin_amdgcn_bitop3_b32(z_hi, z_hi >> 1, 0u, 0x03);
#pragma unroll
            for (int p = 0; p < 4; ++p) {                 // byte j of phase p = cell 4j + p (low word) / 16 + 4j + p (high word)
                match_lo[p] = (same_lo >> (2 * p)) & kByteOnes;
                match_hi[p] = (same_hi >> (2 * p)) & kByteOnes;
            }
        } else {
            const unsigned a_in = sa.next(), b_in = sb.next();
            const unsigned shift_a = 4u & dm, shift_b = 4u & rm;
            const unsigned a_carry = (unsigned)(aw0 >> 60) & dm, b_carry = (unsigned)bw1 & 15u & rm;
            aw1 = (aw1 << shift_a) | a_carry;
            aw0 = (aw0 << shift_a) | (a_in & dm);
            bw0 = (bw0 >> shift_b) | ((unsigned long long)b_carry << 60);
            bw1 = (bw1 >> shift_b) | ((unsigned long long)(b_in & rm) << 60);
            sa.used4 += 4 & dm;
            sb.used4 += 4 & rm;
            // bit 4c of a word clear: its cell c is a match (three-bit codes: bit 0 of z | z >> 1 | z >> 2 = "they differ")
            const unsigned long long z0 = aw0 ^ bw0, z1 = aw1 ^ bw1;
            const unsigned long long d0 = z0 | (z0 >> 1) | (z0 >> 2), d1 = z1 | (z1 >> 1) | (z1 >> 2);
            const unsigned dw[4] = {(unsigned)d0, (unsigned)(d0 >> 32), (unsigned)d1, (unsigned)(d1 >> 32)};    // cells 8w .. 8w + 7
            // [2 (w & 1) + parity]: byte j = cell 8w + 2j + parity of the low (w < 2) / high half
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned *dst = w < 2 ? match_lo : match_hi;
                dst[2 * (w & 1)] = ~dw[w] & kByteOnes;
                dst[2 * (w & 1) + 1] = __builtin_amdgcn_bitop3_b32(dw[w] >> 4, kByteOnes, 0u, 0x0C);       // ~a & b in one (hipcc: v_not, shift, v_and)
            }
        }
        // S of cells (k, k + 16), k = 0 .. 16: left of register k is sv[k], up is sv[k + 1]; past the band's ends: dropped (0)
        unsigned sv[NV + 1];
        sv[0] = pick(rmask, cur[0], cur[NV - 1] << 16);                       // (cell -1, cell 15)
#pragma unroll
        for (int k = 0; k < NV; ++k) sv[k + 1] = pick(rmask, k + 1 < NV ? cur[k + 1] : cur[0] >> 16 /* (cell 16, cell 32) */, cur[k]);
        auto both = [](int k) { return (unsigned)(k << 2) | ((unsigned)((k + NV) << 2) << 16); };   // the two cell indices
        unsigned vl[NV], vu[NV], v[NV];
#pragma unroll
        for (int m = 0; m < NV / 2; ++m) {                // two registers per v_lshl_add_u64 (no half ever carries)
            const unsigned long long pair = (unsigned long long)sv[2 * m] | ((unsigned long long)sv[2 * m + 1] << 32);
            const unsigned long long l = pair + (((unsigned long long)(both(2 * m + 1) + 0x00010001u) << 32) | (both(2 * m) + 0x00010001u));
            vl[2 * m] = (unsigned)l;
            vl[2 * m + 1] = (unsigned)(l >> 32);
            if (m > 0) {
                const unsigned long long u = pair + (((unsigned long long)(both(2 * m) + 0x00020002u) << 32) | (both(2 * m - 1) + 0x00020002u));
                vu[2 * m - 1] = (unsigned)u;
                vu[2 * m] = (unsigned)(u >> 32);
            }
        }
        vu[0] = sv[1] + (both(0) + 0x00020002u);
        vu[NV - 1] = sv[NV] + (both(NV - 1) + 0x00020002u);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const unsigned dsel = pick(rmask, sp[k + 1], sp[k]);              // diagonal (:1897 / :1908)
            // where cell k's match byte lies (cell k + 16: the same place of match_hi): calm, phase k % 4 byte k / 4; exact,
            // word k / 8 parity k % 2 byte (k % 8) / 2
            const unsigned which = kCalm ? (unsigned)(k & 3) : 2u * ((unsigned)k >> 3) + ((unsigned)k & 1u);
            const unsigned j = kCalm ? (unsigned)k >> 2 : (unsigned)(k & 7) >> 1;
            const unsigned sel = 0x000C000Cu | (j << 8) | ((4u + j) << 24);
            const unsigned f = __builtin_amdgcn_perm(match_hi[which], match_lo[which], sel);
            const unsigned vd = dsel + f + (both(k) + 0x00030003u + (unsigned)kScale * 0x10001u);     // dia + 3 / dia + 1, tag 3
            v[k] = sg_pk_max3(vd, vu[k], vl[k]);
            sp_next[k] = sv[k];
        }
        sp_next[NV] = sv[NV];
        const unsigned kmax2 = sg_pk_max3(sg_pk_max3(sg_pk_max3(v[0], v[1], v[2]), sg_pk_max3(v[3], v[4], v[5]), sg_pk_max3(v[6], v[7], v[8])),
                                          sg_pk_max3(sg_pk_max3(v[9], v[10], v[11]), sg_pk_max3(v[12], v[13], v[14]), v[15]), v[15]);
        // tags: the low bytes of registers k and k + 4 (k + 8 and k + 12) gathered as (cell k, k + 4, k + 16, k + 20), masked and
        // shifted together: byte j of the two accumulators = four consecutive cells; two more gathers sort the bytes into the
        // record's centre and outer word
        unsigned acc_a = 0, acc_b = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            acc_a = (acc_a << 2) | (__builtin_amdgcn_perm(v[k + 4], v[k], 0x06020400u) & 0x03030303u);
            acc_b = (acc_b << 2) | (__builtin_amdgcn_perm(v[k + 12], v[k + 8], 0x06020400u) & 0x03030303u);
        }
        // (acc_a bytes: cells 0-3, 4-7, 16-19, 20-23; acc_b bytes: 8-11, 12-15, 24-27, 28-31)
        const unsigned tags_centre = __builtin_amdgcn_perm(acc_b, acc_a, 0x03020504u);   // cells 8 .. 23, cell 8 + k at bits 2k
        const unsigned tags_outer = __builtin_amdgcn_perm(acc_b, acc_a, 0x07060100u);    // cells 0 .. 7 | 24 .. 31
        const unsigned k_lo = kmax2 & 0xFFFFu, k_hi = kmax2 >> 16;
        const int kmax = (int)(k_lo > k_hi ? k_lo : k_hi);
        // true = stored + off; a candidate derived from live cells is >= kPkFloor - 2 >= 4, one derived only from dropped cells
        // (0) is <= 3 and stands for "<= 0" (the reference's guard, source.cpp:1922-1924)
        --off;
        const int stored = kmax >> 7, band_best = stored + off;
        // (calm: the band maximum is a live cell, true value >= the threshold >= 1)
        const int round_best = kCalm ? band_best : max(band_best, 0) & (keep_opaque(3 - stored) >> 31);
        const int imask = alive_m & (keep_opaque(best - round_best) >> 31);  // improved = alive && round_best > best (:1933-1936)
        best = (int)pick(imask, (unsigned)round_best, (unsigned)best);
        best_round = (int)pick(imask, (unsigned)round, (unsigned)best_round);
        // where: the winning value with its cell index (the highest cell among equals, :1957) and the band's column -- the cell
        // and the row are taken out of them once, after the last round
        best_kmax = (int)pick(imask, (unsigned)kmax, (unsigned)best_kmax);
        best_px = (int)pick(imask, (unsigned)pos_x, (unsigned)best_px);
        const int thr_true = best - kXDrop > 1 ? best - kXDrop : 1;           // :1938-1941
        if constexpr (kCalm) {                            // every cell stays above the threshold
#pragma unroll
            for (int k = 0; k < NV; ++k) cur[k] = v[k] & ~((unsigned)(kScale - 1) * 0x10001u);
        } else {
            const unsigned thr2 = __umul24((unsigned)(thr_true - off), (unsigned)kScale * 0x10001u);
#pragma unroll
            for (int k = 0; k < NV; ++k)                  // v_pk_sub_i16, v_pk_ashrrev_i16, v_bitop3 (0x20 = a & ~b & c): dropped -> 0
                cur[k] = __builtin_amdgcn_bitop3_b32(v[k], sg_pk_below(v[k], thr2), ~((unsigned)(kScale - 1) * 0x10001u), 0x20);
            alive_m &= keep_opaque(-round_best) >> 31;    // alive && round_best != 0 (:1943-1946)
        }
        stage_codes[0][lane][round & 15] = tags_centre;   // (one ds_write2st64_b32)
        stage_codes[1][lane][round & 15] = tags_outer;
        if (K == 6 && (base & 8)) {                       // round = 15 (mod 16): same place for every lane of the wavefront
            flush_codes(round >> 4, std::true_type());
            if ((round & 31) == 31) {
                if (real) *my_dirs = dir_word;
                my_dirs += n;
            }
            // re-base: the threshold goes back to kPkFloor (see the split sweep)
            const int delta = (thr_true - off) - kPkFloor;
            const unsigned d2 = __umul24((unsigned)delta, (unsigned)kScale * 0x10001u);
#pragma unroll
            for (int k = 0; k < NV; ++k) cur[k] = sg_pk_sub_sat(cur[k], d2);
#pragma unroll
            for (int k = 0; k <= NV; ++k) sp_next[k] = sg_pk_sub_sat(sp_next[k], d2);
            off += delta;
        }
        // the streams are topped up after round 0 (mod 16): a window of 8 rounds, which starts at 1 or 9 (mod 16), then finds
        // all the characters it can consume in `sreg` (the calm test below looks at them)
        if (K == 7 && (base & 8)) {
            sa.top_up(kStreamStride);
            sb.top_up(kStreamStride);
        }
        last_round = round;
    };

    // CALM WINDOWS.  Every kCalmWindow rounds the wavefront asks whether any of its live alignments has a band cell closer
    // than kCalmMargin to the X-drop threshold.  If none has, no cell can be dropped in the next kCalmWindow rounds: in stored
    // terms a cell never falls below the lowest cell of the band before it (every cell has a gap parent inside the band, and
    // a gap step adds 0), while the threshold climbs by one per round plus the rise of `best`, which is at most one per two
    // rounds (a diagonal step takes two) -- 12 in 8 rounds.  Those rounds then run without the X-drop test (one v_and per
    // register instead of compare, mask, apply) and without the guards for a dropped band maximum: the same results by
    // construction, 15 % fewer instructions.  A band next to its threshold, a dropped cell anywhere in it (the first rounds of
    // every alignment) or exact_only send the window down the exact path.  Two loops, one per kind of window, each running
    // for as long as its kind lasts: hipcc gives the two bodies different register assignments, and the ~180 moves between them
    // are paid only where the kind changes.
    constexpr int kCalmWindow = 8, kCalmMargin = kCalmWindow + kCalmWindow / 2 + 1;
    static_assert((kMaxRound - 1) % 16 == 0 && kCalmWindow == 8, "whole windows; one_round places the 16-round events by K and base");
    // In this kernel a calm window also needs BASES ONLY -- in the band's windows now and among the at most 8 characters of
    // either stream it will consume (no pad of the matrix's edges, no byte that was not 0..3): its rounds keep the sequence
    // windows as 2 bits per cell, 64 bits per sequence, and shift, compare and spread half as many registers.
    auto window_kind = [&](auto from_calm) -> int {       // 1 calm, 0 exact, -1 every alignment of the wavefront has ended
        if (!__any(alive_m != 0)) return -1;
        const unsigned low = sg_pk_min3(sg_pk_min3(sg_pk_min3(cur[0], cur[1], cur[2]), sg_pk_min3(cur[3], cur[4], cur[5]), sg_pk_min3(cur[6], cur[7], cur[8])),
                                        sg_pk_min3(sg_pk_min3(cur[9], cur[10], cur[11]), sg_pk_min3(cur[12], cur[13], cur[14]), cur[15]), cur[15]);
        const int low_stored = (int)min(low & 0xFFFFu, low >> 16) >> 7;
        const int thr_now = (best - kXDrop > 1 ? best - kXDrop : 1) - off;
        unsigned fields = (unsigned)(sa.sreg >> sa.used4) | (unsigned)(sb.sreg >> sb.used4);   // the next 8 characters of both streams
        if constexpr (!decltype(from_calm)::value) {      // (the 2-bit windows of the calm loop hold bases by construction)
            const unsigned long long w = aw0 | aw1 | bw0 | bw1;
            fields |= (unsigned)w | (unsigned)(w >> 32);
        }
        const bool edgy = alive_m != 0 && (low_stored < thr_now + kCalmMargin || (fields & 0x44444444u) != 0);
        return !exact_only && !__any(edgy) ? 1 : 0;
    };
    // 16 fields of 4 bits (values 0..3) <-> 16 fields of 2 bits; only where the kind of window changes
    auto squeeze2 = [](unsigned long long x) -> unsigned long long {
        x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
        x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
        x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
        return (x | (x >> 16)) & 0xFFFFFFFFull;
    };
    auto spread2 = [](unsigned long long x) -> unsigned long long {
        x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
        x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
        x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
        return (x | (x << 2)) & 0x3333333333333333ull;
    };
    int round = 1, kind = window_kind(std::false_type());
    uint32_t calm_windows = 0;                            // (scalar: the loops are wavefront-uniform)
#define SWMI_SG_WINDOW(CALM)                                                                                      \
    one_round(CALM(), std::integral_constant<int, 0>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 1>(), round, sp_b, sp_a);                                        \
    one_round(CALM(), std::integral_constant<int, 2>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 3>(), round, sp_b, sp_a);                                        \
    one_round(CALM(), std::integral_constant<int, 4>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 5>(), round, sp_b, sp_a);                                        \
    one_round(CALM(), std::integral_constant<int, 6>(), round, sp_a, sp_b);                                        \
    one_round(CALM(), std::integral_constant<int, 7>(), round, sp_b, sp_a);                                        \
    round += kCalmWindow
    while (kind >= 0) {
        if (kind == 1) {
            a2 = squeeze2(aw0) | (squeeze2(aw1) << 32);
            b2 = squeeze2(bw0) | (squeeze2(bw1) << 32);
            __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0) in the block that enters the loop: see the split sweep
            do {
                SWMI_SG_WINDOW(std::true_type);
                ++calm_windows;
                kind = round < kMaxRound ? window_kind(std::true_type()) : -1;
            } while (kind == 1);
            aw0 = spread2(a2 & 0xFFFFFFFFull);
            aw1 = spread2(a2 >> 32);
            bw0 = spread2(b2 & 0xFFFFFFFFull);
            bw1 = spread2(b2 >> 32);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);
            do {
                SWMI_SG_WINDOW(std::false_type);
                kind = round < kMaxRound ? window_kind(std::false_type()) : -1;
            } while (kind == 0);
        }
    }
#undef SWMI_SG_WINDOW
    if (lane == 0) {                                      // windows run / of them calm, summed over the launch (swmi_semiglobal_window_stats)
        atomicAdd(&window_stats[0], (uint32_t)(round - 1) / kCalmWindow);
        atomicAdd(&window_stats[1], calm_windows);
    }
    if ((last_round & 15) != 15) flush_codes(last_round >> 4, std::false_type());
    if (real) {
        if ((last_round & 31) != 31) *my_dirs = dir_word >> (31 - (last_round & 31));
        summary[a] = make_int4(best - kXDrop, best_round, (best_kmax >> 2) & 31, best_round - (best_px - 31));     // (pos_y = round - right steps)
    }
}

// Traceback, two kernels.
//
// sg_walk_lane_kernel: one LANE per walk (rounds 1-2 also had a walker with one wavefront per alignment, y and x in scalar
// registers, for small batches: ~40 scalar instructions per step -- this pair beats it down to a batch of one alignment,
// profiles/r03_sg_traceback_small.txt, and it is gone).  The 64 walks of a wavefront
// move in LOCKSTEP BY WINDOW of 16 rounds = one 128-byte line of code records per walk: all lanes consume window w (each at
// its own pace, 8..16 steps) out of LDS, then the whole wavefront swaps in the line of window w-1.  (Refilling per lane,
// whenever a walk left its line, made almost every step wait for some lane's load: 64 walks at random phases.)
// The kernel is bound by the latency of those line fetches -- 1024 wavefronts, one per SIMD, are all a 65536-walk batch
// gives -- so what counts is bytes in flight: THREE windows per walk are requested ahead (three register buffers that
// rotate through the roles, the loop is unrolled by three; 384 bytes per walk in flight; round 2: 64-byte half lines, one
// ahead, 3.5 TB/s).  The walk does not write positions -- their index in the ascending list is unknown until (0,0) is
// reached -- but its moves, 2 bits per step (the tags: 3 diag, 2 up, 1 left), 8 KB per alignment at most.
//
// sg_expand_kernel: one wavefront per alignment turns the moves (the tags) into the (i, j) list of source.cpp:1951-1975, in
// ascending order from (0,0) (see the kernel).
constexpr int kMoveWords = (kMaxRound / 32 + 1 + 15) & ~15;   // uint64 words of 32 moves per alignment (1025), padded to whole
                                                         // 128-byte lines (1040): a walk's words leave sixteen at a time, one line
constexpr int kLinePitch = kHalfQuads + 1;               // LDS row pitch in uint4 (a piece + one: conflict-free 16-byte accesses)

// Inside a window the walks do not loop over their own steps (8 .. 16 of them, a different count per lane, each step one
// chain LDS read -> decode -> (y, x) -> next address: ~400 cycles per step with one wavefront on the SIMD, 3.78 ms for 65536
// walks).  Every lane takes its line into 32 registers and the wavefront goes through the window's SIXTEEN ROUNDS in
// lockstep, round by round downwards: a lane whose walk stands in that round takes its step, the others (a diagonal step
// skips a round) do nothing.  The record of a round is then a compile-time register, the band row of the round does not
// depend on the walk (it is computed off the chain), and what is carried from round to round is -2 y and -round:
//     active = (rho - r - 1) >> 31;  tag = (record >> (2 (31 + rho - rights) - 2 y)) & 3 & active;  -2y += tag & 2;  -r += popcount(tag)
// 15 instructions per round, no branch, no memory access.  The moves leave as the raw tags (3 diag, 2 up, 1 left).
//
// TWO WAVEFRONTS per 64 walks (round 4): a LOADER that only moves records -- global memory to registers two windows ahead, registers
// to one of two LDS slots -- and a DECODER that only walks, one workgroup barrier per window between them.  With both jobs in
// one wavefront (round 3) the kernel took the SUM of its halves: 4.15 ms at 65536 walks, of which the decoding alone (records
// served from the cache) is 1.61 ms and the fetch alone, as a bare streaming kernel of the same shape, 2.75 ms
// (profiles/r04_sg_walk_parts.txt, profiles/r04_hbm_stream.txt) -- a lone wavefront on a SIMD that waits for memory computes
// nothing, and one that computes issues no loads.  Split, the loader's waits and the decoder's chain overlap -- PROVIDED the
// decoder issues no vector memory operation of its own inside the loop: its occasional loads (the band's move bits, once
// per block) and stores (a finished word of 32 moves, every two or three windows) queue up behind the loader's 16 KB of
// requests in the CU's memory pipeline, and the wavefront stalls at the ISSUE of each (3.94 ms with them, 3.03 ms without:
// profiles/r04_sg_walk_parts.txt, build 3).  So both go through the loader and LDS: the loader fetches the move bits of the
// block below and leaves them in LDS before the barrier at which the decoder steps down; the decoder puts every finished
// word of 32 moves into a ring of 32 words per walk in LDS, and the loader writes a walk's words out SIXTEEN at a time, one
// whole 128-byte line (a 64-byte piece of a line costs a read of the whole line on top of the write).  (Written one by one -- 8 bytes per walk, 8 KB apart, 64 separate memory transactions per store
// instruction, 67 M of them per 65536 walks -- the stores cost ~1 ms wherever they were issued: moved from the decoder to the
// loader as they were, the kernel went from 3.94 to 4.28 ms.)
// A/B builds of the walk only (make ab_walk, tools/experiments/sg_walk_parts.sh; never shipped): SG_WALK_EXP = 1 reads window 0
// every time (the records come out of the cache: what the decoding alone costs).
#ifndef SG_WALK_EXP
#define SG_WALK_EXP 0
#endif
__global__ void __launch_bounds__(128)
sg_walk_lane_kernel(uint32_t n, const uint32_t *__restrict__ codes, const uint32_t *__restrict__ dirs,
                    const int4 *__restrict__ summary, unsigned long long *__restrict__ moves,
                    int32_t *__restrict__ scores, uint32_t *__restrict__ lengths, uint32_t *__restrict__ window_stats)
{
    constexpr int WALKS = 64;
    constexpr int kRing = 32;                             // move words per walk in LDS: the loader flushes 16 as soon as it sees 16
    __shared__ uint4 line_codes[2][2][WALKS * kLinePitch];   // two slots of a BLOCK (32 rounds): [upper / lower window][walk][piece], padded
    __shared__ uint32_t dirs_lds[2][WALKS];               // move bits of block b for every walk, slot b & 1 (loader -> decoder)
    __shared__ unsigned long long ring_lds[WALKS][kRing + 1];   // finished move words of every walk, word i in slot i % kRing (row padded)
    __shared__ uint32_t count_lds[WALKS];                 // words finished so far per walk (decoder -> loader, once per trip)
    const int lane = threadIdx.x & 63;
    const bool loader = threadIdx.x >= 64;                // (wave-uniform) wavefront 1 loads, wavefront 0 walks
    const uint32_t a_first = blockIdx.x * WALKS;
    const uint32_t a0 = a_first + lane;
    const bool real = a0 < n;
    const uint32_t a = real ? a0 : n - 1;                 // tail lanes shadow the last alignment and store nothing
    const int4 sum = summary[a];
    // first window of the workgroup = the highest one any of its walks starts in; both wavefronts derive it the same way
    int wmax = sum.y / kCodeWindow;
    wmax = row16_max(wmax);
    wmax = max(max(__builtin_amdgcn_readlane(wmax, 0), __builtin_amdgcn_readlane(wmax, 16)),
               max(__builtin_amdgcn_readlane(wmax, 32), __builtin_amdgcn_readlane(wmax, 48)));
    // The windows are taken block by block of 32 rounds -- upper window 2 blk + 1, lower window 2 blk -- from w_top = wmax | 1
    // down to 0: a workgroup whose first window is a LOWER one starts one window higher; no walk stands in a round of that
    // window, so it passes without a step (its records are fetched but never looked at; past the last window of the buffer the
    // last one is fetched again).  Window w_top - i is processed in trip i and lives in LDS slot i & 1.
    const int w_top = wmax | 1;
    // One barrier per block of two windows: before it the loader has put the next block into the other slot and the decoder has
    // finished reading this trip's slot.  Only LDS traffic has to be visible across it (the fence names the local address space: a
    // plain workgroup fence would also wait for the loader's global loads, i.e. drain its prefetch).
    auto window_barrier = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };

    if (loader) {
        // The WALKS centre pieces of a window (one per walk, 64 bytes) lie side by side in memory: the wavefront fetches them
        // COOPERATIVELY, 1 KB per instruction -- piece p = i * 64 + lane of the block belongs to walk p / 4, quarter p % 4 -- and
        // hands them to the walks through LDS.  (A ragged last block repeats the batch's last piece.)
        const uint4 *all_codes = reinterpret_cast<const uint4 *>(codes);
        static_assert(kHalfQuads == 4, "four quarters per piece");
        auto piece_offset = [&](int i) -> uint32_t {      // uint4 offset of this lane's quarter i inside a window's pieces
            const uint32_t w_a = a_first + (uint32_t)((i * 64 + lane) >> 2);
            return (w_a < n ? w_a : n - 1) * kHalfQuads + (uint32_t)(lane & 3);
        };
        const uint32_t o0 = piece_offset(0), o1 = piece_offset(1), o2 = piece_offset(2), o3 = piece_offset(3);
        uint4 *my_slot = &line_codes[0][0][(lane >> 2) * kLinePitch + (lane & 3)];   // quarter i goes 16 walks (rows) further down
        constexpr int kSlot = WALKS * kLinePitch;                                    // uint4 per window in LDS (slot s, half h: (2 s + h) * kSlot)
        // (the four quarters of a window are eight plain variables and two macros: arrays or structs handed to lambdas by
        // reference stayed in scratch memory in round 3)
#define SG_LOAD_WINDOW(w, P)                                                                                          \
        do {                                                                                                          \
            const int w_ = SG_WALK_EXP == 1 ? 0 : (w) > 0 ? ((w) < kCodeWindows ? (w) : kCodeWindows - 1) : 0;       \
            const uint4 *base_ = all_codes + (size_t)w_ * n * kHalfQuads;                                             \
            P##0 = base_[o0]; P##1 = base_[o1]; P##2 = base_[o2]; P##3 = base_[o3];                                   \
        } while (0)
#define SG_TO_LDS(P, slot)                                                                                            \
        do {                                                                                                          \
            uint4 *dst_ = my_slot + (slot) * kSlot;                                                                   \
            dst_[0 * 16 * kLinePitch] = P##0; dst_[1 * 16 * kLinePitch] = P##1; dst_[2 * 16 * kLinePitch] = P##2;     \
            dst_[3 * 16 * kLinePitch] = P##3;                                                                         \
        } while (0)
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        uint4 au0 = z4, au1 = z4, au2 = z4, au3 = z4, al0 = z4, al1 = z4, al2 = z4, al3 = z4;      // a block's upper and lower window
        uint4 bu0 = z4, bu1 = z4, bu2 = z4, bu3 = z4, bl0 = z4, bl1 = z4, bl2 = z4, bl3 = z4;      // the block below it
        const uint32_t *my_dirs = dirs + a;               // word b of walk `lane` at my_dirs[b * n]: 256 contiguous bytes per wavefront
        unsigned long long *my_moves = moves + (size_t)a * kMoveWords;
        count_lds[lane] = 0u;
        uint32_t flushed = 0;                             // words of this lane's walk already in global memory (a multiple of 16)
        // what the decoder has finished since: out to global memory as soon as there are sixteen words, one aligned line
        auto drain = [&]() {
            const uint32_t have = count_lds[lane];        // (as of the last barrier)
            if (have - flushed >= 16u) {
                const unsigned long long *src = &ring_lds[lane][flushed % kRing];      // 0 or 16: sixteen consecutive slots
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                if (real) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const u64x2 v = {src[2 * q], src[2 * q + 1]};
                        *reinterpret_cast<u64x2 *>(my_moves + flushed + 2 * q) = v;
                    }
                }
                flushed += 16u;
            }
        };
        // A TRIP IS A BLOCK of two windows (one barrier per 32 rounds; with one per window the barriers were a third of the
        // kernel).  Block blk = windows 2 blk + 1 (upper) and 2 blk (lower) lives in LDS slot (top block - blk) & 1; while the
        // decoder walks it, the loader requests block blk - 2 and puts block blk - 1 (requested a trip ago) into the other slot.
        int blk = w_top / 2;
        SG_LOAD_WINDOW(2 * blk + 1, au); SG_LOAD_WINDOW(2 * blk, al);
        SG_LOAD_WINDOW(2 * blk - 1, bu); SG_LOAD_WINDOW(2 * blk - 2, bl);
        SG_TO_LDS(au, 0); SG_TO_LDS(al, 1);
        window_barrier();                                 // trip 0 may start: the top block is in slot 0
        for (;;) {                                        // (two trips per turn: the register sets and the slots swap roles)
            {
                const int nb = blk > 0 ? blk - 1 : 0;     // (block 0 again below block 0: never used)
                const unsigned d_next = my_dirs[(size_t)nb * n];
                SG_LOAD_WINDOW(2 * blk - 3, au); SG_LOAD_WINDOW(2 * blk - 4, al); drain();
                if (blk == 0) break;
                dirs_lds[nb & 1][lane] = d_next;          // before the barrier at which the decoder steps down into block nb
                SG_TO_LDS(bu, 2); SG_TO_LDS(bl, 3); window_barrier(); --blk;
            }
            {
                const int nb = blk > 0 ? blk - 1 : 0;
                const unsigned d_next = my_dirs[(size_t)nb * n];
                SG_LOAD_WINDOW(2 * blk - 3, bu); SG_LOAD_WINDOW(2 * blk - 4, bl); drain();
                if (blk == 0) break;
                dirs_lds[nb & 1][lane] = d_next;
                SG_TO_LDS(au, 0); SG_TO_LDS(al, 1); window_barrier(); --blk;
            }
        }
#undef SG_LOAD_WINDOW
#undef SG_TO_LDS
        window_barrier();                                 // the decoder has left its last words and the final count
        if (real) {
            const uint32_t have = count_lds[lane];        // whatever is left, word by word: fewer than 16 + 2 per walk
            for (uint32_t k = flushed; k < have; ++k) my_moves[k] = ring_lds[lane][k % kRing];
        }
        return;
    }

    // ---- the decoder ----
    const uint32_t *my_dirs = dirs + a;                   // word w at my_dirs[w * n]: the 64 walks read 256 contiguous bytes
    const int y = sum.w + 31 - sum.z;                     // .w = row of the band's top cell in the best round
    const int x = sum.y - y;                              // y + x = the round of the best cell
    int nr = -(y + x), ny2 = -2 * y;                      // minus the round of the walk's cell, minus twice its row
    uint32_t steps = 0;                                   // moves made
    unsigned long long acc = 0;                           // the last (steps & 31) moves, 2 bits each
    // Band row of a round r = r - (right moves up to and including r) = r - (rights_before + popcount of the block's move
    // bits up to r): no state carried from step to step, and a shorter dependency chain than an LDS lookup.
    // d_blk = move bits of the 32-round block the current window lies in; rights_before = right moves before that block,
    // known from the best round's band row when the wavefront reaches the block this walk starts in.
    int blk = wmax / 2;                                   // two windows per block of 32 rounds: 2 blk + 1 (upper), 2 blk (lower)
    const int start_blk = sum.y >> 5;
    unsigned d_blk = my_dirs[(size_t)blk * n];
    int rights_before = 0;
    auto enter_block = [&]() {
        if (blk == start_blk) rights_before = (sum.y - sum.w) - __popc(d_blk & ((2u << (sum.y & 31)) - 1u));
    };
    enter_block();
    const uint4 *my_outer = reinterpret_cast<const uint4 *>(codes) + half_quads(n) + (size_t)a * kHalfQuads;     // + window * n * kHalfQuads
    uint32_t windows_walked = 0, windows_again = 0;       // (scalars; summed over the launch into window_stats[2], [3])
    auto walk_window = [&](int w, const bool upper, const uint4 *my_line) {   // upper: window 2 blk + 1 of its block (a constant at every call)
        uint4 rec[kHalfQuads];                            // centre words of rounds 16 w + 4 j .. + 3 (.x .. .w)
#pragma unroll
        for (int j = 0; j < kHalfQuads; ++j) rec[j] = my_line[j];
        // the window's 16 move bits and the right moves before it
        const unsigned d_win = upper ? d_blk >> 16 : d_blk & 0xFFFFu;
        const int rb_win = rights_before + (upper ? (int)__popc(d_blk & 0xFFFFu) : 0);
        unsigned wm = 0;                                  // this window's moves, cnt2 / 2 of them
        int cnt2 = 0;
        const int nr_in = nr, ny2_in = ny2;
        // FIRST with the centre words alone: a step whose band cell is not one of 8 .. 23 reads nonsense and raises `astray`
        unsigned long long astray = 0;                    // (a scalar pair: one v_cmp + one s_or per round)
#pragma unroll
        for (int i = kCodeWindow - 1; i >= 0; --i) {
            if (i > 0 || w > 0) {                         // (round 0 is the cell (0, 0): nothing to decode)
                const int rho = kCodeWindow * w + i;
                const unsigned cw = i % 4 == 0 ? rec[i / 4].x : i % 4 == 1 ? rec[i / 4].y : i % 4 == 2 ? rec[i / 4].z : rec[i / 4].w;
                // band row of round rho = rho - (right moves up to and including rho); 2 * band cell of the walk = 2 (31 + top) - 2 y
                const int rights = rb_win + (int)__popc(d_win & ((2u << i) - 1u));
                const int shift = 2 * (31 + rho - rights) + ny2 - 16;         // inside the centre word: 0 .. 30
                const int active = keep_opaque(nr + (rho - 1)) >> 31;         // r == rho (r <= rho always): all ones
                astray |= __builtin_amdgcn_uicmp((unsigned)shift & (unsigned)active, 30u, 34 /* ICMP_UGT */);
                const unsigned tag = (cw >> (shift & 31)) & 3u & (unsigned)active;    // 3 diagonal, 2 up, 1 left (never 0 on a live path)
                ny2 += (int)(tag & 2u);                   // diagonal, up: one row back
                nr += (int)__popc(tag);                   // diagonal: two rounds back, up / left: one
                wm |= tag << cnt2;
                cnt2 -= 2 * active;
            }
        }
        ++windows_walked;
        if (astray != 0) {                                // (wavefront-uniform) some walk left the centre: the window again, whole records
            ++windows_again;
            nr = nr_in;
            ny2 = ny2_in;
            wm = 0;
            cnt2 = 0;
            const int w_ = SG_WALK_EXP == 1 ? 0 : w;
            uint4 out[kHalfQuads];                        // this walk's outer piece, straight from memory (the one place the decoder loads)
#pragma unroll
            for (int j = 0; j < kHalfQuads; ++j) out[j] = my_outer[(size_t)w_ * n * kHalfQuads + j];
#pragma unroll
            for (int i = kCodeWindow - 1; i >= 0; --i) {
                if (i > 0 || w > 0) {
                    const int rho = kCodeWindow * w + i;
                    const unsigned c = i % 4 == 0 ? rec[i / 4].x : i % 4 == 1 ? rec[i / 4].y : i % 4 == 2 ? rec[i / 4].z : rec[i / 4].w;
                    const unsigned o = i % 4 == 0 ? out[i / 4].x : i % 4 == 1 ? out[i / 4].y : i % 4 == 2 ? out[i / 4].z : out[i / 4].w;
                    // cells 0 .. 31 at bits 2k: outer low half, centre, outer high half
                    const unsigned long long cw = ((unsigned long long)((c >> 16) | (o & 0xFFFF0000u)) << 32) | ((o & 0xFFFFu) | (c << 16));
                    const int rights = rb_win + (int)__popc(d_win & ((2u << i) - 1u));
                    const int shift = 2 * (31 + rho - rights) + ny2;
                    const int active = keep_opaque(nr + (rho - 1)) >> 31;
                    const unsigned tag = (unsigned)(cw >> (shift & 63)) & 3u & (unsigned)active;
                    ny2 += (int)(tag & 2u);
                    nr += (int)__popc(tag);
                    wm |= tag << cnt2;
                    cnt2 -= 2 * active;
                }
            }
        }
        // the window's moves behind the ones collected so far; a full word of 32 leaves
        const int fill2 = 2 * (int)(steps & 31u);
        acc |= (unsigned long long)wm << fill2;
        if (fill2 + cnt2 >= 64) {                         // (then fill2 >= 32: the shift below is 2 .. 32)
            ring_lds[lane][(steps >> 5) % kRing] = acc;   // word steps / 32 is full: into the ring (the loader writes it out)
            acc = (unsigned long long)wm >> (64 - fill2);
        }
        steps += (uint32_t)(cnt2 >> 1);
        count_lds[lane] = steps >> 5;                     // full words so far (read by the loader after the next barrier)
    };
    auto step_down = [&]() {                              // leaving a block's lower window: the loader has left the block's move bits
        --blk;
        d_blk = dirs_lds[blk & 1][lane];
        rights_before -= (int)__popc(d_blk);
        enter_block();
    };
    const uint4 *up0 = &line_codes[0][0][lane * kLinePitch], *lo0 = &line_codes[0][1][lane * kLinePitch];
    const uint4 *up1 = &line_codes[1][0][lane * kLinePitch], *lo1 = &line_codes[1][1][lane * kLinePitch];
    window_barrier();                                     // the top block is in slot 0
    for (;;) {                                            // blk: the block in slot 0
        walk_window(2 * blk + 1, true, up0); walk_window(2 * blk, false, lo0); if (blk == 0) break; window_barrier();
        step_down();
        walk_window(2 * blk + 1, true, up1); walk_window(2 * blk, false, lo1); if (blk == 0) break; window_barrier();
        step_down();
    }
    if (steps & 31u) {                                    // the last, partial word goes the same way
        ring_lds[lane][(steps >> 5) % kRing] = acc;
        count_lds[lane] = (steps >> 5) + 1u;
    }
    window_barrier();                                     // the loader writes out what is left in the ring
    if (lane == 0) {
        atomicAdd(&window_stats[2], windows_walked);
        atomicAdd(&window_stats[3], windows_again);
    }
    if (real) {
        scores[a] = sum.x;
        lengths[a] = steps + 1;                           // positions = moves + 1
    }
}

// The walk's moves are a bit stream, 2 bits per move (3 diag, 2 up, 1 left), move t at bits 2t of the alignment's move words.
// Position i of the ascending list is the sum of the deltas of moves total-2 .. total-1-i, so a wavefront expands 512
// positions per trip: lane l takes the EIGHT moves of positions 8l .. 8l+7 of the chunk as one 16-bit field of the stream
// (y steps = c1, x steps = c0 of every 2-bit code: two masks), a 64-lane prefix sum of the lanes' popcounts places the
// lane, and position k of the lane is the lane's base + popcount of the top k+1 fields -- two v_bcnt with an accumulator per
// coordinate.  ~0.2 instructions per position (round 2: one position per lane and prefix sum, ~0.75: the kernel was bound by
// instruction issue, 3.4 TB/s of stores).  The lanes' 64-byte results cross LDS so that every store instruction writes
// one contiguous, line-aligned kilobyte.
constexpr int kExpPitch = 5;                              // uint4 per lane row in LDS (4 used): the 80-byte pitch keeps 16-byte accesses conflict-free

__global__ void __launch_bounds__(256)
sg_expand_kernel(uint32_t n, const unsigned long long *__restrict__ moves, const uint32_t *__restrict__ lengths,
                 int32_t *__restrict__ tracebacks, uint32_t cap)
{
    __shared__ uint4 stage[4][64 * kExpPitch];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t a = blockIdx.x * 4 + (threadIdx.x >> 6);                  // one wavefront per alignment
    if (a >= n) return;                                   // wave-uniform; only wave-level synchronisation below
    const uint32_t *stream = reinterpret_cast<const uint32_t *>(moves + (size_t)a * kMoveWords);
    int2 *out = reinterpret_cast<int2 *>(tracebacks) + (size_t)a * cap;
    const int total = (int)lengths[a];                    // positions; moves 0 .. total-2 in walking order
    const int limit = total < (int)cap ? total : (int)cap;                    // positions to write
    // every 1 KB store of the wavefront starts on a 128-byte line: the first chunk begins `skew` positions before out[0]
    const int skew = (int)((reinterpret_cast<uintptr_t>(out) >> 3) & 15u);
    uint4 *my_row = &stage[wv][lane * kExpPitch];
    const uint4 *my_pieces = &stage[wv][(lane >> 2) * kExpPitch + (lane & 3)];            // + j * 16 rows for store j
    unsigned carry = 0;                                   // y | x << 16 of the last position before the chunk
    // the 64 bits of the move stream that hold the 16 this lane wants of the chunk at `base`
    auto fetch = [&](int base) -> unsigned long long {
        const int s = total - 1 - (base + 8 * lane) - 7;
        const int sc = s < 0 ? 0 : s;
        int dw = (2 * sc) >> 5;
        dw = dw < 2 * kMoveWords - 2 ? dw : 2 * kMoveWords - 2;              // the 8-byte fetch stays inside the move words
        return stream[dw] | ((unsigned long long)stream[dw + 1] << 32);
    };
    // The next chunk's moves are requested BEFORE this chunk's positions are stored: loads and stores retire in order, so a
    // fetch issued behind the four 1 KB stores of a trip made the next trip wait for those stores as well.
    unsigned long long w_next = fetch(-skew);
    for (int base = -skew; base < limit; base += 512) {
        const int s = total - 1 - (base + 8 * lane) - 7;  // position base + 8 lane + k takes move s + 7 - k
        // the 16 bits of the stream that start at move s (s < 0 or past the last move: those fields are masked off below)
        const int sc = s < 0 ? 0 : s;
        const unsigned long long w = w_next;
        w_next = fetch(base + 512);                       // (past the list: an address inside the move words, value unused)
        const int up = -2 * s < 16 ? -2 * s : 16;
        unsigned f = s >= 0 ? (unsigned)(w >> ((2 * sc) & 31)) : (unsigned)w << up;
        const int lo = -s < 0 ? 0 : (-s > 8 ? 8 : -s), hi = total - 1 - s < 0 ? 0 : (total - 1 - s > 8 ? 8 : total - 1 - s);
        f &= ((1u << (2 * hi)) - 1u) & ~((1u << (2 * lo)) - 1u);             // fields lo .. hi-1 hold moves 0 .. total-2
        const unsigned yb = (f >> 1) & 0x5555u, xb = f & 0x5555u;             // a row step / a column step per move
        // inclusive prefix sum of the lanes' totals (y in the low half, x in the high half: both stay below 2^15)
        const unsigned own = (unsigned)__popc(yb) | ((unsigned)__popc(xb) << 16);
        unsigned v = own;
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, true);
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, true);
        v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, true);
        const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 15), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 31),
                       r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 47), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
        const int row = lane >> 4;
        v += carry + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u) - own;   // = the position before this lane's first
        const unsigned by = v & 0xFFFFu, bx = v >> 16;
        carry += r0 + r1 + r2 + r3;
        // position k of the lane = base + the steps of fields 7 .. 7-k
        auto py = [&](int k) { return (int)(by + (unsigned)__popc(yb & ((0xFFFFu << (2 * (7 - k))) & 0xFFFFu))); };
        auto px = [&](int k) { return (int)(bx + (unsigned)__popc(xb & ((0xFFFFu << (2 * (7 - k))) & 0xFFFFu))); };
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // (the previous trip's reads of the stage are done)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) my_row[q] = make_uint4((unsigned)py(2 * q), (unsigned)px(2 * q), (unsigned)py(2 * q + 1), (unsigned)px(2 * q + 1));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        if (base >= 0 && base + 512 <= limit) {           // (wave-uniform) a chunk inside the list: four unconditional 1 KB stores
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // store j: piece g = 64 j + lane = positions base + 2g, base + 2g + 1
                const uint4 pc = my_pieces[j * 16 * kExpPitch];
                const v4u both = {pc.x, pc.y, pc.z, pc.w};
                __builtin_nontemporal_store(both, reinterpret_cast<v4u *>(out + base + 2 * (64 * j + lane)));   // streamed: never read again on the GPU
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 pc = my_pieces[j * 16 * kExpPitch];
                const int e = base + 2 * (64 * j + lane);
                if (e >= 0 && e + 1 < limit) {
                    const v4u both = {pc.x, pc.y, pc.z, pc.w};
                    __builtin_nontemporal_store(both, reinterpret_cast<v4u *>(out + e));
                } else {
                    if (e >= 0 && e < limit) out[e] = make_int2((int)pc.x, (int)pc.y);
                    if (e + 1 >= 0 && e + 1 < limit) out[e + 1] = make_int2((int)pc.z, (int)pc.w);
                }
            }
        }
    }
}

}  // namespace

namespace {
inline size_t round16(size_t v) { return (v + 127) & ~size_t(127); }   // (every part of the workspace starts on a 128-byte line)
inline size_t codes_bytes(size_t n) { return round16(n * (size_t)kCodeWindows * kCodeWindow * sizeof(uint2)); }
inline size_t dirs_bytes(size_t n) { return round16(n * (size_t)kDirWords * sizeof(uint32_t)); }
}  // namespace

inline size_t streams_bytes(size_t n) { return round16(((n + 63) / 64 * 64) * 2 * (size_t)kStreamWords * sizeof(unsigned long long)); }   // whole blocks of 64 / 32 / 16
inline size_t moves_bytes(size_t n) { return round16(n * (size_t)kMoveWords * sizeof(unsigned long long)); }

constexpr size_t kStatsBytes = 128;                       // window counters of the sweep, at the start of the workspace

size_t semiglobal_workspace_bytes(size_t n)
{
    return kStatsBytes + codes_bytes(n) + dirs_bytes(n) + round16(n * sizeof(int4)) + streams_bytes(n) + moves_bytes(n);
}

namespace {
// Which sweep a batch of n alignments runs: 10 * G + W = band over G lanes (G = 4, 2: sg_forward_split_kernel, 8 / 16 cells per
// lane, 16 / 32 alignments per wavefront; G = 1: sg_forward_lane_kernel, 64 per wavefront), compiled for W resident
// wavefronts per SIMD.  swmi_semiglobal_set_mapping / SWMI_SG_SWEEP force one: G or 10 * G + W.  (Rounds 1-2 had a third
// mapping, one band cell per lane: the packed split sweep beats it down to a batch of ONE alignment, 9.5 against 10.2 ms,
// profiles/r03_sg_small_batches.txt, and it is gone -- with it the second record format.)  The kernels
// are compiled once per scheduling target W (amdgpu_waves_per_eu): hipcc orders the round for W resident wavefronts per
// SIMD, and a build runs W wavefronts per SIMD at a time -- a batch that gives a SIMD fewer leaves SIMDs idle (the
// dispatcher fills a SIMD to W before it moves on), one that gives it more runs in turns.
//
// Cost model, measured on 256 CUs on SpeedtestSemiGlobal inputs, i.e. calm windows (profiles/r04_sg_kernel_matrix.txt, sweep phase in ms): a build runs
// floor(w / W) full turns + one partial turn, w = wavefronts per SIMD the batch yields with that G.
struct SweepBuild {
    int id, lanes, waves;        // 10 G + W, G, W
    float full;                  // one turn of W wavefronts per SIMD
    float part[3];               // a last turn of 1 .. W - 1 wavefronts per SIMD
};
constexpr SweepBuild kSweepBuilds[] = {
    {41, 4, 1, 7.05f, {0, 0, 0}},
    {21, 2, 1, 9.75f, {0, 0, 0}},
    {22, 2, 2, 17.0f, {7.8f, 0, 0}},             // (a batch of ONE wavefront per SIMD on this build: 16.5 -- 21 is there for it)
    {11, 1, 1, 14.6f, {0, 0, 0}},
    {12, 1, 2, 25.9f, {12.0f, 0, 0}},            // (the same: 25.2 -- 11 is there)
};
int choose_sweep(size_t n, int compute_units, const SgTuning &tuning)
{
    if (tuning.force_sweep >= 0) {
        const int s = tuning.force_sweep;
        return s == 4 ? 44 : s == 2 ? 22 : s == 1 ? 12 : s;
    }
    const size_t simds = (size_t)(compute_units > 0 ? compute_units : 256) * 4;     // (a partitioned gfx950 reports fewer CUs)
    int best = 41;
    float best_t = 0;
    for (const SweepBuild &b : kSweepBuilds) {
        const size_t wavefronts = (n * b.lanes + 63) / 64;
        const int w = (int)((wavefronts + simds - 1) / simds);
        const int turns = w / b.waves, rest = w % b.waves;
        float t = turns * b.full + (rest ? b.part[rest - 1] : 0.0f);
        if (b.waves == 2 && w == 1) t = b.id == 12 ? 25.2f : 16.5f;
        if (best_t == 0 || t < best_t) {
            best_t = t;
            best = b.id;
        }
    }
    return best;
}
inline int sweep_lanes(int sweep) { return sweep / 10; }
}  // namespace

void semiglobal_kernel_names(size_t n, int compute_units, char *sweep_name, size_t sweep_len, char *tb_name, size_t tb_len,
                             SgTuning tuning)
{
    const int sweep = choose_sweep(n, compute_units, tuning);
    if (sweep_name && sweep_len) {
        if (sweep_lanes(sweep) == 1) snprintf(sweep_name, sweep_len, "sg_forward_lane_kernel<%d>", sweep % 10);
        else snprintf(sweep_name, sweep_len, "sg_forward_split_kernel<%d, %d>", sweep_lanes(sweep), sweep % 10);
    }
    if (tb_name && tb_len)
        snprintf(tb_name, tb_len, "sg_walk_lane_kernel + sg_expand_kernel");
}

size_t semiglobal_move_words() { return (size_t)kMoveWords; }

hipError_t launch_semiglobal(const uint8_t *d_seq1s, const uint8_t *d_seq2s, size_t n, void *d_workspace,
                             int32_t *d_scores, int32_t *d_tracebacks, size_t cap, uint32_t *d_lengths, hipStream_t stream,
                             hipEvent_t between, int compute_units, SgTuning tuning, unsigned long long *d_moves_out)
{
    if (n == 0) return hipSuccess;
    // the first line of the workspace: windows of the last launch's sweep wavefronts, [0] all, [1] calm; of its walk wavefronts,
    // [2] all, [3] decoded a second time with the records' outer halves (4096 wavefronts x 4096 windows stay below 2^32)
    uint32_t *window_stats = static_cast<uint32_t *>(d_workspace);
    {
        const hipError_t ez = hipMemsetAsync(window_stats, 0, kStatsBytes, stream);
        if (ez != hipSuccess) return ez;
    }
    char *ws = static_cast<char *>(d_workspace) + kStatsBytes;
    uint32_t *codes = reinterpret_cast<uint32_t *>(ws);
    uint32_t *top = reinterpret_cast<uint32_t *>(ws + codes_bytes(n));          // the band's move bits (kDirWords x n)
    int4 *summary = reinterpret_cast<int4 *>(ws + codes_bytes(n) + dirs_bytes(n));
    unsigned long long *streams = reinterpret_cast<unsigned long long *>(ws + codes_bytes(n) + dirs_bytes(n) + round16(n * sizeof(int4)));
    // the walk's moves: into the workspace, or straight into the caller's buffer (the entries that return moves, not positions)
    unsigned long long *moves = d_moves_out ? d_moves_out
                                            : reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(streams) + streams_bytes(n));
    // (Cutting the batch into sub-batches so that traceback k overlaps sweep k+1 was tried and is slower: below ~16k
    // alignments the sweep is latency bound, and four short sweeps in sequence cost four times one.)
    const int sweep = choose_sweep(n, compute_units, tuning);
    {
        const uint32_t per_block = 64 / sweep_lanes(sweep);                 // alignments per sweep wavefront: 64 / G
        const size_t tiles = ((n + per_block - 1) / per_block) * (size_t)(kStreamWords / kPackWords);        // one wavefront each
        hipLaunchKernelGGL(sg_pack_streams_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, d_seq1s, d_seq2s,
                           (uint32_t)n, streams, per_block);
        const dim3 grid4((unsigned)((n + 15) / 16)), grid2((unsigned)((n + 31) / 32)), grid1((unsigned)((n + 63) / 64));
#define SWMI_SG_LAUNCH1(W) \
    hipLaunchKernelGGL((sg_forward_lane_kernel<W>), grid1, dim3(64), 0, stream, streams, (uint32_t)n, codes, top, summary, tuning.exact_only, window_stats)
#define SWMI_SG_LAUNCH(G, W, GRID) \
    hipLaunchKernelGGL((sg_forward_split_kernel<G, W>), GRID, dim3(64), 0, stream, streams, (uint32_t)n, codes, top, summary, tuning.exact_only, window_stats)
        switch (sweep) {
        case 41: SWMI_SG_LAUNCH(4, 1, grid4); break;
        case 42: SWMI_SG_LAUNCH(4, 2, grid4); break;
        case 43: SWMI_SG_LAUNCH(4, 3, grid4); break;
        case 44: SWMI_SG_LAUNCH(4, 4, grid4); break;
        case 21: SWMI_SG_LAUNCH(2, 1, grid2); break;
        case 22: SWMI_SG_LAUNCH(2, 2, grid2); break;
        case 23: SWMI_SG_LAUNCH(2, 3, grid2); break;
        case 11: SWMI_SG_LAUNCH1(1); break;
        case 12: SWMI_SG_LAUNCH1(2); break;
        default: return hipErrorInvalidValue;
        }
#undef SWMI_SG_LAUNCH
#undef SWMI_SG_LAUNCH1
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && between) e = hipEventRecord(between, stream);      // phase timing (swmi_semiglobal_time_device)
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sg_walk_lane_kernel, dim3((unsigned)((n + 63) / 64)), dim3(128), 0, stream, (uint32_t)n, codes, top, summary,
                       moves, d_scores, d_lengths, window_stats);
    if (d_tracebacks || !d_moves_out)           // (moves only: the caller expands them itself, swmi_semiglobal_expand_moves)
        hipLaunchKernelGGL(sg_expand_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, stream, (uint32_t)n, moves, d_lengths,
                           d_tracebacks, (uint32_t)cap);
    return hipGetLastError();
}

}  // namespace swmi
