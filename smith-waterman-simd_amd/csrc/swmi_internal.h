// swmi_internal.h -- declarations shared by the HIP kernels (sw_kernels.hip) and the C-ABI host side (swmi_api.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace swmi {

// The 4x4 int8 score matrix as four dwords: rows.r[a] holds sm[a*4 + 0..3] in bytes 0..3
// (index order of source.cpp:50: seq1 base selects the row, seq2 base the column).
struct SmRows {
    uint32_t r[4];
};

enum ScheduleFlags : unsigned {
    kNoGapFold = 1u,   // never use the gap-folded recurrence
    kUseLut = 4u,      // LDS score-lookup kernel (sw128_lut_kernel)
    kUseI16 = 2u,      // compiler-scheduled 16-bit max variant (v_max_i16 is full rate, but see DESIGN.md section 5)
    kNoPacked = 8u,    // never use the packed kernel (sw128_pk_kernel, what L = 4, 8, 16 run otherwise): A/B against the int32 cell
};

struct LaunchConfig {
    int lanes_per_alignment;   // 64, 32, 16, 8, 4, 2
    bool fold_gap;             // rows carry sm + gap (requires every sm + gap to fit int8)
    bool use_i16;
    unsigned extra_lds_bytes;  // unused dynamic LDS per workgroup: caps workgroups per CU (occupancy sweep, SWMI_EXTRA_LDS)
    bool use_lut;              // LDS score lookup instead of v_dot4 (gap-folded body only, L in {16, 8, 4})
    bool use_pk;               // L = 4, 8 or 16, no other variant asked for: packed kernel, two alignments per register;
    int pk_bias;               //   rows then hold s + gap + pk_bias (bytes 0..255), pk_bias = max(0, -(min s + gap))
    int pk_variant;            //   cell body: 0 = no bias needed, 1 = biased, 2 = vertical-offset form (rows hold s + 2 gap);
                               //   sw_kernels.hip PkVariant
};

// Score n pairs resident in device memory. packed = 2-bit inputs (32 B per sequence).
hipError_t launch_score(const LaunchConfig &cfg, const uint8_t *d_seq1s, const uint8_t *d_seq2s, int32_t *d_scores,
                        size_t n, const SmRows &rows, int gap, bool packed, hipStream_t stream);
// Score n_seq1 sequences against one seq2 (device pointers; d_seq2 = 128 bytes).
hipError_t launch_score_one_vs_many(const LaunchConfig &cfg, const uint8_t *d_seq1s, const uint8_t *d_seq2,
                                    int32_t *d_scores, size_t n_seq1, const SmRows &rows, int gap, hipStream_t stream);
// Banded (128 diagonals) affine-gap local alignment of n pairs of `len`-mers (device pointers).
hipError_t launch_banded_affine(const uint8_t *d_seq1s, const uint8_t *d_seq2s, int32_t *d_scores, size_t n, int len,
                                const SmRows &rows, int gap_open, int gap_ext, hipStream_t stream,
                                bool allow_i16 = true,      // false: never the 16-bit-max build (SWMI_BANDED_NO_I16, A/B)
                                bool allow_pk = true);      // false: never the packed kernel (SWMI_BANDED_NO_PK, A/B)
// 2 = sw_banded_affine_pk_kernel (two alignments per wavefront), 1 = the int32 cell with 16-bit maxes, 0 = the int32 cell
int banded_affine_kernel_choice(int len, const SmRows &rows, int gap_open, int gap_ext, bool allow_i16, bool allow_pk);
hipError_t launch_generate(uint8_t *d_seq1s, uint8_t *d_seq2s, size_t n, uint64_t seed, uint64_t first_pair,
                           hipStream_t stream);
// Exhaustive check that v_pk_maximum3_f16 is a packed integer max on [0, 0x7C00)^2 (d_counts: two zeroed 64-bit words:
// comparisons made, comparisons that failed).
hipError_t launch_pk_max3_selftest(unsigned long long *d_counts, hipStream_t stream);
hipError_t launch_unpack(const uint8_t *d_packed, uint8_t *d_unpacked, size_t n_seqs, hipStream_t stream);

bool schedule_supported(int lanes_per_alignment);

// Semi-global adaptive-band X-drop aligner (sg_kernels.hip). Workspace: codes + band rows + summaries for n alignments.
size_t semiglobal_workspace_bytes(size_t n);
// Override of the sweep mapping (swmi_semiglobal_set_mapping; SWMI_SG_SWEEP gives the initial value at swmi_init): -1 = automatic
struct SgTuning {
    int force_sweep = -1;        // G or 10 * G + W (sg_kernels.hip choose_sweep)
    int exact_only = 0;          // 1: no calm windows -- every round runs the X-drop test (A/B and tests; same results either way)
};
hipError_t launch_semiglobal(const uint8_t *d_seq1s, const uint8_t *d_seq2s, size_t n, void *d_workspace,
                             int32_t *d_scores, int32_t *d_tracebacks, size_t cap, uint32_t *d_lengths, hipStream_t stream,
                             hipEvent_t between = nullptr,    // recorded between the sweep and the traceback kernel
                             int compute_units = 256,         // of the device: picks the sweep mapping (wavefronts per SIMD)
                             SgTuning tuning = SgTuning(),
                             unsigned long long *d_moves_out = nullptr);   // non-NULL: the walk's 2-bit moves go here
                                                                           //   ([n][semiglobal_move_words()]); with d_tracebacks
                                                                           //   NULL the expand kernel is skipped
size_t semiglobal_move_words();                  // 64-bit words of moves per alignment (32 moves each)
// Names of the sweep / traceback kernels launch_semiglobal picks for n alignments on a device with that many CUs.
void semiglobal_kernel_names(size_t n, int compute_units, char *sweep_name, size_t sweep_len, char *tb_name, size_t tb_len,
                             SgTuning tuning = SgTuning());

}  // namespace swmi
