/*
 * swmi.h -- C ABI of the MI355X-native batched Smith-Waterman scorer (libswmi.so).
 *
 * Drop-in boundary for ONE path of eukaryo/smith-waterman-simd: the fixed-shape
 * (128 x 128, 2-bit alphabet held one base per byte, 4x4 int8 score matrix, linear gap,
 * score only) Smith-Waterman scorer.  The reference has no FFI layer; its boundary is
 * the free-function signature
 *
 *     int SmithWaterman_simdN(const std::array<uint8_t,128>& seq1,
 *                             const std::array<uint8_t,128>& seq2,
 *                             const std::array<int8_t,16>&  score_matrix,
 *                             const int8_t gap_penalty);
 *
 * (source.cpp:35-39 scalar, :462-466 simd4, :758-762 simd7, :953-957 simd9).  Every entry
 * point below cites the reference interface it replaces.  Semantics are those of the
 * scalar SmithWaterman (source.cpp:49-53) in int32:
 *
 *     H(i,j) = max(0, H(i-1,j-1) + sm[seq1[i-1]*4 + seq2[j-1]], H(i-1,j) - gap, H(i,j-1) - gap)
 *     score  = max over the 128 x 128 cells
 *
 * Valid domain (bit-exact with the reference scalar, and with simd..simd8 wherever those
 * are themselves valid): every score_matrix entry in [-128,127], gap_penalty in [0,127].
 * Bases are taken modulo 4 (the reference indexes out of range for bases >= 4).
 * gap_penalty < 0 is rejected with SWMI_ERR_DOMAIN.
 *
 * There is NO CPU fallback: every scoring entry point runs hand-written gfx950 HIP kernels
 * and fails with an error code when no MI355X-class (gfx950) device is usable.
 *
 * Plain C: pointers and sizes only, no C++/torch types.
 */
#ifndef SWMI_H
#define SWMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SWMI_API __attribute__((visibility("default")))
#else
#define SWMI_API
#endif

#define SWMI_SEQ_LEN 128        /* std::array<uint8_t,128>, source.cpp:463-464 */
#define SWMI_PACKED_LEN 32      /* std::array<uint8_t,32>,  source.cpp:1580     */
#define SWMI_VERSION 300

enum swmi_status {
    SWMI_OK = 0,
    SWMI_ERR_NOT_INITIALIZED = -1,
    SWMI_ERR_NO_DEVICE = -2,         /* no HIP device / HIP runtime unusable            */
    SWMI_ERR_UNSUPPORTED_ARCH = -3,  /* device is not gfx950                             */
    SWMI_ERR_INVALID_ARGUMENT = -4,  /* NULL pointer, bad size, unknown schedule ...     */
    SWMI_ERR_DOMAIN = -5,            /* gap_penalty < 0                                  */
    SWMI_ERR_ALIGNMENT = -6,         /* device pointer not 16-byte aligned               */
    SWMI_ERR_HIP = -7,               /* a HIP call failed; text in swmi_last_error()     */
    SWMI_ERR_QUEUE_FULL = -8
};

/* ---- lifetime -----------------------------------------------------------------------
 *
 * The library keeps one CONTEXT per bound GPU (streams, staging buffers, workspaces).  Two ways to run:
 *   one process per GPU   swmi_init(device): one context; what bench.py's ranks and every single-GPU caller use
 *   one process, G GPUs   swmi_init_all(G) / swmi_init_devices(list): G contexts, index 0..G-1; the *_multi and
 *                         swmi_sharded_* entry points below split a batch over them
 * Every single-GPU entry point addresses the context the calling THREAD has selected with swmi_use_gpu(index)
 * (default: index 0) and makes that context's device current (hipSetDevice) before it touches HIP -- the model of
 * hipSetDevice itself.  The reference has no counterpart (a single-threaded CPU program, source.cpp:3275-3301). */

/* Bind the process to one GPU (device = ordinal as seen by HIP, or -1 for "LOCAL_RANK env var if set, else 0").
 * Idempotent for the same device; another device needs swmi_shutdown() first. */
SWMI_API int swmi_init(int device);
/* Bind the first n_gpus visible devices (n_gpus <= 0: all of them).  Returns the number of contexts (> 0) or a negative
 * swmi_status; every device must be gfx950.  Enables peer access between the bound devices where the platform allows. */
SWMI_API int swmi_init_all(int n_gpus);
/* Bind an explicit list.  A device may appear more than once -- two contexts on one GPU behave like two GPUs that share
 * the hardware (how the multi-GPU path is rehearsed on a one-GPU box). */
SWMI_API int swmi_init_devices(const int *devices, int n);
SWMI_API int swmi_num_gpus(void);                 /* number of contexts (0 before init) */
SWMI_API int swmi_use_gpu(int index);             /* select the context this thread's single-GPU calls address */
/* Releases every context (streams, staging buffers, workspaces).  Must not race with other calls into the library.
 * Handles created earlier (swmi_queue, swmi_sharded_batch) stay valid OBJECTS: every call on one returns
 * SWMI_ERR_NOT_INITIALIZED from now on -- also after a new swmi_init* -- and its *_destroy call still releases what the
 * handle owns.  Destroy handles before shutting down where you can; you do not have to (a garbage-collected binding cannot
 * promise the order). */
SWMI_API int swmi_shutdown(void);
/* Text of the last error on the calling thread ("" if none). Never NULL. */
SWMI_API const char *swmi_last_error(void);
SWMI_API int swmi_version(void);

/* ---- scoring ------------------------------------------------------------------------ */

/* Replaces a call to SmithWaterman / SmithWaterman_simd .. _simd9
 * (source.cpp:35-39, :462-466, :758-762, :953-957; call sites :2961-2970, :3077, :3212).
 * std::array<>::data() passes straight through.  Synchronous: one launch per call, so it
 * is correct but launch-latency bound -- use the batch or queue entry points for
 * throughput.  Returns the score (>= 0) or a negative swmi_status. */
SWMI_API int swmi_score_pair(const uint8_t seq1[SWMI_SEQ_LEN], const uint8_t seq2[SWMI_SEQ_LEN],
                             const int8_t score_matrix[16], int8_t gap_penalty);

/* The reference's 1M-call loop (source.cpp:3074-3082: `for 1,000,000: score = simd4(a,b,sm,gap)`)
 * as ONE call: pair k is the 128 bytes at seq1s + 128*k and seq2s + 128*k (the per-pair
 * layout of std::array<uint8_t,128>, concatenated).  Host buffers (pageable or pinned).  The PCIe link bounds this entry
 * (256 B per pair in, against ~1 ns of kernel time per pair), so the batch goes through the GPU in GRANULES on a few device
 * buffer sets with a stream each: granule k's kernel runs while granule k+1 is being copied in, and the granules TAPER by the
 * ratio of kernel time to copy time per pair, so that no kernel is still running when the next granule has landed and
 * almost nothing is left to compute when the last copy ends: at 256 B per pair each granule is three quarters of what is
 * left (1M pairs at most, 16K at least: a 1M-pair batch goes as 768K, 192K, 48K, 16K); the one-vs-many entry (128 B per
 * pair) halves; the 2-bit packed entry (64 B per pair), whose copy and kernel take the same time, uses near-equal granules
 * (DESIGN.md section 6).  The scores come back in ONE copy per 16M pairs after the last kernel -- a copy into pageable
 * memory blocks the caller until the stream reaches it, so copying scores back behind
 * every granule (round 2) serialised copy and kernel.  swmi_host_granules_for() reports the schedule.  The copies are issued
 * straight from the caller's memory (the HIP runtime stages pageable pages itself; an extra copy into library-owned pinned
 * memory measured slower); batches of up to 64 pairs go through a pinned, device-visible buffer
 * instead (no copy commands at all).  Thread-safe: calls on one context serialise.
 * scores[k] receives what SmithWaterman(seq1_k, seq2_k, score_matrix, gap) returns.
 * n may be 0.  Returns SWMI_OK or a negative swmi_status. */
SWMI_API int swmi_score_batch(const uint8_t *seq1s, const uint8_t *seq2s, size_t n,
                              const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores);

/* The granules a host batch of n pairs is cut into (the pipeline above), in order; returns how many there are and writes
 * the first `cap` sizes to granules (NULL to count).  Needs no device. */
SWMI_API size_t swmi_host_granules(size_t n, size_t *granules, size_t cap);

/* The same for any of the three host-batch entries: SWMI_ENTRY_PAIRS = swmi_score_batch (256 B per pair over the link;
 * what swmi_host_granules reports), SWMI_ENTRY_PACKED = swmi_score_batch_packed (64 B), SWMI_ENTRY_ONE_VS_MANY =
 * swmi_score_one_vs_many (128 B).  Returns 0 for an unknown entry.  Needs no device. */
#define SWMI_ENTRY_PAIRS 0
#define SWMI_ENTRY_PACKED 1
#define SWMI_ENTRY_ONE_VS_MANY 2
SWMI_API size_t swmi_host_granules_for(size_t n, int entry, size_t *granules, size_t cap);

/* Same contract with all three buffers already resident in device memory (16-byte aligned
 * device pointers; `stream` is a hipStream_t, NULL meaning the HIP null stream as usual).
 * Asynchronous: returns after the launch; the caller synchronises the stream.  This is the
 * entry bench.py times (inputs resident in HBM). */
SWMI_API int swmi_score_batch_device(const void *d_seq1s, const void *d_seq2s, size_t n,
                                     const int8_t score_matrix[16], int8_t gap_penalty,
                                     void *d_scores, void *stream);

/* One-vs-many shape of SmithWaterman_8b111x32mark1/2/3 (source.cpp:1227-1230: 32 seq1 x one
 * seq2 -> int[32]) generalised to n_seq1 sequences and arbitrary parameters:
 * scores[k] = SmithWaterman(seq1s + 128*k, seq2, sm, gap).  Host buffers. */
SWMI_API int swmi_score_one_vs_many(const uint8_t *seq1s, size_t n_seq1, const uint8_t seq2[SWMI_SEQ_LEN],
                                    const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores);

/* Same, device-resident (d_seq1s: n_seq1 x 128 bytes, d_seq2: 128 bytes, 16-byte aligned), asynchronous on `stream`. */
SWMI_API int swmi_score_one_vs_many_device(const void *d_seq1s, size_t n_seq1, const void *d_seq2,
                                           const int8_t score_matrix[16], int8_t gap_penalty,
                                           void *d_scores, void *stream);

/* 2-bit packed inputs in the reference's own wire format (unpack(), source.cpp:1580-1583:
 * base k of byte i = (src[i] >> 2k) & 3): pair k is the 32 bytes at seq1s_packed + 32*k.
 * The kernel unpacks on the fly (no unpacked copy in HBM).  Host buffers.  64 B per pair over the link: copy and kernel
 * take the same time, so this entry runs near-equal granules and issues them from TWO host threads -- the caller's and a
 * persistent helper the context creates on first use -- so that one copy command's DMA runs while the other thread
 * prepares the next (SWMI_HOST_THREADS=1 in the environment keeps everything on the calling thread). */
SWMI_API int swmi_score_batch_packed(const uint8_t *seq1s_packed, const uint8_t *seq2s_packed, size_t n,
                                     const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores);
SWMI_API int swmi_score_batch_packed_device(const void *d_seq1s_packed, const void *d_seq2s_packed, size_t n,
                                            const int8_t score_matrix[16], int8_t gap_penalty,
                                            void *d_scores, void *stream);

/* ---- multi-GPU: one batch over the G bound GPUs -----------------------------------------
 * The reference's 1M-call loop (source.cpp:3074-3082) pointed at G GPUs.  Pairs are independent, so shard g of G is the
 * contiguous range swmi_shard_bounds(n, g, G) (shards differ by at most one pair) and the only exchange step is the
 * final gather of the int32 scores.  No input byte ever crosses GPUs. */
SWMI_API int swmi_shard_bounds(size_t n, int shard, int n_shards, size_t *lo, size_t *hi);   /* needs no device */

/* swmi_score_batch / swmi_score_batch_packed over every bound GPU: one host thread and one stream set per GPU, each
 * copying its shard in, scoring it and copying its scores straight into the caller's slice scores[lo..hi). */
SWMI_API int swmi_score_batch_multi(const uint8_t *seq1s, const uint8_t *seq2s, size_t n,
                                    const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores);
SWMI_API int swmi_score_batch_packed_multi(const uint8_t *seq1s_packed, const uint8_t *seq2s_packed, size_t n,
                                           const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores);

/* A batch whose shards stay RESIDENT on the GPUs (what a multi-GPU caller times): shard g of the inputs and of the scores
 * lives in GPU g's HBM; swmi_sharded_score launches every GPU's kernel on that GPU's own stream and then runs the gather:
 *   SWMI_GATHER_NONE   scores stay sharded (read them with swmi_sharded_scores_host)
 *   SWMI_GATHER_ROOT   every GPU pushes its shard into the full int32[n] vector on GPU 0 (peer-to-peer DMA over xGMI)
 *   SWMI_GATHER_ALL    every GPU ends up with the full vector: RCCL (librccl, loaded on first use; ncclAllGather for
 *                      equal shards, grouped ncclBroadcast for ragged ones); peer copies when a GPU is bound twice
 * All calls are asynchronous until swmi_sharded_wait. */
typedef struct swmi_sharded_batch swmi_sharded_batch;
enum swmi_gather { SWMI_GATHER_NONE = 0, SWMI_GATHER_ROOT = 1, SWMI_GATHER_ALL = 2 };
SWMI_API int swmi_sharded_create(size_t n, int packed, swmi_sharded_batch **out);
SWMI_API int swmi_sharded_destroy(swmi_sharded_batch *b);
/* inputs: generated on each GPU from (seed, first_pair + global pair index), or copied from host arrays of n pairs */
SWMI_API int swmi_sharded_generate(swmi_sharded_batch *b, uint64_t seed, uint64_t first_pair);
SWMI_API int swmi_sharded_upload(swmi_sharded_batch *b, const uint8_t *seq1s, const uint8_t *seq2s);
SWMI_API int swmi_sharded_score(swmi_sharded_batch *b, const int8_t score_matrix[16], int8_t gap_penalty, int gather);
SWMI_API int swmi_sharded_wait(swmi_sharded_batch *b);
/* scores[0..n) in pair order from the shards (synchronous, D2H from every GPU into its slice) */
SWMI_API int swmi_sharded_scores_host(swmi_sharded_batch *b, int32_t *scores);
/* device pointer of the gathered int32[n] vector on GPU `index` (valid after a score call with ROOT (index 0) / ALL) */
SWMI_API int swmi_sharded_gathered_device(swmi_sharded_batch *b, int index, void **d_scores);
/* the same vector copied to host memory (synchronous): what GPU `index` holds after the gather */
SWMI_API int swmi_sharded_gathered_host(swmi_sharded_batch *b, int index, int32_t *scores);
/* What SWMI_GATHER_ALL runs on for this batch: 2 = RCCL, 1 = peer copies (a GPU bound twice, librccl not loadable,
 * ncclCommInitAll failed, or SWMI_GATHER_BACKEND=p2p), 0 = not decided yet (no SWMI_GATHER_ALL call so far).  Falling back
 * to peer copies is not an error (same bytes), but it is never silent: the call that decides leaves the reason in
 * swmi_last_error() while returning SWMI_OK, and swmi_sharded_gather_note() returns it at any later time ("" while RCCL
 * is in use or nothing is decided). */
SWMI_API int swmi_sharded_gather_backend(swmi_sharded_batch *b);
SWMI_API int swmi_sharded_gather_note(swmi_sharded_batch *b, char *text, size_t text_len);
/* 1 if librccl can be loaded with every entry point the gather needs, else 0 with the loader's reason in `why`.  Needs no
 * device.  (The library is dlopen()ed on first use, never linked; SWMI_RCCL_LIB names another file to load.) */
SWMI_API int swmi_rccl_probe(char *why, size_t why_len);
/* Measurement: `iters` score calls back to back; kernel_ms[g] = average kernel time of GPU g (HIP events on its stream),
 * gather_ms[g] = average time from the end of GPU g's kernel to the end of its part of the gather, *wall_ms = host wall
 * time per call, everything drained.  kernel_ms / gather_ms have swmi_num_gpus() entries (NULL to skip). */
SWMI_API int swmi_sharded_time(swmi_sharded_batch *b, const int8_t score_matrix[16], int8_t gap_penalty, int gather,
                               int iters, float *kernel_ms, float *gather_ms, double *wall_ms);

/* ---- extension: banded affine-gap scoring (BASELINE.json configs[4]; NO reference counterpart) -----------
 * The reference has linear gaps and no band on this path, so this entry point replaces
 * nothing in source.cpp; its semantics are defined by oracle/sw_oracle.c sw_oracle_banded_affine() and its
 * parity is NOT pinned by the reference.  n pairs of `len`-mers (64 <= len <= 1792, pair k at byte offset
 * len*k), local alignment restricted to the 128 diagonals -64 <= j - i <= 63, a gap of length k costs
 * gap_open + (k-1)*gap_extend (both in [0,127]).  One wavefront per alignment -- or per TWO alignments that share every
 * register as 16-bit halves (sw_banded_affine_pk_kernel, round 4) where len * max(s) + 2 max(0, -min s) + gap_open +
 * gap_extend + 64 < 0x7C00; swmi_banded_affine_kernel_for() reports which -- see DESIGN.md section 9. */
SWMI_API int swmi_score_banded_affine(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int len,
                                      const int8_t score_matrix[16], int gap_open, int gap_extend,
                                      int32_t *scores);
SWMI_API int swmi_score_banded_affine_device(const void *d_seq1s, const void *d_seq2s, size_t n, int len,
                                             const int8_t score_matrix[16], int gap_open, int gap_extend,
                                             void *d_scores, void *stream);
/* Which kernel instantiation a banded-affine launch with these parameters runs, e.g. "sw_banded_affine_pk_kernel<1>" or
 * "sw_banded_affine_kernel<1,1>", and how many alignments one wavefront scores (2 / 1; NULL to skip).  Needs no device. */
SWMI_API int swmi_banded_affine_kernel_for(int len, const int8_t score_matrix[16], int gap_open, int gap_extend, char *name,
                                           size_t name_len, int *alignments_per_wavefront);

/* ---- semi-global adaptive-band X-drop aligner -------------------------------------
 * Replaces SemiGlobal_AdaptiveBanded_XDrop_111_32_70 and its _simd / _simd_mark2..4 variants
 * (source.cpp:1836-1976, :1978-2725; call sites TestSemiGlobal :2774-2778, SpeedtestSemiGlobal :2818-2856):
 * two 16384-mers per alignment (alignment k at byte offset 16384*k), match +1 / mismatch -1 / gap -1, band of 32,
 * X-drop 70, result = (score, traceback).  scores[k] = .first; tracebacks + k*cap*2 receives the (i, j) pairs of
 * .second in the reference's order (from (0,0) to the best cell), at most `cap` of them; lengths[k] = .second.size()
 * (<= 32769).  Host buffers.  Where the reference reads one byte past its padded sequences (the band at the very
 * last position, source.cpp:1917-1919) this implementation reads a pad.  Bases must be 0..3: the reference's sweep scores
 * any other byte as a mismatch against everything (:1918-1920) but its traceback indexes the 4x4 matrix with it (:1961,
 * out of range); here such a byte is a mismatch against everything in both. */
#define SWMI_SG_LEN 16384
#define SWMI_SG_MAX_TRACEBACK 32769
SWMI_API int swmi_semiglobal_xdrop(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int32_t *scores,
                                   int32_t *tracebacks, size_t cap, uint32_t *lengths);
/* Same with every buffer resident in device memory (every pointer 16-byte aligned: the kernels use 16-byte loads and
 * 8-byte stores); asynchronous on `stream`.  The library keeps one workspace per (GPU, stream) of ~0.29 MB per alignment
 * (2-bit predecessor codes, the band's move bits, packed character streams, traceback moves), grown on demand and kept until
 * swmi_semiglobal_release_workspaces() / swmi_shutdown(): calls on one stream serialise by themselves, calls on
 * different streams use different workspaces and may be in flight together, from any threads. */
SWMI_API int swmi_semiglobal_xdrop_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores,
                                          void *d_tracebacks, size_t cap, void *d_lengths, void *stream);
/* The same alignment with the traceback returned as MOVES instead of positions (round 4): the list of source.cpp:1962-1975
 * is 8 bytes per position, up to 262 KB per alignment, which made the host entry above 11 x slower than the device entry --
 * it ships positions over PCIe.  The walk itself produces 2 bits per step: moves + k * SWMI_SG_MOVE_WORDS receives
 * alignment k's steps in WALKING order (step 0 leaves the best cell, the last one arrives at (0, 0)), step t at bits
 * 2 (t % 32) of word t / 32: 3 = diagonal (i - 1, j - 1), 2 = up (i - 1), 1 = left (j - 1); lengths[k] = positions of the
 * reference's list = steps + 1; words past the last step are unspecified.  swmi_semiglobal_expand_moves() turns one
 * alignment's moves into the reference's (i, j) list on the host (no device; `cap` positions at most) -- the C++ overload of
 * swmi_compat.hpp does that on several threads.  The best cell is (number of steps with bit 1, number with bit 0). */
#define SWMI_SG_MOVE_WORDS 1040      /* 1025 words hold the longest path; rows are padded to whole 128-byte lines */
SWMI_API int swmi_semiglobal_xdrop_moves(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int32_t *scores,
                                         uint64_t *moves, uint32_t *lengths);
SWMI_API int swmi_semiglobal_xdrop_moves_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores,
                                                void *d_moves, void *d_lengths, void *stream);
SWMI_API int swmi_semiglobal_expand_moves(const uint64_t *moves, uint32_t length, int32_t *traceback, size_t cap);
/* Which sweep kernel the aligner runs is chosen from the batch size; this overrides the choice, the
 * way swmi_set_schedule does for the scorer -- every mapping returns the same (score, traceback), tests/test_semiglobal.py
 * runs them all.  sweep: -1 = automatic, 4 / 2 / 1 = the band over 4 / 2 lanes or in one lane (16 / 32 / 64 alignments per
 * wavefront; the build for the most wavefronts per SIMD), 10 * lanes + W = that mapping compiled for W wavefronts per SIMD (41..44, 21..23, 11..12).  Anything else:
 * SWMI_ERR_INVALID_ARGUMENT.  Process-wide, one atomic word.  SWMI_SG_SWEEP in the environment sets the initial value at
 * swmi_init* (a value this call would reject is ignored).  (The traceback has one mapping: a lane per walk + expand kernel.) */
SWMI_API int swmi_semiglobal_set_mapping(int sweep);
/* The sweeps skip the X-drop test in windows of 16 rounds in which no band cell of the wavefront's alignments can reach the
 * threshold (a margin test once per window, DESIGN.md section 10: the same results by construction).  exact_only = 1 makes
 * every round run the test (A/B, tests); 0 = default.  Process-wide; SWMI_SG_EXACT in the environment gives the initial
 * value at swmi_init*. */
SWMI_API int swmi_semiglobal_set_exact(int exact_only);
/* What the last swmi_semiglobal_xdrop[_moves]_device call on `stream` of the current GPU ran: counts[0] = windows of 8 rounds
 * summed over its sweep wavefronts, counts[1] = how many of them were calm (no X-drop test); counts[2] = windows of 16 rounds
 * summed over its traceback wavefronts, counts[3] = how many of them were decoded a second time because a walk left band cells
 * 8 .. 23 (the traceback fetches that half of the predecessor records only, the other half on demand: DESIGN.md section 10).
 * Waits for the stream.  SWMI_ERR_INVALID_ARGUMENT when no call has run on that stream. */
SWMI_API int swmi_semiglobal_window_stats(void *stream, uint64_t counts[4]);
/* Free the per-stream workspaces of the current GPU (synchronises the device first). */
SWMI_API int swmi_semiglobal_release_workspaces(void);
/* Names of the sweep and traceback kernels a call with n alignments runs on the current GPU (the mapping depends on the
 * batch size and the device's CU count) -- so that a profiler-side tool asks instead of guessing. */
SWMI_API int swmi_semiglobal_kernels_for_batch(size_t n, char *sweep, size_t sweep_len, char *traceback, size_t traceback_len);
/* Measurement helper (no reference counterpart): one swmi_semiglobal_xdrop_device call bracketed by HIP events on
 * `stream`, synchronous; phase_ms[0] = the sweep kernel (source.cpp:1886-1949), phase_ms[1] = the traceback kernel
 * (source.cpp:1951-1975). */
SWMI_API int swmi_semiglobal_time_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores,
                                         void *d_tracebacks, size_t cap, void *d_lengths, void *stream, float phase_ms[2]);

/* unpack() itself (source.cpp:1580-1583) for n packed sequences, on the GPU. Host buffers. */
SWMI_API int swmi_unpack(const uint8_t *packed, size_t n_seqs, uint8_t *unpacked);

/* ---- deferred queue behind the per-pair signature -------------------------------------
 * Lets a per-pair caller (the reference's timing loop) keep its call shape while the
 * library batches: submit() copies the pair into pinned staging memory and returns its
 * ticket (0,1,2,...); full staging blocks are shipped to the GPU asynchronously;
 * swmi_queue_wait() drains everything and exposes scores[ticket]. */
typedef struct swmi_queue swmi_queue;
SWMI_API int swmi_queue_create(size_t max_pairs, const int8_t score_matrix[16], int8_t gap_penalty,
                               swmi_queue **out);
SWMI_API long long swmi_queue_submit(swmi_queue *q, const uint8_t seq1[SWMI_SEQ_LEN],
                                     const uint8_t seq2[SWMI_SEQ_LEN]);
SWMI_API int swmi_queue_wait(swmi_queue *q, const int32_t **scores, size_t *n_scores);
SWMI_API int swmi_queue_reset(swmi_queue *q);
SWMI_API int swmi_queue_destroy(swmi_queue *q);

/* ---- schedules ------------------------------------------------------------------------
 * The reference ships nine schedules of one semantics (simd .. simd9).  So does this
 * library: `lanes_per_alignment` L in {64,32,16,8,4,2} lanes of a 64-lane wavefront walk one
 * alignment's anti-diagonal (each lane owns 128/L consecutive rows); L = 64 is literally
 * "one wavefront per alignment".  0 (the default) lets the library choose by batch size: L = 4 from ~100 000 pairs per
 * launch on (fewest instructions per cell), more lanes per alignment below that (lowest latency: L = 64 up to 2048 pairs,
 * then 32, 16, 8); swmi_get_schedule reports 0 in that case.
 * flags (all give identical scores): bit 0 = never fold the gap into the matrix rows (general cell body);
 * bit 1 = 16-bit-max cell body; bit 2 = LDS score-lookup kernel (L in 16, 8, 4 and foldable parameters only);
 * bit 3 = never the packed kernel.  (Without it, L = 4, 8 and 16 -- what the automatic choice resolves to from 5121 pairs
 * up -- run sw128_pk_kernel<MODE, VARIANT, L>: two alignments per register and L lanes per PAIR of alignments, i.e. 32 / 16 / 8
 * alignments per wavefront, 16-bit cells, v_pk_maximum3_f16 as a packed integer max.  VARIANT is the cell body, chosen from
 * the parameters: 0 when every score_matrix entry + gap_penalty is >= 0 (e.g. (1,-1,1), the parameters of the reference's
 * SmithWaterman_8bit111simd, source.cpp:1105-1225: ~1.55x the int32 kernel), 2 when every entry + 2 * gap_penalty lies in
 * [0, 255] (e.g. the harness's (10,-30,15): ~1.4x), 1 otherwise (~1.3x).  DESIGN.md section 5a.) */
SWMI_API int swmi_set_schedule(int lanes_per_alignment, unsigned flags);
SWMI_API int swmi_get_schedule(int *lanes_per_alignment, unsigned *flags);
/* Lanes per alignment a launch of n pairs runs with under the current setting (what 0 = automatic resolves to). */
SWMI_API int swmi_schedule_for_batch(size_t n);
/* Which kernel instantiation a launch of n pairs with these parameters runs under the current setting, e.g.
 * "sw128_pk_kernel<0,1>" or "sw128_kernel<64,1,0,0>" (template arguments as tools/isa_census.py prints them), and how many
 * alignments one of its wavefronts scores.  mode: 0 = pairs, 1 = 2-bit packed input, 2 = one-vs-many.  For profiling
 * tools and bench.py, which derive the issue-bound fraction from the disassembly of exactly that kernel. */
SWMI_API int swmi_score_kernel_for_batch(size_t n, const int8_t score_matrix[16], int8_t gap_penalty, int mode,
                                         char *name, size_t name_len, int *alignments_per_wavefront);

/* Self-test of the premise the packed kernel rests on: gfx950's v_pk_maximum3_f16, applied to 16-bit integers in
 * [0, 0x7C00) held two per register, is a packed THREE-INPUT INTEGER MAX (such integers order like the half-precision
 * numbers with the same bit patterns; the kernels pin MODE.FP_DENORM so that the patterns below 1024, which are f16
 * denormals, are kept).  Runs the instruction on EVERY pair (a, b) of that range -- 31744^2 pairs, six operand
 * arrangements each, the third operand one of the two or a pseudo-random third value -- inside a kernel that sets the
 * mode exactly as the scoring kernels do, and compares with the integer maximum of each half on the device.
 * *checked = comparisons made (6 * 31744^2), *mismatches = how many failed (0 on gfx950).  No reference counterpart. */
SWMI_API int swmi_selftest_pk_max3(unsigned long long *checked, unsigned long long *mismatches);

/* ---- synthetic inputs ----------------------------------------------------
 * Counter-based generator, identical on host and device: pair p, sequence s (0/1), 64-bit
 * word w (0..3): x = splitmix64(seed ^ ((p*2+s)*4+w) * 0x9E3779B97F4A7C15); base k of the
 * word = (x >> 2k) & 3.  Stands in for the reference's mt19937_64 + uniform_int_distribution
 * draw (source.cpp:3033-3040), which is implementation-defined across standard libraries. */
SWMI_API int swmi_generate_pairs_device(void *d_seq1s, void *d_seq2s, size_t n, uint64_t seed,
                                        uint64_t first_pair, void *stream);
SWMI_API int swmi_generate_pairs_host(uint8_t *seq1s, uint8_t *seq2s, size_t n, uint64_t seed,
                                      uint64_t first_pair);

/* ---- measurement ----------------------------------------------------------------------
 * Launches the batch kernel `iters` times back to back on `stream` (NULL = the null stream)
 * bracketed by hipEvents on that same stream and returns the average per-launch duration. */
SWMI_API int swmi_time_batch_device(const void *d_seq1s, const void *d_seq2s, size_t n,
                                    const int8_t score_matrix[16], int8_t gap_penalty,
                                    void *d_scores, void *stream, int iters, float *avg_ms);

typedef struct swmi_device_info {
    int device;
    int compute_units;
    int clock_khz;
    int wavefront_size;
    size_t hbm_bytes;
    char arch[64];
    char name[128];
} swmi_device_info;
SWMI_API int swmi_get_device_info(swmi_device_info *info);

#ifdef __cplusplus
}
#endif
#endif /* SWMI_H */
