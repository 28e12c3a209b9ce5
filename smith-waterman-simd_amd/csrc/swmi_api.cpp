// swmi_api.cpp -- host side of libswmi.so: the C ABI of include/swmi.h over the gfx950 kernels.
//
// Host responsibilities only: argument checking, score-matrix packing, chunked H2D / kernel / D2H
// pipelining for host-resident batches, the deferred queue behind the per-pair signature, and
// hipEvent timing.  There is deliberately no CPU implementation of the scoring path in this
// library: without a usable gfx950 device every scoring entry point returns an error.
#include "../../include/swmi.h"
#include "swmi_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace {

using swmi::LaunchConfig;
using swmi::SmRows;

thread_local char t_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_error, sizeof t_error, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(SWMI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr size_t kSeq = SWMI_SEQ_LEN;
constexpr size_t kChunkPairs = size_t(1) << 20;      // host-batch pipeline granule: 1M pairs = 128 MiB per input array
constexpr size_t kMaxLaunchPairs = size_t(1) << 30;  // pairs per kernel launch (kernel indexes pairs with uint32)
constexpr int kSlots = 2;
constexpr size_t kPinPairs = 64;        // host batches up to this size go through the pinned staging buffer

struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *d_seq1 = nullptr, *d_seq2 = nullptr;
    int32_t *d_scores = nullptr;
    size_t capacity = 0;   // pairs
};

struct Context {
    bool ready = false;
    int device = -1;
    hipDeviceProp_t prop{};
    hipStream_t stream = nullptr;       // library-owned stream (host-side helpers)
    Slot slots[kSlots];
    int lanes = 0;                      // 0 = automatic: by batch size (auto_lanes)
    unsigned flags = 0;
    void *sg_workspace = nullptr;       // semi-global aligner workspace (device), grown on demand
    size_t sg_workspace_bytes = 0;
    unsigned extra_lds = 0;             // SWMI_EXTRA_LDS: occupancy sweep knob (BASELINE config 3)
    // pinned, device-visible staging for tiny host batches (the per-pair call): the kernel reads the pairs from host
    // memory and writes the scores back there, so a call is one launch + one synchronisation, no copies
    // device buffers of the host-buffer semi-global entry (two chunks in flight), kept between calls and grown on demand
    struct SgSet {
        uint8_t *d1 = nullptr, *d2 = nullptr;
        void *ws = nullptr;
        int32_t *d_scores = nullptr, *d_tb = nullptr;
        uint32_t *d_len = nullptr;
        size_t alignments = 0, tb_entries = 0;      // capacity
        size_t off = 0, m = 0;                      // chunk in flight
    } sg_sets[2];
    uint8_t *pin = nullptr;             // [kPinPairs * 128] seq1s, [kPinPairs * 128] seq2s, [kPinPairs] int32 scores
    void *pin_dev = nullptr;            // the same memory as the device sees it
    std::mutex mu;                      // serialises use of the slots
};

Context g_ctx;
std::mutex g_init_mu;

int check_ready()
{
    if (!g_ctx.ready) return fail(SWMI_ERR_NOT_INITIALIZED, "swmi_init() has not been called (or failed)");
    return SWMI_OK;
}

int check_params(const int8_t *sm, int gap)
{
    if (!sm) return fail(SWMI_ERR_INVALID_ARGUMENT, "score_matrix is NULL");
    if (gap < 0) return fail(SWMI_ERR_DOMAIN, "gap_penalty %d < 0 is outside the supported domain [0,127]", gap);
    return SWMI_OK;
}

// rows.r[a] = sm[a*4 + 0..3] (+ gap when folded), one int8 per byte
SmRows pack_rows(const int8_t *sm, int add)
{
    SmRows rows;
    for (int a = 0; a < 4; ++a) {
        uint32_t r = 0;
        for (int b = 0; b < 4; ++b) r |= uint32_t(uint8_t(int8_t(sm[4 * a + b] + add))) << (8 * b);
        rows.r[a] = r;
    }
    return rows;
}

// Lanes per alignment when the caller has not fixed a schedule: L = 4 (32 rows per lane) issues the fewest instructions
// per cell and wins once the batch fills the chip; a small batch wants many lanes per alignment instead -- a single pair
// takes 6 us with L = 64 and 42 us with L = 4 (tools/small_batch_schedule.py, profiles/r01_small_batch_schedule.txt).
int auto_lanes(size_t n)
{
    return n <= 2048 ? 64 : n <= 5120 ? 32 : n <= 20480 ? 16 : 4;
}

// Choose the schedule and the cell body: the gap-folded recurrence needs every sm + gap to fit int8.
LaunchConfig make_config(const int8_t *sm, int gap, SmRows *rows, size_t n)
{
    LaunchConfig cfg;
    cfg.lanes_per_alignment = g_ctx.lanes ? g_ctx.lanes : (g_ctx.flags & swmi::kUseLut) ? 4 : auto_lanes(n);
    cfg.use_i16 = (g_ctx.flags & swmi::kUseI16) != 0;
    cfg.use_lut = (g_ctx.flags & swmi::kUseLut) != 0;
    cfg.extra_lds_bytes = g_ctx.extra_lds;
    bool fold = !(g_ctx.flags & swmi::kNoGapFold);
    for (int k = 0; k < 16 && fold; ++k) {
        const int v = int(sm[k]) + gap;
        if (v < -128 || v > 127) fold = false;
    }
    cfg.fold_gap = fold;
    *rows = pack_rows(sm, fold ? gap : 0);
    return cfg;
}

int ensure_slot(Slot &s, size_t pairs)
{
    if (s.capacity >= pairs) return SWMI_OK;
    if (s.d_seq1) { (void)hipFree(s.d_seq1); s.d_seq1 = nullptr; }
    if (s.d_seq2) { (void)hipFree(s.d_seq2); s.d_seq2 = nullptr; }
    if (s.d_scores) { (void)hipFree(s.d_scores); s.d_scores = nullptr; }
    s.capacity = 0;
    HIP_TRY(hipMalloc(&s.d_seq1, pairs * kSeq));
    HIP_TRY(hipMalloc(&s.d_seq2, pairs * kSeq));
    HIP_TRY(hipMalloc(&s.d_scores, pairs * sizeof(int32_t)));
    s.capacity = pairs;
    return SWMI_OK;
}

int launch_device(const void *d1, const void *d2, size_t n, const int8_t *sm, int gap, void *d_out, hipStream_t st,
                  bool packed)
{
    SmRows rows;
    const LaunchConfig cfg = make_config(sm, gap, &rows, n);
    const size_t stride = packed ? SWMI_PACKED_LEN : kSeq;
    for (size_t off = 0; off < n; off += kMaxLaunchPairs) {
        const size_t m = n - off < kMaxLaunchPairs ? n - off : kMaxLaunchPairs;
        HIP_TRY(swmi::launch_score(cfg, static_cast<const uint8_t *>(d1) + off * stride,
                                   static_cast<const uint8_t *>(d2) + off * stride,
                                   static_cast<int32_t *>(d_out) + off, m, rows, gap, packed, st));
    }
    return SWMI_OK;
}

// Host-resident batch: two slots, each with its own stream; chunk k+1's H2D copy overlaps chunk k's kernel.
int score_host_batch(const uint8_t *s1, const uint8_t *s2, size_t n, const int8_t *sm, int gap, int32_t *out,
                     bool packed, bool one_vs_many)
{
    std::lock_guard<std::mutex> lock(g_ctx.mu);
    const size_t in_stride = packed ? SWMI_PACKED_LEN : kSeq;
    const size_t chunk = n < kChunkPairs ? n : kChunkPairs;
    SmRows rows;
    const LaunchConfig cfg = make_config(sm, gap, &rows, chunk);
    for (int k = 0; k < kSlots && size_t(k) * chunk < n; ++k) {
        const int rc = ensure_slot(g_ctx.slots[k], chunk);
        if (rc != SWMI_OK) return rc;
    }
    if (one_vs_many)   // the single seq2 goes to the head of slot 0's seq2 buffer... of every slot in use
        for (int k = 0; k < kSlots && size_t(k) * chunk < n; ++k)
            HIP_TRY(hipMemcpyAsync(g_ctx.slots[k].d_seq2, s2, kSeq, hipMemcpyHostToDevice, g_ctx.slots[k].stream));
    size_t idx = 0;
    for (size_t off = 0; off < n; off += chunk, ++idx) {
        Slot &s = g_ctx.slots[idx % kSlots];
        const size_t m = n - off < chunk ? n - off : chunk;
        HIP_TRY(hipMemcpyAsync(s.d_seq1, s1 + off * in_stride, m * in_stride, hipMemcpyHostToDevice, s.stream));
        if (!one_vs_many)
            HIP_TRY(hipMemcpyAsync(s.d_seq2, s2 + off * in_stride, m * in_stride, hipMemcpyHostToDevice, s.stream));
        if (one_vs_many)
            HIP_TRY(swmi::launch_score_one_vs_many(cfg, s.d_seq1, s.d_seq2, s.d_scores, m, rows, gap, s.stream));
        else
            HIP_TRY(swmi::launch_score(cfg, s.d_seq1, s.d_seq2, s.d_scores, m, rows, gap, packed, s.stream));
        HIP_TRY(hipMemcpyAsync(out + off, s.d_scores, m * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
    }
    for (int k = 0; k < kSlots; ++k)
        if (g_ctx.slots[k].stream) HIP_TRY(hipStreamSynchronize(g_ctx.slots[k].stream));
    return SWMI_OK;
}

}  // namespace

// ---- deferred queue ------------------------------------------------------------------------------

struct swmi_queue {
    size_t max_pairs = 0, count = 0, shipped = 0;
    size_t block = 1 << 16;             // pairs per asynchronous shipment
    int8_t sm[16];
    int gap = 0;
    uint8_t *h_seq1 = nullptr, *h_seq2 = nullptr;   // pinned staging, max_pairs * 128 each
    int32_t *h_scores = nullptr;                    // pinned results
    uint8_t *d_seq1 = nullptr, *d_seq2 = nullptr;
    int32_t *d_scores = nullptr;
    hipStream_t stream = nullptr;
};

namespace {
int queue_ship(swmi_queue *q, size_t upto)
{
    if (upto <= q->shipped) return SWMI_OK;
    const size_t off = q->shipped, m = upto - q->shipped;
    HIP_TRY(hipMemcpyAsync(q->d_seq1 + off * kSeq, q->h_seq1 + off * kSeq, m * kSeq, hipMemcpyHostToDevice, q->stream));
    HIP_TRY(hipMemcpyAsync(q->d_seq2 + off * kSeq, q->h_seq2 + off * kSeq, m * kSeq, hipMemcpyHostToDevice, q->stream));
    const int rc = launch_device(q->d_seq1 + off * kSeq, q->d_seq2 + off * kSeq, m, q->sm, q->gap, q->d_scores + off,
                                 q->stream, false);
    if (rc != SWMI_OK) return rc;
    HIP_TRY(hipMemcpyAsync(q->h_scores + off, q->d_scores + off, m * sizeof(int32_t), hipMemcpyDeviceToHost, q->stream));
    q->shipped = upto;
    return SWMI_OK;
}
}  // namespace

extern "C" {

int swmi_version(void) { return SWMI_VERSION; }

const char *swmi_last_error(void) { return t_error; }

int swmi_init(int device)
{
    std::lock_guard<std::mutex> lock(g_init_mu);
    t_error[0] = 0;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(SWMI_ERR_NO_DEVICE, "no HIP device available (%s); libswmi has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0) {
        const char *lr = getenv("LOCAL_RANK");
        device = lr ? atoi(lr) % count : 0;
    }
    if (device >= count) return fail(SWMI_ERR_INVALID_ARGUMENT, "device %d out of range (count %d)", device, count);
    if (g_ctx.ready) {
        if (g_ctx.device == device) return SWMI_OK;
        return fail(SWMI_ERR_INVALID_ARGUMENT, "already initialised on device %d (one process per GPU); call swmi_shutdown() first",
                    g_ctx.device);
    }
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDeviceProperties(&g_ctx.prop, device));
    if (strncmp(g_ctx.prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SWMI_ERR_UNSUPPORTED_ARCH, "device %d is %s; libswmi carries gfx950 (MI355X) code objects only", device,
                    g_ctx.prop.gcnArchName);
    HIP_TRY(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
    for (auto &s : g_ctx.slots) HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    g_ctx.device = device;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&g_ctx.pin), kPinPairs * (2 * kSeq + sizeof(int32_t)), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(&g_ctx.pin_dev, g_ctx.pin, 0));
    const char *env_x = getenv("SWMI_EXTRA_LDS");
    g_ctx.extra_lds = env_x ? (unsigned)atoi(env_x) : 0;
    const char *env_l = getenv("SWMI_LANES");
    if (env_l && swmi::schedule_supported(atoi(env_l))) g_ctx.lanes = atoi(env_l);
    g_ctx.ready = true;
    return SWMI_OK;
}

int swmi_shutdown(void)
{
    std::lock_guard<std::mutex> lock(g_init_mu);
    if (!g_ctx.ready) return SWMI_OK;
    (void)hipSetDevice(g_ctx.device);
    (void)hipDeviceSynchronize();
    for (auto &s : g_ctx.slots) {
        if (s.d_seq1) (void)hipFree(s.d_seq1);
        if (s.d_seq2) (void)hipFree(s.d_seq2);
        if (s.d_scores) (void)hipFree(s.d_scores);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        s = Slot{};
    }
    if (g_ctx.stream) (void)hipStreamDestroy(g_ctx.stream);
    g_ctx.stream = nullptr;
    if (g_ctx.sg_workspace) (void)hipFree(g_ctx.sg_workspace);
    g_ctx.sg_workspace = nullptr;
    g_ctx.sg_workspace_bytes = 0;
    for (auto &g : g_ctx.sg_sets) {
        (void)hipFree(g.d1); (void)hipFree(g.d2); (void)hipFree(g.ws); (void)hipFree(g.d_scores); (void)hipFree(g.d_len); (void)hipFree(g.d_tb);
        g = Context::SgSet{};
    }
    if (g_ctx.pin) (void)hipHostFree(g_ctx.pin);
    g_ctx.pin = nullptr;
    g_ctx.pin_dev = nullptr;
    g_ctx.ready = false;
    g_ctx.device = -1;
    return SWMI_OK;
}

int swmi_set_schedule(int lanes_per_alignment, unsigned flags)
{
    if (lanes_per_alignment != 0 && !swmi::schedule_supported(lanes_per_alignment))
        return fail(SWMI_ERR_INVALID_ARGUMENT, "lanes_per_alignment must be one of 64,32,16,8,4,2 (got %d)", lanes_per_alignment);
    if (flags & ~7u) return fail(SWMI_ERR_INVALID_ARGUMENT, "unknown schedule flags 0x%x", flags);
    g_ctx.lanes = lanes_per_alignment;
    g_ctx.flags = flags;
    return SWMI_OK;
}

int swmi_get_schedule(int *lanes_per_alignment, unsigned *flags)
{
    if (lanes_per_alignment) *lanes_per_alignment = g_ctx.lanes;      // 0 = automatic
    if (flags) *flags = g_ctx.flags;
    return SWMI_OK;
}

int swmi_schedule_for_batch(size_t n)
{
    return g_ctx.lanes ? g_ctx.lanes : (g_ctx.flags & swmi::kUseLut) ? 4 : auto_lanes(n);
}

int swmi_get_device_info(swmi_device_info *info)
{
    if (!info) return fail(SWMI_ERR_INVALID_ARGUMENT, "info is NULL");
    int rc = check_ready();
    if (rc != SWMI_OK) return rc;
    memset(info, 0, sizeof *info);
    info->device = g_ctx.device;
    info->compute_units = g_ctx.prop.multiProcessorCount;
    info->clock_khz = g_ctx.prop.clockRate;
    info->wavefront_size = g_ctx.prop.warpSize;
    info->hbm_bytes = g_ctx.prop.totalGlobalMem;
    snprintf(info->arch, sizeof info->arch, "%s", g_ctx.prop.gcnArchName);
    snprintf(info->name, sizeof info->name, "%s", g_ctx.prop.name);
    return SWMI_OK;
}

int swmi_score_batch(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, const int8_t score_matrix[16],
                     int8_t gap_penalty, int32_t *scores)
{
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!seq1s || !seq2s || !scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    HIP_TRY(hipSetDevice(g_ctx.device));
    if (n <= kPinPairs) {               // the per-pair call and its small relatives: no copies, one launch, one wait
        std::lock_guard<std::mutex> lock(g_ctx.mu);
        uint8_t *h1 = g_ctx.pin, *h2 = g_ctx.pin + kPinPairs * kSeq;
        int32_t *hs = reinterpret_cast<int32_t *>(g_ctx.pin + 2 * kPinPairs * kSeq);
        uint8_t *dev = static_cast<uint8_t *>(g_ctx.pin_dev);
        memcpy(h1, seq1s, n * kSeq);
        memcpy(h2, seq2s, n * kSeq);
        rc = launch_device(dev, dev + kPinPairs * kSeq, n, score_matrix, gap_penalty, dev + 2 * kPinPairs * kSeq, g_ctx.stream, false);
        if (rc != SWMI_OK) return rc;
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        memcpy(scores, hs, n * sizeof(int32_t));
        return SWMI_OK;
    }
    return score_host_batch(seq1s, seq2s, n, score_matrix, gap_penalty, scores, false, false);
}

int swmi_score_pair(const uint8_t seq1[SWMI_SEQ_LEN], const uint8_t seq2[SWMI_SEQ_LEN], const int8_t score_matrix[16],
                    int8_t gap_penalty)
{
    int32_t score = 0;
    const int rc = swmi_score_batch(seq1, seq2, 1, score_matrix, gap_penalty, &score);
    return rc == SWMI_OK ? score : rc;
}

int swmi_score_one_vs_many(const uint8_t *seq1s, size_t n_seq1, const uint8_t seq2[SWMI_SEQ_LEN],
                           const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores)
{
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (n_seq1 == 0) return SWMI_OK;
    if (!seq1s || !seq2 || !scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n_seq1 = %zu", n_seq1);
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    HIP_TRY(hipSetDevice(g_ctx.device));
    return score_host_batch(seq1s, seq2, n_seq1, score_matrix, gap_penalty, scores, false, true);
}

int swmi_score_one_vs_many_device(const void *d_seq1s, size_t n_seq1, const void *d_seq2, const int8_t score_matrix[16],
                                  int8_t gap_penalty, void *d_scores, void *stream)
{
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (n_seq1 == 0) return SWMI_OK;
    if (!d_seq1s || !d_seq2 || !d_scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n_seq1 = %zu", n_seq1);
    if ((reinterpret_cast<uintptr_t>(d_seq1s) | reinterpret_cast<uintptr_t>(d_seq2) | reinterpret_cast<uintptr_t>(d_scores)) & 15)
        return fail(SWMI_ERR_ALIGNMENT, "device pointers must be 16-byte aligned");
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    SmRows rows;
    const LaunchConfig cfg = make_config(score_matrix, gap_penalty, &rows, n_seq1);
    for (size_t off = 0; off < n_seq1; off += kMaxLaunchPairs) {
        const size_t m = n_seq1 - off < kMaxLaunchPairs ? n_seq1 - off : kMaxLaunchPairs;
        HIP_TRY(swmi::launch_score_one_vs_many(cfg, static_cast<const uint8_t *>(d_seq1s) + off * kSeq,
                                               static_cast<const uint8_t *>(d_seq2), static_cast<int32_t *>(d_scores) + off, m,
                                               rows, gap_penalty, static_cast<hipStream_t>(stream)));
    }
    return SWMI_OK;
}

int swmi_score_batch_packed(const uint8_t *seq1s_packed, const uint8_t *seq2s_packed, size_t n,
                            const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores)
{
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!seq1s_packed || !seq2s_packed || !scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    HIP_TRY(hipSetDevice(g_ctx.device));
    return score_host_batch(seq1s_packed, seq2s_packed, n, score_matrix, gap_penalty, scores, true, false);
}

static int device_entry(const void *d1, const void *d2, size_t n, const int8_t *sm, int gap, void *d_out, void *stream,
                        bool packed)
{
    int rc = check_params(sm, gap);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!d1 || !d2 || !d_out) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    if ((reinterpret_cast<uintptr_t>(d1) | reinterpret_cast<uintptr_t>(d2) | reinterpret_cast<uintptr_t>(d_out)) & 15)
        return fail(SWMI_ERR_ALIGNMENT, "device pointers must be 16-byte aligned");
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);   // NULL = the HIP null (default) stream, as in any HIP call
    return launch_device(d1, d2, n, sm, gap, d_out, st, packed);
}

int swmi_score_batch_device(const void *d_seq1s, const void *d_seq2s, size_t n, const int8_t score_matrix[16],
                            int8_t gap_penalty, void *d_scores, void *stream)
{
    return device_entry(d_seq1s, d_seq2s, n, score_matrix, gap_penalty, d_scores, stream, false);
}

int swmi_score_batch_packed_device(const void *d_seq1s_packed, const void *d_seq2s_packed, size_t n,
                                   const int8_t score_matrix[16], int8_t gap_penalty, void *d_scores, void *stream)
{
    return device_entry(d_seq1s_packed, d_seq2s_packed, n, score_matrix, gap_penalty, d_scores, stream, true);
}

static int check_banded(const int8_t *sm, int len, int open, int ext)
{
    if (!sm) return fail(SWMI_ERR_INVALID_ARGUMENT, "score_matrix is NULL");
    if (len < 64 || len > 1792) return fail(SWMI_ERR_INVALID_ARGUMENT, "len %d outside [64, 1792]", len);
    if (open < 0 || open > 127 || ext < 0 || ext > 127)
        return fail(SWMI_ERR_DOMAIN, "gap_open %d / gap_extend %d outside [0,127]", open, ext);
    return SWMI_OK;
}

int swmi_score_banded_affine_device(const void *d_seq1s, const void *d_seq2s, size_t n, int len,
                                    const int8_t score_matrix[16], int gap_open, int gap_extend, void *d_scores, void *stream)
{
    int rc = check_banded(score_matrix, len, gap_open, gap_extend);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!d_seq1s || !d_seq2s || !d_scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    const SmRows rows = pack_rows(score_matrix, 0);
    const size_t max_launch = size_t(1) << 24;
    for (size_t off = 0; off < n; off += max_launch) {
        const size_t m = n - off < max_launch ? n - off : max_launch;
        HIP_TRY(swmi::launch_banded_affine(static_cast<const uint8_t *>(d_seq1s) + off * size_t(len),
                                           static_cast<const uint8_t *>(d_seq2s) + off * size_t(len),
                                           static_cast<int32_t *>(d_scores) + off, m, len, rows, gap_open, gap_extend,
                                           static_cast<hipStream_t>(stream)));
    }
    return SWMI_OK;
}

int swmi_score_banded_affine(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int len, const int8_t score_matrix[16],
                             int gap_open, int gap_extend, int32_t *scores)
{
    int rc = check_banded(score_matrix, len, gap_open, gap_extend);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!seq1s || !seq2s || !scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    std::lock_guard<std::mutex> lock(g_ctx.mu);
    HIP_TRY(hipSetDevice(g_ctx.device));
    // the slot buffers are sized in 128-byte pairs: a len-mer pair occupies ceil(len / 128) of them
    const size_t per = (size_t(len) + kSeq - 1) / kSeq;
    const size_t chunk_cap = kChunkPairs / per;
    const size_t chunk = n < chunk_cap ? n : chunk_cap;
    for (int k = 0; k < kSlots && size_t(k) * chunk < n; ++k) {
        rc = ensure_slot(g_ctx.slots[k], chunk * per);
        if (rc != SWMI_OK) return rc;
    }
    size_t idx = 0;
    for (size_t off = 0; off < n; off += chunk, ++idx) {
        Slot &s = g_ctx.slots[idx % kSlots];
        const size_t m = n - off < chunk ? n - off : chunk;
        HIP_TRY(hipMemcpyAsync(s.d_seq1, seq1s + off * size_t(len), m * size_t(len), hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.d_seq2, seq2s + off * size_t(len), m * size_t(len), hipMemcpyHostToDevice, s.stream));
        rc = swmi_score_banded_affine_device(s.d_seq1, s.d_seq2, m, len, score_matrix, gap_open, gap_extend, s.d_scores, s.stream);
        if (rc != SWMI_OK) return rc;
        HIP_TRY(hipMemcpyAsync(scores + off, s.d_scores, m * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
    }
    for (int k = 0; k < kSlots; ++k)
        if (g_ctx.slots[k].stream) HIP_TRY(hipStreamSynchronize(g_ctx.slots[k].stream));
    return SWMI_OK;
}

static int semiglobal_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores, void *d_tracebacks,
                             size_t cap, void *d_lengths, void *stream, hipEvent_t between)
{
    if (n == 0) return SWMI_OK;
    if (!d_seq1s || !d_seq2s || !d_scores || !d_lengths || (!d_tracebacks && cap != 0))
        return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    if (n > (size_t(1) << 18)) return fail(SWMI_ERR_INVALID_ARGUMENT, "at most 2^18 alignments per call (got %zu)", n);
    int rc = check_ready();
    if (rc != SWMI_OK) return rc;
    {
        std::lock_guard<std::mutex> lock(g_ctx.mu);
        const size_t need = swmi::semiglobal_workspace_bytes(n);
        if (need > g_ctx.sg_workspace_bytes) {
            HIP_TRY(hipDeviceSynchronize());              // the old workspace may still be in use by an earlier launch
            if (g_ctx.sg_workspace) (void)hipFree(g_ctx.sg_workspace);
            g_ctx.sg_workspace = nullptr;
            g_ctx.sg_workspace_bytes = 0;
            HIP_TRY(hipMalloc(&g_ctx.sg_workspace, need));
            g_ctx.sg_workspace_bytes = need;
        }
    }
    HIP_TRY(swmi::launch_semiglobal(static_cast<const uint8_t *>(d_seq1s), static_cast<const uint8_t *>(d_seq2s), n,
                                    g_ctx.sg_workspace, static_cast<int32_t *>(d_scores), static_cast<int32_t *>(d_tracebacks),
                                    cap, static_cast<uint32_t *>(d_lengths), static_cast<hipStream_t>(stream), between));
    return SWMI_OK;
}

int swmi_semiglobal_xdrop_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores, void *d_tracebacks,
                                 size_t cap, void *d_lengths, void *stream)
{
    return semiglobal_device(d_seq1s, d_seq2s, n, d_scores, d_tracebacks, cap, d_lengths, stream, nullptr);
}

int swmi_semiglobal_time_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores, void *d_tracebacks,
                                size_t cap, void *d_lengths, void *stream, float phase_ms[2])
{
    if (!phase_ms) return fail(SWMI_ERR_INVALID_ARGUMENT, "phase_ms is NULL");
    if (n == 0) return fail(SWMI_ERR_INVALID_ARGUMENT, "n is 0");
    int rc = check_ready();
    if (rc != SWMI_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipError_t he = hipSuccess;
    for (int k = 0; k < 3 && he == hipSuccess; ++k) he = hipEventCreate(&ev[k]);
    if (he == hipSuccess) he = hipEventRecord(ev[0], st);
    if (he == hipSuccess) rc = semiglobal_device(d_seq1s, d_seq2s, n, d_scores, d_tracebacks, cap, d_lengths, stream, ev[1]);
    if (he == hipSuccess && rc == SWMI_OK) he = hipEventRecord(ev[2], st);
    if (he == hipSuccess && rc == SWMI_OK) he = hipEventSynchronize(ev[2]);
    if (he == hipSuccess && rc == SWMI_OK) he = hipEventElapsedTime(&phase_ms[0], ev[0], ev[1]);
    if (he == hipSuccess && rc == SWMI_OK) he = hipEventElapsedTime(&phase_ms[1], ev[1], ev[2]);
    for (int k = 0; k < 3; ++k)
        if (ev[k]) (void)hipEventDestroy(ev[k]);
    if (he != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_semiglobal_time_device: %s", hipGetErrorString(he));
    return rc;
}

int swmi_semiglobal_xdrop(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int32_t *scores, int32_t *tracebacks,
                          size_t cap, uint32_t *lengths)
{
    if (n == 0) return SWMI_OK;
    if (!seq1s || !seq2s || !scores || !lengths || (!tracebacks && cap != 0))
        return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    int rc = check_ready();
    if (rc != SWMI_OK) return rc;
    std::lock_guard<std::mutex> lock(g_ctx.mu);
    HIP_TRY(hipSetDevice(g_ctx.device));
    constexpr size_t kLen = SWMI_SG_LEN;
    // Chunks of up to 8192 alignments (~0.35 MB of workspace + cap*8 B of output each), two sets of device buffers: while
    // the host is busy receiving chunk k (a copy into pageable memory blocks the caller), the GPU works on chunk k+1.
    const size_t chunk = n < 8192 ? n : 8192;
    using Set = Context::SgSet;
    Set *sets = g_ctx.sg_sets;
    const int n_sets = n > chunk ? 2 : 1;
    hipStream_t streams[2] = {g_ctx.slots[0].stream, g_ctx.slots[1].stream};
    hipError_t e = hipSuccess;
    for (int k = 0; k < n_sets && e == hipSuccess; ++k) {
        Set &s = sets[k];
        s.off = s.m = 0;
        const size_t tb_need = chunk * (cap ? cap : 1);
        if (s.alignments < chunk) {
            (void)hipFree(s.d1); (void)hipFree(s.d2); (void)hipFree(s.ws); (void)hipFree(s.d_scores); (void)hipFree(s.d_len);
            s.d1 = s.d2 = nullptr; s.ws = nullptr; s.d_scores = nullptr; s.d_len = nullptr; s.alignments = 0;
            e = hipMalloc(&s.d1, chunk * kLen);
            if (e == hipSuccess) e = hipMalloc(&s.d2, chunk * kLen);
            if (e == hipSuccess) e = hipMalloc(&s.ws, swmi::semiglobal_workspace_bytes(chunk));
            if (e == hipSuccess) e = hipMalloc(&s.d_scores, chunk * sizeof(int32_t));
            if (e == hipSuccess) e = hipMalloc(&s.d_len, chunk * sizeof(uint32_t));
            if (e == hipSuccess) s.alignments = chunk;
        }
        if (e == hipSuccess && s.tb_entries < tb_need) {
            (void)hipFree(s.d_tb);
            s.d_tb = nullptr; s.tb_entries = 0;
            e = hipMalloc(&s.d_tb, tb_need * 2 * sizeof(int32_t));
            if (e == hipSuccess) s.tb_entries = tb_need;
        }
    }
    // results of the chunk a set holds -> host; only as many positions per alignment as the longest path of the chunk has
    auto drain = [&](int which) -> hipError_t {
        Set &s = sets[which];
        hipStream_t st = streams[which];
        if (s.m == 0) return hipSuccess;
        hipError_t r = hipMemcpyAsync(scores + s.off, s.d_scores, s.m * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (r == hipSuccess) r = hipMemcpyAsync(lengths + s.off, s.d_len, s.m * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        if (r == hipSuccess && cap) {
            size_t longest = 0;
            for (size_t k = 0; k < s.m; ++k) longest = lengths[s.off + k] > longest ? lengths[s.off + k] : longest;
            if (longest > cap) longest = cap;
            const size_t pitch = cap * 2 * sizeof(int32_t);
            r = hipMemcpy2DAsync(tracebacks + s.off * cap * 2, pitch, s.d_tb, pitch, longest * 2 * sizeof(int32_t), s.m,
                                 hipMemcpyDeviceToHost, st);
            if (r == hipSuccess) r = hipStreamSynchronize(st);
        }
        s.m = 0;
        return r;
    };
    int turn = 0;
    for (size_t off = 0; e == hipSuccess && off < n; off += chunk, turn ^= 1) {
        const int which = n_sets == 2 ? turn : 0;
        Set &s = sets[which];
        hipStream_t st = streams[which];
        e = drain(which);                                   // (two chunks ago; normally already empty)
        if (e != hipSuccess) break;
        s.off = off;
        s.m = n - off < chunk ? n - off : chunk;
        e = hipMemcpyAsync(s.d1, seq1s + off * kLen, s.m * kLen, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s.d2, seq2s + off * kLen, s.m * kLen, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = swmi::launch_semiglobal(s.d1, s.d2, s.m, s.ws, s.d_scores, s.d_tb, cap, s.d_len, st);
        if (e == hipSuccess && n_sets == 2) e = drain(turn ^ 1);         // the previous chunk, while this one computes
    }
    for (int k = 0; k < n_sets; ++k) {
        if (e == hipSuccess) e = drain(k);
        if (e != hipSuccess) (void)hipStreamSynchronize(streams[k]);
        sets[k].m = 0;
    }
    if (e != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_semiglobal_xdrop: %s", hipGetErrorString(e));
    return SWMI_OK;
}

int swmi_unpack(const uint8_t *packed, size_t n_seqs, uint8_t *unpacked)
{
    if (n_seqs == 0) return SWMI_OK;
    if (!packed || !unpacked) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n_seqs = %zu", n_seqs);
    int rc = check_ready();
    if (rc != SWMI_OK) return rc;
    std::lock_guard<std::mutex> lock(g_ctx.mu);
    HIP_TRY(hipSetDevice(g_ctx.device));
    // reuse slot 0: seq1 buffer holds the packed bytes, seq2 buffer the unpacked ones
    const size_t chunk = n_seqs < kChunkPairs ? n_seqs : kChunkPairs;
    rc = ensure_slot(g_ctx.slots[0], chunk);
    if (rc != SWMI_OK) return rc;
    Slot &s = g_ctx.slots[0];
    for (size_t off = 0; off < n_seqs; off += chunk) {
        const size_t m = n_seqs - off < chunk ? n_seqs - off : chunk;
        HIP_TRY(hipMemcpyAsync(s.d_seq1, packed + off * SWMI_PACKED_LEN, m * SWMI_PACKED_LEN, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(swmi::launch_unpack(s.d_seq1, s.d_seq2, m, s.stream));
        HIP_TRY(hipMemcpyAsync(unpacked + off * kSeq, s.d_seq2, m * kSeq, hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(hipStreamSynchronize(s.stream));
    }
    return SWMI_OK;
}

int swmi_generate_pairs_device(void *d_seq1s, void *d_seq2s, size_t n, uint64_t seed, uint64_t first_pair, void *stream)
{
    if (n == 0) return SWMI_OK;
    if (!d_seq1s || !d_seq2s) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    if ((reinterpret_cast<uintptr_t>(d_seq1s) | reinterpret_cast<uintptr_t>(d_seq2s)) & 15)
        return fail(SWMI_ERR_ALIGNMENT, "device pointers must be 16-byte aligned");
    int rc = check_ready();
    if (rc != SWMI_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);   // NULL = the HIP null (default) stream, as in any HIP call
    HIP_TRY(swmi::launch_generate(static_cast<uint8_t *>(d_seq1s), static_cast<uint8_t *>(d_seq2s), n, seed, first_pair, st));
    return SWMI_OK;
}

static uint64_t host_splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int swmi_generate_pairs_host(uint8_t *seq1s, uint8_t *seq2s, size_t n, uint64_t seed, uint64_t first_pair)
{
    if (n == 0) return SWMI_OK;
    if (!seq1s || !seq2s) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    for (size_t k = 0; k < n; ++k)
        for (uint64_t s = 0; s < 2; ++s) {
            uint8_t *dst = (s ? seq2s : seq1s) + k * kSeq;
            for (uint64_t w = 0; w < 4; ++w) {
                const uint64_t ctr = ((first_pair + k) * 2 + s) * 4 + w;
                const uint64_t x = host_splitmix64(seed ^ (ctr * 0x9E3779B97F4A7C15ull));
                for (int b = 0; b < 32; ++b) dst[32 * w + b] = uint8_t((x >> (2 * b)) & 3);
            }
        }
    return SWMI_OK;
}

int swmi_time_batch_device(const void *d_seq1s, const void *d_seq2s, size_t n, const int8_t score_matrix[16],
                           int8_t gap_penalty, void *d_scores, void *stream, int iters, float *avg_ms)
{
    if (!avg_ms || iters <= 0) return fail(SWMI_ERR_INVALID_ARGUMENT, "avg_ms is NULL or iters <= 0");
    int rc = check_ready();
    if (rc != SWMI_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);   // NULL = the HIP null (default) stream, as in any HIP call
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (he == hipSuccess) he = hipEventRecord(e0, st);
    for (int it = 0; he == hipSuccess && rc == SWMI_OK && it < iters; ++it)
        rc = device_entry(d_seq1s, d_seq2s, n, score_matrix, gap_penalty, d_scores, st, false);
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (he != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_time_batch_device: %s", hipGetErrorString(he));
    if (rc != SWMI_OK) return rc;
    *avg_ms = ms / float(iters);
    return SWMI_OK;
}

// ---- queue ---------------------------------------------------------------------------------------

int swmi_queue_create(size_t max_pairs, const int8_t score_matrix[16], int8_t gap_penalty, swmi_queue **out)
{
    if (!out) return fail(SWMI_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (max_pairs == 0) return fail(SWMI_ERR_INVALID_ARGUMENT, "max_pairs is 0");
    rc = check_ready();
    if (rc != SWMI_OK) return rc;
    HIP_TRY(hipSetDevice(g_ctx.device));
    swmi_queue *q = new (std::nothrow) swmi_queue;
    if (!q) return fail(SWMI_ERR_INVALID_ARGUMENT, "out of host memory");
    q->max_pairs = max_pairs;
    memcpy(q->sm, score_matrix, 16);
    q->gap = gap_penalty;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&q->h_seq1), max_pairs * kSeq, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&q->h_seq2), max_pairs * kSeq, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&q->h_scores), max_pairs * sizeof(int32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&q->d_seq1, max_pairs * kSeq);
    if (e == hipSuccess) e = hipMalloc(&q->d_seq2, max_pairs * kSeq);
    if (e == hipSuccess) e = hipMalloc(&q->d_scores, max_pairs * sizeof(int32_t));
    if (e != hipSuccess) {
        swmi_queue_destroy(q);
        return fail(SWMI_ERR_HIP, "queue allocation failed: %s", hipGetErrorString(e));
    }
    *out = q;
    return SWMI_OK;
}

long long swmi_queue_submit(swmi_queue *q, const uint8_t seq1[SWMI_SEQ_LEN], const uint8_t seq2[SWMI_SEQ_LEN])
{
    if (!q || !seq1 || !seq2) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL argument");
    if (q->count >= q->max_pairs) return fail(SWMI_ERR_QUEUE_FULL, "queue holds %zu pairs (its capacity)", q->count);
    const size_t k = q->count++;
    memcpy(q->h_seq1 + k * kSeq, seq1, kSeq);
    memcpy(q->h_seq2 + k * kSeq, seq2, kSeq);
    if (q->count - q->shipped >= q->block) {
        const int rc = queue_ship(q, q->count);
        if (rc != SWMI_OK) return rc;
    }
    return (long long)k;
}

int swmi_queue_wait(swmi_queue *q, const int32_t **scores, size_t *n_scores)
{
    if (!q) return fail(SWMI_ERR_INVALID_ARGUMENT, "queue is NULL");
    int rc = queue_ship(q, q->count);
    if (rc != SWMI_OK) return rc;
    HIP_TRY(hipStreamSynchronize(q->stream));
    if (scores) *scores = q->h_scores;
    if (n_scores) *n_scores = q->count;
    return SWMI_OK;
}

int swmi_queue_reset(swmi_queue *q)
{
    if (!q) return fail(SWMI_ERR_INVALID_ARGUMENT, "queue is NULL");
    HIP_TRY(hipStreamSynchronize(q->stream));
    q->count = q->shipped = 0;
    return SWMI_OK;
}

int swmi_queue_destroy(swmi_queue *q)
{
    if (!q) return SWMI_OK;
    if (q->stream) (void)hipStreamSynchronize(q->stream);
    if (q->h_seq1) (void)hipHostFree(q->h_seq1);
    if (q->h_seq2) (void)hipHostFree(q->h_seq2);
    if (q->h_scores) (void)hipHostFree(q->h_scores);
    if (q->d_seq1) (void)hipFree(q->d_seq1);
    if (q->d_seq2) (void)hipFree(q->d_seq2);
    if (q->d_scores) (void)hipFree(q->d_scores);
    if (q->stream) (void)hipStreamDestroy(q->stream);
    delete q;
    return SWMI_OK;
}

}  // extern "C"
