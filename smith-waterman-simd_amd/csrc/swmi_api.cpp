// swmi_api.cpp -- host side of libswmi.so: the C ABI of include/swmi.h over the gfx950 kernels.
//
// Host responsibilities only: argument checking, score-matrix packing, chunked H2D / kernel / D2H
// pipelining for host-resident batches, the deferred queue behind the per-pair signature, and
// hipEvent timing.  There is deliberately no CPU implementation of the scoring path in this
// library: without a usable gfx950 device every scoring entry point returns an error.
#include "swmi_host.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>

namespace swmi {
namespace host {

namespace {
thread_local char t_error[512] = "";
thread_local int t_status = SWMI_OK;               // code of the last fail() on this thread
thread_local int t_gpu = 0;                        // context index this thread addresses (swmi_use_gpu)
std::vector<std::shared_ptr<Context>> g_ctxs;      // written only under g_init_mu, by swmi_init* / swmi_shutdown
std::mutex g_init_mu;
Knobs g_knobs;                                     // written only under g_init_mu (read_knobs)
bool g_knobs_read = false;
// Schedule setting, process-wide: lanes in the low half, flags in the high half, ONE atomic so that a launch that races
// with swmi_set_schedule sees either the old pair or the new one, never a mix.
std::atomic<uint64_t> g_schedule{0};
// Mapping override of the semi-global sweep (swmi_semiglobal_set_mapping; SWMI_SG_SWEEP gives the initial
// value at swmi_init): one atomic word, the mapping + 1 so that 0 = automatic.
std::atomic<uint64_t> g_sg_mapping{0};
}  // namespace

std::atomic<uint64_t> &sg_mapping_word() { return g_sg_mapping; }

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_error, sizeof t_error, fmt, ap);
    va_end(ap);
    t_status = code;
    return code;
}

int last_status() { return t_status; }

std::mutex &init_mutex() { return g_init_mu; }
int num_contexts() { return (int)g_ctxs.size(); }
Context *context_at(int index) { return index >= 0 && index < (int)g_ctxs.size() ? g_ctxs[index].get() : nullptr; }
std::shared_ptr<Context> context_ref(int index) { return index >= 0 && index < (int)g_ctxs.size() ? g_ctxs[index] : nullptr; }

int check_alive(const Context &ctx)
{
    if (!ctx.dead.load(std::memory_order_acquire)) return SWMI_OK;
    return fail(SWMI_ERR_NOT_INITIALIZED, "swmi_shutdown() ran after this handle was created: only its *_destroy call is valid now");
}

const Knobs &knobs() { return g_knobs; }

void read_knobs()
{
    Knobs k;
    auto num = [](const char *name, long long lo, long long hi, long long dflt) {
        const char *v = getenv(name);
        if (!v || !*v) return dflt;
        const long long x = atoll(v);
        return x < lo || x > hi ? dflt : x;
    };
    k.host_granule = (size_t)num("SWMI_HOST_GRANULE", 1024, (long long)kChunkPairs, 0);
    k.host_serial = num("SWMI_HOST_SERIAL", 0, 1, 0) != 0;
    k.host_taper_pct = (unsigned)num("SWMI_HOST_TAPER", 1, 99, 0);
    k.host_min_granule = (size_t)num("SWMI_HOST_MIN_GRANULE", 1024, (long long)kChunkPairs, 0);
    if (const char *sched = getenv("SWMI_HOST_SCHEDULE")) {
        int c = 0;
        for (const char *q = sched; *q && c < 16;) {
            char *end = nullptr;
            const long long v = strtoll(q, &end, 10);
            if (end == q) break;
            if (v >= 1024 && v <= (long long)kChunkPairs) k.host_schedule[c++] = (size_t)v;
            q = *end == ',' ? end + 1 : end;
        }
    }
    k.host_trace = num("SWMI_HOST_TRACE", 0, 1, 0) != 0;
    k.host_threads = (int)num("SWMI_HOST_THREADS", 1, kHostThreads, 0);
    k.host_slots = (int)num("SWMI_HOST_SLOTS", 2, kSlots / kHostThreads, 0);
    k.score_group = (size_t)num("SWMI_TEST_SCORE_GROUP", 4096, (long long)kScoreGroup, (long long)kScoreGroup);
    k.extra_lds = (unsigned)num("SWMI_EXTRA_LDS", 0, 160 * 1024, 0);
    k.lanes = (int)num("SWMI_LANES", 0, 64, 0);
    k.banded_no_i16 = getenv("SWMI_BANDED_NO_I16") != nullptr;
    k.banded_no_pk = getenv("SWMI_BANDED_NO_PK") != nullptr;
    k.sg_sweep = (int)num("SWMI_SG_SWEEP", 1, 44, -1);
    k.sg_exact = (int)num("SWMI_SG_EXACT", 0, 1, -1);
    const char *gb = getenv("SWMI_GATHER_BACKEND");
    k.gather_p2p = gb && strcmp(gb, "p2p") == 0;
    k.gather_piece = (size_t)num("SWMI_TEST_GATHER_PIECE", 1, 1ll << 40, 0);
    if (const char *lib = getenv("SWMI_RCCL_LIB")) snprintf(k.rccl_lib, sizeof k.rccl_lib, "%s", lib);
    g_knobs = k;
    g_knobs_read = true;
}

Context *current()
{
    if (g_ctxs.empty()) {
        fail(SWMI_ERR_NOT_INITIALIZED, "swmi_init() has not been called (or failed)");
        return nullptr;
    }
    Context *ctx = context_at(t_gpu);
    if (!ctx) {
        fail(SWMI_ERR_INVALID_ARGUMENT, "this thread selected GPU index %d but only %d are bound", t_gpu, (int)g_ctxs.size());
        return nullptr;
    }
    const hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) {
        fail(SWMI_ERR_HIP, "hipSetDevice(%d) failed: %s", ctx->device, hipGetErrorString(e));
        return nullptr;
    }
    return ctx;
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

namespace {
// Lanes per alignment when the caller has not fixed a schedule: L = 4 (32 rows per lane) issues the fewest instructions
// per cell and wins once the batch fills the chip; a small batch wants many lanes per alignment instead -- a single pair
// takes 6 us with L = 64 and 62 us with the packed L = 4 kernel.  L = 16, 8 and 4 run the packed kernel (8 / 16 / 32
// alignments per wavefront); the thresholds are where their measured times cross (tools/small_batch_schedule.py,
// profiles/r02_small_batch_schedule.txt).
int auto_lanes(size_t n)
{
    return n <= 2048 ? 64 : n <= 5120 ? 32 : n <= 24576 ? 16 : n <= 98304 ? 8 : 4;
}
int resolve_lanes(uint64_t schedule, size_t n)
{
    const int lanes = int(schedule & 0xffffffffu);
    const unsigned flags = unsigned(schedule >> 32);
    return lanes ? lanes : (flags & swmi::kUseLut) ? 4 : auto_lanes(n);
}
}  // namespace

// Choose the schedule and the cell body: the gap-folded recurrence needs every sm + gap to fit int8.
LaunchConfig make_config(const Context &ctx, const int8_t *sm, int gap, SmRows *rows, size_t n)
{
    const uint64_t schedule = g_schedule.load(std::memory_order_relaxed);     // one snapshot per launch
    const unsigned flags = unsigned(schedule >> 32);
    LaunchConfig cfg;
    cfg.lanes_per_alignment = resolve_lanes(schedule, n);
    cfg.use_i16 = (flags & swmi::kUseI16) != 0;
    cfg.use_lut = (flags & swmi::kUseLut) != 0;
    cfg.extra_lds_bytes = ctx.extra_lds;
    cfg.pk_bias = 0;
    cfg.pk_variant = 0;
    bool fold = !(flags & swmi::kNoGapFold);
    for (int k = 0; k < 16 && fold; ++k) {
        const int v = int(sm[k]) + gap;
        if (v < -128 || v > 127) fold = false;
    }
    cfg.fold_gap = fold;
    // L = 4, 8 or 16 with no A/B variant asked for runs the packed kernel (two alignments per register, L lanes per PAIR of
    // alignments): its rows hold
    // s + gap + Q as unsigned bytes, Q = max(0, -(min s + gap)), which fits one byte for every int8 matrix and gap <= 127
    cfg.use_pk = (cfg.lanes_per_alignment == 4 || cfg.lanes_per_alignment == 8 || cfg.lanes_per_alignment == 16) &&
                 !(flags & (swmi::kNoPacked | swmi::kNoGapFold | swmi::kUseI16 | swmi::kUseLut));
    cfg.pk_bias = 0;
    cfg.pk_variant = 0;
    if (cfg.use_pk) {
        int lowest = 255, highest = -255;
        for (int k = 0; k < 16; ++k) {
            lowest = int(sm[k]) < lowest ? int(sm[k]) : lowest;
            highest = int(sm[k]) > highest ? int(sm[k]) : highest;
        }
        // Which cell body (sw_kernels.hip PkVariant; tests/test_gpu_pk_kernels.py restates the rule):
        //   0  every s + gap >= 0: nothing to bias
        //   2  every s + 2 gap in [0, 255] and the largest value of the vertical-offset form, 128 max(s) + gap * 34 + 255,
        //      stays a finite half-precision pattern (< 0x7C00): rows carry s + 2 gap
        //   1  anything else: values shifted by Q = -(min s + gap)
        int add = gap;
        if (lowest + gap >= 0) {
            cfg.pk_variant = 0;
        } else if (lowest + 2 * gap >= 0 && highest + 2 * gap <= 255 &&
                   128 * (highest > 0 ? highest : 0) + 34 * gap + 256 < 0x7C00) {
            cfg.pk_variant = 2;
            add = 2 * gap;
        } else {
            cfg.pk_variant = 1;
            cfg.pk_bias = -(lowest + gap);
            add = gap + cfg.pk_bias;
        }
        for (int a = 0; a < 4; ++a) {
            uint32_t r = 0;
            for (int b = 0; b < 4; ++b) r |= uint32_t(int(sm[4 * a + b]) + add) << (8 * b);
            rows->r[a] = r;
        }
        return cfg;
    }
    *rows = pack_rows(sm, fold ? gap : 0);
    return cfg;
}

namespace {
// bytes: what each of the slot's two input buffers must hold; scores: entries of the slot's own score buffer (0 = none)
int ensure_slot(Slot &s, size_t bytes, size_t scores)
{
    if (s.capacity < bytes) {
        if (s.d_seq1) { (void)hipFree(s.d_seq1); s.d_seq1 = nullptr; }
        if (s.d_seq2) { (void)hipFree(s.d_seq2); s.d_seq2 = nullptr; }
        s.capacity = 0;
        SWMI_HIP_TRY(hipMalloc(&s.d_seq1, bytes));
        SWMI_HIP_TRY(hipMalloc(&s.d_seq2, bytes));
        s.capacity = bytes;
    }
    if (s.score_capacity < scores) {
        if (s.d_scores) { (void)hipFree(s.d_scores); s.d_scores = nullptr; }
        s.score_capacity = 0;
        SWMI_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s.d_scores), scores * sizeof(int32_t)));
        s.score_capacity = scores;
    }
    return SWMI_OK;
}

}  // namespace

int launch_device(Context &ctx, const void *d1, const void *d2, size_t n, const int8_t *sm, int gap, void *d_out,
                  hipStream_t st, bool packed)
{
    SmRows rows;
    const LaunchConfig cfg = make_config(ctx, sm, gap, &rows, n);
    const size_t stride = packed ? SWMI_PACKED_LEN : kSeq;
    for (size_t off = 0; off < n; off += kMaxLaunchPairs) {
        const size_t m = n - off < kMaxLaunchPairs ? n - off : kMaxLaunchPairs;
        SWMI_HIP_TRY(swmi::launch_score(cfg, static_cast<const uint8_t *>(d1) + off * stride,
                                        static_cast<const uint8_t *>(d2) + off * stride,
                                        static_cast<int32_t *>(d_out) + off, m, rows, gap, packed, st));
    }
    return SWMI_OK;
}

// Pipeline granules of a host batch.  Copies run back to back on the link (c seconds per pair), kernels back to back on the
// GPU (k seconds per pair); granule i's kernel can start when its copy has landed, so the batch ends at n c + (last
// granule) k PROVIDED no kernel is still running when the next granule's copy lands: g[i+1] c >= g[i] k.  The schedule
// therefore TAPERS geometrically with the ratio f = k / c (plus a margin): each granule is (1 - f) of what is left, at most
// kChunkPairs, and whatever is left below the smallest granule goes as one.  k is ~1.07 ns per pair for every entry, c is
// the entry's bytes per pair over the ~53.6 GB/s a PCIe Gen5 x16 link delivers from pageable memory, so f = 59 / bytes
// per pair; with the margin, 64 / bytes:
//   pairs        256 B  f = 1/4   1M pairs go as 768 K, 192 K, 48 K, 16 K (the kernel is ~4x faster than its granule's copy)
//   one-vs-many  128 B  f = 1/2   512 K, 256 K, 128 K, 64 K, 32 K, 16 K, 16 K
//   2-bit packed  64 B  f -> 1    copy and kernel take the same time: (near-)equal granules, the end exposed is one
//                                  granule's kernel.  Round 3 ran this entry on the 256-byte taper: its 768 K-pair kernel
//                                  (0.8 ms) then ran after a 0.9 ms copy with only 0.3 ms of copies left to hide under.
// SWMI_HOST_TAPER / SWMI_HOST_MIN_GRANULE / SWMI_HOST_GRANULE override f (percent), the smallest and a fixed granule
// (tools/host_pipeline_experiment.py).
size_t host_entry_bytes(int entry)
{
    return entry == kEntryPacked ? 2 * SWMI_PACKED_LEN : entry == kEntryOneVsMany ? kSeq : 2 * kSeq;
}

size_t next_granule(size_t remaining, size_t bytes_per_pair)
{
    const Knobs &kn = knobs();
    if (kn.host_granule) return remaining < kn.host_granule ? remaining : kn.host_granule;
    const size_t smallest = kn.host_min_granule ? kn.host_min_granule : kMinGranule;
    if (remaining <= smallest) return remaining;
    size_t f1024 = kn.host_taper_pct ? size_t(kn.host_taper_pct) * 1024 / 100 : 65536 / bytes_per_pair;
    if (f1024 > kMaxTaper1024) f1024 = kMaxTaper1024;
    size_t g = (remaining - remaining * f1024 / 1024) & ~size_t(4095);
    if (g > kChunkPairs) g = kChunkPairs;
    if (g < smallest) g = smallest;
    return g;
}

namespace {
int ensure_scores_all(Context &ctx, size_t pairs)
{
    if (ctx.scores_all_capacity >= pairs) return SWMI_OK;
    if (ctx.d_scores_all) { (void)hipFree(ctx.d_scores_all); ctx.d_scores_all = nullptr; }
    ctx.scores_all_capacity = 0;
    SWMI_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ctx.d_scores_all), pairs * sizeof(int32_t)));
    ctx.scores_all_capacity = pairs;
    return SWMI_OK;
}
}  // namespace

// Host-resident batch (the body of swmi_score_batch and its relatives).  kSlots input buffer sets, a stream each;
// granule k: H2D of its two arrays, then its kernel, on one slot's stream, writing into ONE device score vector for the
// whole group of granules.  Nothing is copied back until every granule of the group has been issued: a copy into pageable
// memory blocks the calling thread until the stream reaches it, and round 2's pipeline -- D2H behind every granule --
// therefore never had granule k + 1's H2D in flight while granule k's kernel ran (profiles/r02_host_staging_experiment.txt:
// 4 x (4.9 + 1.5) ms for 4M pairs).  One D2H per group (up to kScoreGroup pairs, 64 MiB of scores) follows the group's last
// kernel.
// TWO host threads issue the granules, even ones on the calling thread, odd ones on the context's persistent helper
// (round 4): a copy command from pageable memory costs the issuing thread ~20 us of set-up and tear-down around its DMA
// (the runtime pins and unpins the pages) during which the link idles -- 8 commands per 1M packed pairs, 0.2 ms of 1.75 --
// and with two threads one command's DMA runs while the other thread prepares the next.  Each thread rotates over its own
// two slots, so buffer reuse stays ordered by the slot's stream.
namespace {
struct Granule { size_t off, m; };                   // inside its group

// SWMI_HOST_TRACE=1 (diagnostics only): three timing events per granule -- before its copies, behind them, behind its kernel
struct TraceRow { size_t index, pairs; int thread; hipEvent_t ev[3]; double host_issue_ms[2]; };
std::mutex g_trace_mu;
std::vector<TraceRow> g_trace;
double host_now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// issue granules first, first + step, ... of `list` (one issuing thread's share); slots[] = this thread's buffer sets
// early: granules with an index below it belong to the part of the group whose scores go back to the host while the last
// granules still run (0 = none); behind every such granule the slot's "early" event is recorded again, so that it ends up
// behind the slot's last early kernel, and early_done is set once this thread has issued its last early granule
hipError_t issue_granules(Context &ctx, const std::vector<Granule> &list, size_t first, size_t step, Slot *const *slots, int n_slots,
                          const uint8_t *s1, const uint8_t *s2, size_t group, size_t in_stride, const int8_t *sm, int gap,
                          int32_t *out, bool packed, bool one_vs_many, bool serial, bool *slot_used, size_t early = 0,
                          bool *slot_early_used = nullptr, std::atomic<bool> *early_done = nullptr)
{
    hipError_t e = hipSuccess;
    size_t j = 0;
    for (size_t i = first; i < list.size() && e == hipSuccess; i += step, ++j) {
        const size_t off = list[i].off, m = list[i].m, at = group + off;
        const int which = int(j % size_t(n_slots));
        Slot &s = *slots[which];
        SmRows rows;
        const LaunchConfig cfg = make_config(ctx, sm, gap, &rows, m);      // a small tail granule runs more lanes per alignment
        TraceRow tr{i, m, int(first), {nullptr, nullptr, nullptr}, {0, 0}};
        const bool trace = knobs().host_trace;
        if (trace) {
            for (auto &ev : tr.ev) (void)hipEventCreate(&ev);
            tr.host_issue_ms[0] = host_now_ms();
            (void)hipEventRecord(tr.ev[0], s.stream);
        }
        e = hipMemcpyAsync(s.d_seq1, s1 + at * in_stride, m * in_stride, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess && !one_vs_many)
            e = hipMemcpyAsync(s.d_seq2, s2 + at * in_stride, m * in_stride, hipMemcpyHostToDevice, s.stream);
        if (trace) (void)hipEventRecord(tr.ev[1], s.stream);
        if (e == hipSuccess)
            e = one_vs_many ? swmi::launch_score_one_vs_many(cfg, s.d_seq1, s.d_seq2, ctx.d_scores_all + off, m, rows, gap, s.stream)
                            : swmi::launch_score(cfg, s.d_seq1, s.d_seq2, ctx.d_scores_all + off, m, rows, gap, packed, s.stream);
        if (trace) {
            (void)hipEventRecord(tr.ev[2], s.stream);
            tr.host_issue_ms[1] = host_now_ms();
            std::lock_guard<std::mutex> l(g_trace_mu);
            g_trace.push_back(tr);
        }
        if (e == hipSuccess && serial)
            e = hipMemcpyAsync(out + at, ctx.d_scores_all + off, m * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream);
        slot_used[which] = true;
        if (i < early && e == hipSuccess) {
            e = hipEventRecord(ctx.slot_early[&s - ctx.slots], s.stream);
            slot_early_used[which] = true;
        }
        if (early_done && i + step >= early) early_done->store(true, std::memory_order_release);     // no early granule of this thread follows
    }
    if (early_done) early_done->store(true, std::memory_order_release);
    return e;
}
}  // namespace

// The granules of one score group.  Where the link is the bound (256 / 128 bytes per pair) the tapered schedule of
// next_granule.  Where copy and kernel are LEVEL (the 2-bit packed entry, 64 bytes per pair: f would exceed 15/16) a taper
// cannot work -- the kernel never gains on the copies -- and the batch ends at about
//     (first granule's copy) + max(all copies, all kernels + a launch ramp of ~13 us per granule) + (last granule's kernel)
// + the score copy: equal granules of 128 K pairs (8 per 1M pairs: below that the launch ramps and the ~20 us a copy
// command costs its issuing thread add up, above it the two ends grow) between a short first and last one (32 K, 96 K).
// Measured against equal granules of 32 K .. 512 K, tapers of 50 .. 94 % and other ramp shapes, one and two issuing threads,
// two and three buffer sets per thread: profiles/r04_host_pipeline_experiment.txt.
void granule_list(size_t group_n, size_t bytes_per_pair, std::vector<size_t> *out)
{
    out->clear();
    const Knobs &kn = knobs();
    if (kn.host_schedule[0]) {                         // explicit list (experiments): the last entry repeats
        size_t k = 0;
        for (size_t rem = group_n; rem;) {
            const size_t want = kn.host_schedule[k], g = rem < want ? rem : want;
            out->push_back(g);
            rem -= g;
            if (k + 1 < 16 && kn.host_schedule[k + 1]) ++k;
        }
        return;
    }
    const bool balanced = bytes_per_pair * 15 <= 1024;
    if (balanced && !kn.host_granule && !kn.host_taper_pct) {
        const size_t steady = kn.host_min_granule ? kn.host_min_granule : kBalancedGranule;
        const size_t ramp[2] = {steady / 4, steady - steady / 4};     // 32 K, 96 K: together one steady granule
        if (group_n < 4 * steady) {                             // too short for the shape: equal granules of half the size
            for (size_t rem = group_n; rem;) {
                const size_t g = rem < steady / 2 ? rem : steady / 2;
                out->push_back(g);
                rem -= g;
            }
            return;
        }
        out->push_back(ramp[0]);
        out->push_back(ramp[1]);
        const size_t middle = group_n - 2 * steady;
        const size_t count = (middle + steady - 1) / steady;
        for (size_t k = 0, done = 0; k < count; ++k) {          // near-equal: cut at multiples of 4096, the last takes the rest
            const size_t upto = k + 1 == count ? middle : (middle * (k + 1) / count) & ~size_t(4095);
            out->push_back(upto - done);
            done = upto;
        }
        out->push_back(ramp[1]);
        out->push_back(ramp[0]);
        return;
    }
    for (size_t rem = group_n; rem;) {
        const size_t g = next_granule(rem, bytes_per_pair);
        out->push_back(g);
        rem -= g;
    }
}

int score_host_batch(Context &ctx, const uint8_t *s1, const uint8_t *s2, size_t n, const int8_t *sm, int gap, int32_t *out,
                     bool packed, bool one_vs_many)
{
    std::lock_guard<std::mutex> lock(ctx.mu);
    const size_t in_stride = packed ? SWMI_PACKED_LEN : kSeq;
    const size_t per_pair = host_entry_bytes(packed ? kEntryPacked : one_vs_many ? kEntryOneVsMany : kEntryPairs);
    const bool serial = knobs().host_serial;             // round 2's order of issue, for the A/B
    const size_t score_group = knobs().score_group;
    const size_t first_group = n < score_group ? n : score_group;
    std::vector<size_t> sizes;
    granule_list(first_group, per_pair, &sizes);                      // no later group is larger, so none has a larger granule
    size_t largest = 0;
    for (size_t g : sizes) largest = g > largest ? g : largest;
    const size_t first_count = sizes.size();
    // issuing threads: two where copy and kernel are level (the 2-bit packed entry) and there is more than one granule; where
    // the link alone is the bound (256 / 128 bytes per pair) a second thread only makes the two threads' copies and kernels
    // compete (measured: profiles/r04_host_pipeline_experiment.txt)
    const bool balanced = per_pair * 15 <= 1024;
    const int want_threads = knobs().host_threads ? knobs().host_threads : balanced ? 2 : 1;
    const int threads = (serial || want_threads < 2 || first_count < 2) ? 1 : 2;
    // buffer sets: two per thread when two threads issue (three measured slower: copies that run further ahead compete with the
    // kernels), three for a lone thread (round 3's pipeline)
    const int per_thread = knobs().host_slots ? knobs().host_slots : threads == 2 ? 2 : 3;
    const int used_slots = threads == 2 ? 2 * per_thread : n > score_group || first_count > size_t(per_thread) ? per_thread : int(first_count);
    for (int k = 0; k < used_slots; ++k) {
        const int rc = ensure_slot(ctx.slots[k], largest * in_stride, 0);
        if (rc != SWMI_OK) return rc;
    }
    {
        const int rc = ensure_scores_all(ctx, first_group);
        if (rc != SWMI_OK) return rc;
    }
    if (threads == 2 && !ctx.copier) ctx.copier.reset(new Worker);
    hipError_t e = hipSuccess;
    if (one_vs_many)   // the single seq2 goes to the head of the seq2 buffer of every slot in use
        for (int k = 0; k < used_slots && e == hipSuccess; ++k)
            e = hipMemcpyAsync(ctx.slots[k].d_seq2, s2, kSeq, hipMemcpyHostToDevice, ctx.slots[k].stream);
    // thread t's slots: t, t + threads, ... (one thread: all of them in turn)
    Slot *mine[kSlots], *theirs[kSlots];
    int n_mine = 0, n_theirs = 0;
    for (int k = 0; k < used_slots; ++k) {
        if (threads == 1 || k % 2 == 0) mine[n_mine++] = &ctx.slots[k];
        else                            theirs[n_theirs++] = &ctx.slots[k];
    }
    std::vector<Granule> list;
    for (size_t group = 0; group < n && e == hipSuccess; group += score_group) {
        const size_t group_n = n - group < score_group ? n - group : score_group;
        list.clear();
        granule_list(group_n, per_pair, &sizes);
        {
            size_t off = 0;
            for (size_t m : sizes) {
                list.push_back(Granule{off, m});
                off += m;
            }
        }
        bool used_mine[kSlots] = {}, used_theirs[kSlots] = {};
        hipError_t e2 = hipSuccess;
        // With two issuing threads the group's scores return in TWO copies: the early part -- everything but the last two
        // granules -- from the helper, as soon as it has issued its share, on the context's own stream and under the last
        // kernels; the rest from this thread behind the last kernel.  (One copy behind everything left ~80 us of link time
        // exposed at the end of a 1M-pair packed batch.)
        const size_t early = threads == 2 && list.size() >= 4 ? list.size() - 2 : 0;
        const size_t early_pairs = early ? list[early].off : 0;
        if (threads == 2 && list.size() > 1) {
            const int device = ctx.device;
            bool early_mine[kSlots] = {}, early_theirs[kSlots] = {};
            std::atomic<bool> mine_done{false};
            ctx.copier->submit([&, device] {
                e2 = hipSetDevice(device);                  // per host thread, like every HIP "current device"
                if (e2 == hipSuccess)
                    e2 = issue_granules(ctx, list, 1, 2, theirs, n_theirs, s1, s2, group, in_stride, sm, gap, out, packed, one_vs_many,
                                        serial, used_theirs, early, early_theirs);
                if (e2 != hipSuccess || early == 0) return;
                while (!mine_done.load(std::memory_order_acquire)) std::this_thread::yield();     // (its early events are recorded)
                for (int k = 0; k < n_mine && e2 == hipSuccess; ++k)
                    if (early_mine[k]) e2 = hipStreamWaitEvent(ctx.stream, ctx.slot_early[mine[k] - ctx.slots], 0);
                for (int k = 0; k < n_theirs && e2 == hipSuccess; ++k)
                    if (early_theirs[k]) e2 = hipStreamWaitEvent(ctx.stream, ctx.slot_early[theirs[k] - ctx.slots], 0);
                if (e2 == hipSuccess)
                    e2 = hipMemcpyAsync(out + group, ctx.d_scores_all, early_pairs * sizeof(int32_t), hipMemcpyDeviceToHost, ctx.stream);
                const hipError_t es = hipStreamSynchronize(ctx.stream);     // (a pinned destination does not block the copy call)
                if (e2 == hipSuccess) e2 = es;
            });
            e = issue_granules(ctx, list, 0, 2, mine, n_mine, s1, s2, group, in_stride, sm, gap, out, packed, one_vs_many, serial, used_mine,
                               early, early_mine, &mine_done);
            mine_done.store(true, std::memory_order_release);       // (also after a failure: the helper must not wait for ever)
            ctx.copier->wait();                             // (it has ISSUED its share and copied the early scores)
            if (e == hipSuccess) e = e2;
        } else {
            e = issue_granules(ctx, list, 0, 1, mine, n_mine, s1, s2, group, in_stride, sm, gap, out, packed, one_vs_many, serial, used_mine);
        }
        if (e != hipSuccess) break;
        // the slot that took the group's LAST granule carries the score copy
        const size_t last = list.size() - 1;
        const bool last_is_mine = threads == 1 || list.size() == 1 || last % 2 == 0;
        const size_t last_j = threads == 1 || list.size() == 1 ? last : last / 2;
        Slot *last_slot = last_is_mine ? mine[last_j % size_t(n_mine)] : theirs[last_j % size_t(n_theirs)];
        const bool more = group + group_n < n;
        if (serial) {
            // every granule copied its own scores back on its own stream; the next group's kernels overwrite d_scores_all, so
            // every stream that still holds such a copy has to drain first (a copy into pinned memory does not block the caller)
            for (int k = 0; k < n_mine && more && e == hipSuccess; ++k)
                if (used_mine[k]) e = hipStreamSynchronize(mine[k]->stream);
            continue;
        }
        // the group's (remaining) scores: one copy on the last granule's stream, behind every other slot's last kernel
        auto join = [&](Slot *s) {
            if (s == last_slot || e != hipSuccess) return;
            hipEvent_t ev = ctx.slot_done[s - ctx.slots];
            e = hipEventRecord(ev, s->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(last_slot->stream, ev, 0);
        };
        for (int k = 0; k < n_mine; ++k)
            if (used_mine[k]) join(mine[k]);
        for (int k = 0; k < n_theirs; ++k)
            if (used_theirs[k]) join(theirs[k]);
        if (e == hipSuccess)
            e = hipMemcpyAsync(out + group + early_pairs, ctx.d_scores_all + early_pairs, (group_n - early_pairs) * sizeof(int32_t),
                               hipMemcpyDeviceToHost, last_slot->stream);
        // the next group overwrites d_scores_all: drain this one first (only batches above the score group get here twice)
        if (e == hipSuccess && more) e = hipStreamSynchronize(last_slot->stream);
    }
    for (int k = 0; k < kSlots; ++k)
        if (ctx.slots[k].stream) {
            const hipError_t es = hipStreamSynchronize(ctx.slots[k].stream);      // on failure too: copies in flight use the caller's buffers
            if (e == hipSuccess) e = es;
        }
    if (knobs().host_trace) {                           // (every stream has drained: the events are complete)
        std::lock_guard<std::mutex> l(g_trace_mu);
        const double t_end = host_now_ms();
        hipEvent_t ref = nullptr;
        double host0 = 0;
        for (auto &r : g_trace)
            if (r.index == 0) { ref = r.ev[0]; host0 = r.host_issue_ms[0]; }
        fprintf(stderr, "swmi host batch, %zu pairs, %zu B per pair: granule, pairs, issuing thread | GPU: copies start .. end, kernel end | host: issue start .. end (ms)\n",
                n, per_pair);
        for (auto &r : g_trace) {
            float t[3] = {0, 0, 0};
            for (int k = 0; k < 3; ++k)
                if (ref) (void)hipEventElapsedTime(&t[k], ref, r.ev[k]);
            fprintf(stderr, "  %3zu %8zu  t%d | %7.3f .. %7.3f  %7.3f | %7.3f .. %7.3f\n", r.index, r.pairs, r.thread, t[0], t[1], t[2],
                    r.host_issue_ms[0] - host0, r.host_issue_ms[1] - host0);
        }
        for (auto &r : g_trace)
            for (auto &ev : r.ev) (void)hipEventDestroy(ev);
        fprintf(stderr, "  call returns at %.3f ms\n", t_end - host0);
        g_trace.clear();
    }
    if (e != hipSuccess) return fail(SWMI_ERR_HIP, "host batch on GPU %d failed: %s", ctx.device, hipGetErrorString(e));
    return SWMI_OK;
}

}  // namespace host
}  // namespace swmi

using namespace swmi::host;
using swmi::LaunchConfig;
using swmi::SmRows;
#define HIP_TRY SWMI_HIP_TRY

// ---- deferred queue ------------------------------------------------------------------------------

struct swmi_queue {
    std::shared_ptr<Context> ctx;       // the GPU the queue was created on (kept alive past swmi_shutdown, see Context::dead)
    int device = -1;                    // its HIP ordinal: all that swmi_queue_destroy needs
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
    const int alive = check_alive(*q->ctx);
    if (alive != SWMI_OK) return alive;
    HIP_TRY(hipSetDevice(q->device));
    const size_t off = q->shipped, m = upto - q->shipped;
    HIP_TRY(hipMemcpyAsync(q->d_seq1 + off * kSeq, q->h_seq1 + off * kSeq, m * kSeq, hipMemcpyHostToDevice, q->stream));
    HIP_TRY(hipMemcpyAsync(q->d_seq2 + off * kSeq, q->h_seq2 + off * kSeq, m * kSeq, hipMemcpyHostToDevice, q->stream));
    const int rc = launch_device(*q->ctx, q->d_seq1 + off * kSeq, q->d_seq2 + off * kSeq, m, q->sm, q->gap, q->d_scores + off,
                                 q->stream, false);
    if (rc != SWMI_OK) return rc;
    HIP_TRY(hipMemcpyAsync(q->h_scores + off, q->d_scores + off, m * sizeof(int32_t), hipMemcpyDeviceToHost, q->stream));
    q->shipped = upto;
    return SWMI_OK;
}

void destroy_context(Context &c)
{
    c.copier.reset();                   // joins the helper thread (idle: host batches hold c.mu while it works)
    (void)hipSetDevice(c.device);
    (void)hipDeviceSynchronize();
    for (auto &s : c.slots) {
        if (s.d_seq1) (void)hipFree(s.d_seq1);
        if (s.d_seq2) (void)hipFree(s.d_seq2);
        if (s.d_scores) (void)hipFree(s.d_scores);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        s = Slot{};
    }
    if (c.stream) (void)hipStreamDestroy(c.stream);
    c.stream = nullptr;
    if (c.d_scores_all) (void)hipFree(c.d_scores_all);
    c.d_scores_all = nullptr;
    c.scores_all_capacity = 0;
    for (auto &ev : c.slot_done) {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
    }
    for (auto &ev : c.slot_early) {
        if (ev) (void)hipEventDestroy(ev);
        ev = nullptr;
    }
    for (auto &w : c.sg_workspaces)
        if (w.second.ptr) (void)hipFree(w.second.ptr);
    c.sg_workspaces.clear();
    for (auto &g : c.sg_sets) {
        (void)hipFree(g.d1); (void)hipFree(g.d2); (void)hipFree(g.ws); (void)hipFree(g.d_scores); (void)hipFree(g.d_len); (void)hipFree(g.d_tb);
        (void)hipFree(g.d_moves);
        g = SgSet{};
    }
    if (c.pin) (void)hipHostFree(c.pin);
    c.pin = nullptr;
    c.pin_dev = nullptr;
}

int create_context(Context &c, int index, int device)
{
    c.index = index;
    c.device = device;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipGetDeviceProperties(&c.prop, device));
    if (strncmp(c.prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SWMI_ERR_UNSUPPORTED_ARCH, "device %d is %s; libswmi carries gfx950 (MI355X) code objects only", device,
                    c.prop.gcnArchName);
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    for (auto &s : c.slots) HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    for (auto &ev : c.slot_done) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto &ev : c.slot_early) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&c.pin), kPinPairs * (2 * kSeq + sizeof(int32_t)), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer(&c.pin_dev, c.pin, 0));
    c.extra_lds = knobs().extra_lds;
    return SWMI_OK;
}

// Bind the listed devices (under g_init_mu).  All or nothing.
int init_list(const int *devices, int n)
{
    read_knobs();                       // the SWMI_* environment, once per initialisation
    std::vector<std::shared_ptr<Context>> fresh;
    for (int k = 0; k < n; ++k) {
        fresh.emplace_back(std::make_shared<Context>());
        const int rc = create_context(*fresh.back(), k, devices[k]);
        if (rc != SWMI_OK) {
            for (auto &c : fresh) destroy_context(*c);
            return rc;
        }
    }
    // peer access between distinct bound devices: lets the score gather go GPU to GPU over xGMI (swmi_multi.cpp); where the
    // platform refuses, hipMemcpyPeerAsync still works (staged by the runtime), so failures are not errors
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            if (devices[a] != devices[b]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[a]);
                    (void)hipDeviceEnablePeerAccess(devices[b], 0);
                    (void)hipGetLastError();            // "already enabled" is fine
                }
            }
    (void)hipSetDevice(devices[0]);
    g_ctxs = std::move(fresh);
    if (knobs().lanes && swmi::schedule_supported(knobs().lanes)) g_schedule.store(uint64_t(knobs().lanes));
    // like SWMI_LANES above: the environment gives an initial value only when it is set; a mapping the program chose
    // through swmi_semiglobal_set_mapping survives a shutdown / re-init
    if (knobs().sg_sweep >= 0 && swmi_semiglobal_set_mapping(knobs().sg_sweep) != SWMI_OK) (void)swmi_semiglobal_set_mapping(-1);
    if (knobs().sg_exact >= 0) (void)swmi_semiglobal_set_exact(knobs().sg_exact ? 1 : 0);
    return SWMI_OK;
}

int device_count_or_fail(int *count)
{
    *count = 0;
    const hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess || *count <= 0)
        return fail(SWMI_ERR_NO_DEVICE, "no HIP device available (%s); libswmi has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
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
    int rc = device_count_or_fail(&count);
    if (rc != SWMI_OK) return rc;
    if (device < 0) {
        const char *lr = getenv("LOCAL_RANK");
        device = lr ? atoi(lr) % count : 0;
    }
    if (device >= count) return fail(SWMI_ERR_INVALID_ARGUMENT, "device %d out of range (count %d)", device, count);
    if (!g_ctxs.empty()) {
        if (g_ctxs.size() == 1 && g_ctxs[0]->device == device) return SWMI_OK;
        return fail(SWMI_ERR_INVALID_ARGUMENT, "already initialised (%zu GPU(s), first is device %d); call swmi_shutdown() first",
                    g_ctxs.size(), g_ctxs[0]->device);
    }
    return init_list(&device, 1);
}

int swmi_init_devices(const int *devices, int n)
{
    std::lock_guard<std::mutex> lock(g_init_mu);
    t_error[0] = 0;
    if (!devices || n <= 0 || n > 64) return fail(SWMI_ERR_INVALID_ARGUMENT, "device list is NULL or its length %d is outside [1, 64]", n);
    int count = 0;
    int rc = device_count_or_fail(&count);
    if (rc != SWMI_OK) return rc;
    for (int k = 0; k < n; ++k)
        if (devices[k] < 0 || devices[k] >= count)
            return fail(SWMI_ERR_INVALID_ARGUMENT, "device %d (entry %d) out of range (count %d)", devices[k], k, count);
    if (!g_ctxs.empty()) {
        bool same = (int)g_ctxs.size() == n;
        for (int k = 0; same && k < n; ++k) same = g_ctxs[k]->device == devices[k];
        if (same) return n;
        return fail(SWMI_ERR_INVALID_ARGUMENT, "already initialised with a different device list; call swmi_shutdown() first");
    }
    rc = init_list(devices, n);
    return rc == SWMI_OK ? n : rc;
}

int swmi_init_all(int n_gpus)
{
    int count = 0;
    {
        std::lock_guard<std::mutex> lock(g_init_mu);
        t_error[0] = 0;
        const int rc = device_count_or_fail(&count);
        if (rc != SWMI_OK) return rc;
    }
    if (n_gpus <= 0) n_gpus = count;
    if (n_gpus > count) return fail(SWMI_ERR_INVALID_ARGUMENT, "%d GPUs requested, %d visible", n_gpus, count);
    std::vector<int> devices(n_gpus);
    for (int k = 0; k < n_gpus; ++k) devices[k] = k;
    return swmi_init_devices(devices.data(), n_gpus);
}

int swmi_num_gpus(void) { return num_contexts(); }

int swmi_use_gpu(int index)
{
    if (index < 0 || (!g_ctxs.empty() && index >= (int)g_ctxs.size()))
        return fail(SWMI_ERR_INVALID_ARGUMENT, "GPU index %d out of range (%d bound)", index, (int)g_ctxs.size());
    t_gpu = index;
    return SWMI_OK;
}

int swmi_shutdown(void)
{
    stop_workers();                     // before the contexts go: a worker holds its context's device current
    std::lock_guard<std::mutex> lock(g_init_mu);
    for (auto &c : g_ctxs) {
        destroy_context(*c);
        c->dead.store(true, std::memory_order_release);     // handles that outlive the shutdown find this, not freed memory
    }
    g_ctxs.clear();
    t_gpu = 0;
    return SWMI_OK;
}

int swmi_set_schedule(int lanes_per_alignment, unsigned flags)
{
    if (lanes_per_alignment != 0 && !swmi::schedule_supported(lanes_per_alignment))
        return fail(SWMI_ERR_INVALID_ARGUMENT, "lanes_per_alignment must be one of 64,32,16,8,4,2 (got %d)", lanes_per_alignment);
    if (flags & ~15u) return fail(SWMI_ERR_INVALID_ARGUMENT, "unknown schedule flags 0x%x", flags);
    g_schedule.store(uint64_t(uint32_t(lanes_per_alignment)) | (uint64_t(flags) << 32));
    return SWMI_OK;
}

int swmi_get_schedule(int *lanes_per_alignment, unsigned *flags)
{
    const uint64_t s = g_schedule.load();
    if (lanes_per_alignment) *lanes_per_alignment = int(s & 0xffffffffu);      // 0 = automatic
    if (flags) *flags = unsigned(s >> 32);
    return SWMI_OK;
}

int swmi_schedule_for_batch(size_t n) { return resolve_lanes(g_schedule.load(), n); }

int swmi_score_kernel_for_batch(size_t n, const int8_t score_matrix[16], int8_t gap_penalty, int mode, char *name,
                                size_t name_len, int *alignments_per_wavefront)
{
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (mode < 0 || mode > 2 || !name || name_len == 0) return fail(SWMI_ERR_INVALID_ARGUMENT, "mode %d outside 0..2, or no name buffer", mode);
    Context dummy;                          // make_config reads only extra_lds from the context
    SmRows rows;
    const LaunchConfig cfg = make_config(dummy, score_matrix, gap_penalty, &rows, n);
    const int L = cfg.lanes_per_alignment;
    int per_wave = 64 / L;
    if (cfg.use_pk) {
        if (L == 4) snprintf(name, name_len, "sw128_pk_kernel<%d,%d>", mode, cfg.pk_variant);
        else        snprintf(name, name_len, "sw128_pk_kernel<%d,%d,%d>", mode, cfg.pk_variant, L);
        per_wave = 128 / L;
    } else if (cfg.fold_gap && cfg.use_lut && L >= 4 && L <= 16) {
        snprintf(name, name_len, "sw128_lut_kernel<%d,%d>", L, mode);
    } else {
        snprintf(name, name_len, "sw128_kernel<%d,%d,%d,%d>", L, cfg.fold_gap ? 1 : 0, cfg.use_i16 ? 1 : 0, mode);
    }
    if (alignments_per_wavefront) *alignments_per_wavefront = per_wave;
    return SWMI_OK;
}

int swmi_get_device_info(swmi_device_info *info)
{
    if (!info) return fail(SWMI_ERR_INVALID_ARGUMENT, "info is NULL");
    Context *ctx = current();
    if (!ctx) return last_status();
    memset(info, 0, sizeof *info);
    info->device = ctx->device;
    info->compute_units = ctx->prop.multiProcessorCount;
    info->clock_khz = ctx->prop.clockRate;
    info->wavefront_size = ctx->prop.warpSize;
    info->hbm_bytes = ctx->prop.totalGlobalMem;
    snprintf(info->arch, sizeof info->arch, "%s", ctx->prop.gcnArchName);
    snprintf(info->name, sizeof info->name, "%s", ctx->prop.name);
    return SWMI_OK;
}

int swmi_score_batch(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, const int8_t score_matrix[16],
                     int8_t gap_penalty, int32_t *scores)
{
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!seq1s || !seq2s || !scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    Context *ctx = current();
    if (!ctx) return last_status();
    if (n <= kPinPairs) {               // the per-pair call and its small relatives: no copies, one launch, one wait
        std::lock_guard<std::mutex> lock(ctx->mu);
        uint8_t *h1 = ctx->pin, *h2 = ctx->pin + kPinPairs * kSeq;
        int32_t *hs = reinterpret_cast<int32_t *>(ctx->pin + 2 * kPinPairs * kSeq);
        uint8_t *dev = static_cast<uint8_t *>(ctx->pin_dev);
        memcpy(h1, seq1s, n * kSeq);
        memcpy(h2, seq2s, n * kSeq);
        rc = launch_device(*ctx, dev, dev + kPinPairs * kSeq, n, score_matrix, gap_penalty, dev + 2 * kPinPairs * kSeq, ctx->stream, false);
        const hipError_t es = hipStreamSynchronize(ctx->stream);
        if (rc != SWMI_OK) return rc;
        if (es != hipSuccess) return fail(SWMI_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
        memcpy(scores, hs, n * sizeof(int32_t));
        return SWMI_OK;
    }
    return score_host_batch(*ctx, seq1s, seq2s, n, score_matrix, gap_penalty, scores, false, false);
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
    Context *ctx = current();
    if (!ctx) return last_status();
    return score_host_batch(*ctx, seq1s, seq2, n_seq1, score_matrix, gap_penalty, scores, false, true);
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
    Context *ctx = current();
    if (!ctx) return last_status();
    SmRows rows;
    const LaunchConfig cfg = make_config(*ctx, score_matrix, gap_penalty, &rows, n_seq1);
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
    Context *ctx = current();
    if (!ctx) return last_status();
    return score_host_batch(*ctx, seq1s_packed, seq2s_packed, n, score_matrix, gap_penalty, scores, true, false);
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
    Context *ctx = current();           // makes the context's device current: the launch must not land on whatever
    if (!ctx) return last_status();   // device the calling thread happened to have selected
    hipStream_t st = static_cast<hipStream_t>(stream);   // NULL = the HIP null (default) stream, as in any HIP call
    return launch_device(*ctx, d1, d2, n, sm, gap, d_out, st, packed);
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

static swmi::SgTuning sg_tuning()
{
    const uint64_t m = sg_mapping_word().load(std::memory_order_relaxed);
    swmi::SgTuning t;
    t.force_sweep = int(m & 0xffffffffu) - 1;
    t.exact_only = int((m >> 32) & 1u);
    return t;
}

static int check_banded(const int8_t *sm, int len, int open, int ext)
{
    if (!sm) return fail(SWMI_ERR_INVALID_ARGUMENT, "score_matrix is NULL");
    if (len < 64 || len > 1792) return fail(SWMI_ERR_INVALID_ARGUMENT, "len %d outside [64, 1792]", len);
    if (open < 0 || open > 127 || ext < 0 || ext > 127)
        return fail(SWMI_ERR_DOMAIN, "gap_open %d / gap_extend %d outside [0,127]", open, ext);
    return SWMI_OK;
}

static int banded_device(Context &ctx, const void *d_seq1s, const void *d_seq2s, size_t n, int len, const int8_t *sm,
                         int gap_open, int gap_extend, void *d_scores, hipStream_t stream)
{
    const SmRows rows = pack_rows(sm, 0);
    const size_t max_launch = size_t(1) << 24;
    for (size_t off = 0; off < n; off += max_launch) {
        const size_t m = n - off < max_launch ? n - off : max_launch;
        HIP_TRY(swmi::launch_banded_affine(static_cast<const uint8_t *>(d_seq1s) + off * size_t(len),
                                           static_cast<const uint8_t *>(d_seq2s) + off * size_t(len),
                                           static_cast<int32_t *>(d_scores) + off, m, len, rows, gap_open, gap_extend, stream,
                                           !knobs().banded_no_i16, !knobs().banded_no_pk));
    }
    return SWMI_OK;
}

int swmi_banded_affine_kernel_for(int len, const int8_t score_matrix[16], int gap_open, int gap_extend, char *name, size_t name_len,
                                  int *alignments_per_wavefront)
{
    const int rc = check_banded(score_matrix, len, gap_open, gap_extend);
    if (rc != SWMI_OK) return rc;
    if (!name || name_len == 0) return fail(SWMI_ERR_INVALID_ARGUMENT, "no name buffer");
    const int choice = swmi::banded_affine_kernel_choice(len, pack_rows(score_matrix, 0), gap_open, gap_extend, !knobs().banded_no_i16,
                                                         !knobs().banded_no_pk);
    const int oge = gap_open >= gap_extend ? 1 : 0;
    if (choice == 2) snprintf(name, name_len, "sw_banded_affine_pk_kernel<%d>", oge);
    else             snprintf(name, name_len, "sw_banded_affine_kernel<%d,%d>", oge, choice);
    if (alignments_per_wavefront) *alignments_per_wavefront = choice == 2 ? 2 : 1;
    return SWMI_OK;
}

int swmi_score_banded_affine_device(const void *d_seq1s, const void *d_seq2s, size_t n, int len,
                                    const int8_t score_matrix[16], int gap_open, int gap_extend, void *d_scores, void *stream)
{
    int rc = check_banded(score_matrix, len, gap_open, gap_extend);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!d_seq1s || !d_seq2s || !d_scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    Context *ctx = current();
    if (!ctx) return last_status();
    return banded_device(*ctx, d_seq1s, d_seq2s, n, len, score_matrix, gap_open, gap_extend, d_scores, static_cast<hipStream_t>(stream));
}

int swmi_score_banded_affine(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int len, const int8_t score_matrix[16],
                             int gap_open, int gap_extend, int32_t *scores)
{
    int rc = check_banded(score_matrix, len, gap_open, gap_extend);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!seq1s || !seq2s || !scores) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    Context *ctx = current();
    if (!ctx) return last_status();
    std::lock_guard<std::mutex> lock(ctx->mu);
    // chunks of at most 128 MiB per input array, as in the 128 x 128 pipeline
    const size_t chunk_cap = kChunkPairs * kSeq / size_t(len);
    const size_t chunk = n < chunk_cap ? n : chunk_cap;
    for (int k = 0; k < kSlots && size_t(k) * chunk < n; ++k) {
        rc = ensure_slot(ctx->slots[k], chunk * size_t(len), chunk);
        if (rc != SWMI_OK) return rc;
    }
    size_t idx = 0;
    hipError_t e = hipSuccess;
    rc = SWMI_OK;
    for (size_t off = 0; off < n && e == hipSuccess && rc == SWMI_OK; off += chunk, ++idx) {
        Slot &s = ctx->slots[idx % kSlots];
        const size_t m = n - off < chunk ? n - off : chunk;
        e = hipMemcpyAsync(s.d_seq1, seq1s + off * size_t(len), m * size_t(len), hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s.d_seq2, seq2s + off * size_t(len), m * size_t(len), hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) rc = banded_device(*ctx, s.d_seq1, s.d_seq2, m, len, score_matrix, gap_open, gap_extend, s.d_scores, s.stream);
        if (e == hipSuccess && rc == SWMI_OK)
            e = hipMemcpyAsync(scores + off, s.d_scores, m * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream);
    }
    for (int k = 0; k < kSlots; ++k)
        if (ctx->slots[k].stream) {
            const hipError_t es = hipStreamSynchronize(ctx->slots[k].stream);    // also after a failure: the caller's buffers are in use
            if (e == hipSuccess) e = es;
        }
    if (rc != SWMI_OK) return rc;
    if (e != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_score_banded_affine: %s", hipGetErrorString(e));
    return SWMI_OK;
}

static int semiglobal_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores, void *d_tracebacks,
                             size_t cap, void *d_lengths, void *stream, hipEvent_t between, void *d_moves = nullptr)
{
    if (n == 0) return SWMI_OK;
    if (!d_seq1s || !d_seq2s || !d_scores || !d_lengths || (!d_tracebacks && cap != 0))
        return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    if ((reinterpret_cast<uintptr_t>(d_seq1s) | reinterpret_cast<uintptr_t>(d_seq2s) | reinterpret_cast<uintptr_t>(d_scores) |
         reinterpret_cast<uintptr_t>(d_lengths) | reinterpret_cast<uintptr_t>(d_tracebacks) | reinterpret_cast<uintptr_t>(d_moves)) & 15)
        return fail(SWMI_ERR_ALIGNMENT, "device pointers must be 16-byte aligned (the kernels use 16-byte loads and 8-byte stores)");
    if (n > (size_t(1) << 18)) return fail(SWMI_ERR_INVALID_ARGUMENT, "at most 2^18 alignments per call (got %zu)", n);
    Context *ctx = current();
    if (!ctx) return last_status();
    hipStream_t st = static_cast<hipStream_t>(stream);
    // The workspace belongs to (context, stream): calls on one stream serialise by themselves, calls on different streams
    // get different workspaces.  Lookup, growth and the launch happen under one lock, so a concurrent call can never free
    // a workspace between this call's lookup and its launch; growing synchronises only this stream.
    std::lock_guard<std::mutex> lock(ctx->ws_mu);
    Workspace &ws = ctx->sg_workspaces[st];
    const size_t need = swmi::semiglobal_workspace_bytes(n);
    if (need > ws.bytes) {
        HIP_TRY(hipStreamSynchronize(st));                // earlier launches on this stream may still use the old one
        if (ws.ptr) (void)hipFree(ws.ptr);
        ws.ptr = nullptr;
        ws.bytes = 0;
        HIP_TRY(hipMalloc(&ws.ptr, need));
        ws.bytes = need;
    }
    HIP_TRY(swmi::launch_semiglobal(static_cast<const uint8_t *>(d_seq1s), static_cast<const uint8_t *>(d_seq2s), n, ws.ptr,
                                    static_cast<int32_t *>(d_scores), static_cast<int32_t *>(d_tracebacks), cap,
                                    static_cast<uint32_t *>(d_lengths), st, between, ctx->prop.multiProcessorCount, sg_tuning(),
                                    static_cast<unsigned long long *>(d_moves)));
    return SWMI_OK;
}

int swmi_semiglobal_xdrop_moves_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores, void *d_moves,
                                       void *d_lengths, void *stream)
{
    if (n && !d_moves) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    return semiglobal_device(d_seq1s, d_seq2s, n, d_scores, nullptr, 0, d_lengths, stream, nullptr, d_moves);
}

// Host-side expansion of one alignment's moves into the reference's traceback (source.cpp:1962-1975: the (i, j) list in
// ascending order from (0, 0)).  No device involved.  Move t sits at bits 2 (t % 32) of word t / 32, in WALKING order (move 0
// leaves the best cell); the list therefore applies them last to first: 3 = diagonal, 2 = a row step, 1 = a column step.
int swmi_semiglobal_expand_moves(const uint64_t *moves, uint32_t length, int32_t *traceback, size_t cap)
{
    if (!moves || (!traceback && cap)) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (length == 0 || length > SWMI_SG_MAX_TRACEBACK) return fail(SWMI_ERR_INVALID_ARGUMENT, "length %u outside [1, %d]", length, SWMI_SG_MAX_TRACEBACK);
    const size_t count = length < cap ? length : cap;
    if (count == 0) return SWMI_OK;
    // position k as ONE 64-bit word (i in the low half, j in the high half: two int32 on a little-endian host); a step adds its
    // row / column increment to it.  Step t of the walk produces position length - 1 - t of the list.
    static const uint64_t kStep[4] = {0, uint64_t(1) << 32, 1, (uint64_t(1) << 32) | 1};
    uint64_t pos = 0;
    memcpy(traceback, &pos, sizeof pos);                 // (traceback need not be 8-byte aligned)
    int64_t t = int64_t(length) - 2;
    for (size_t k = 1; k < count; ++k, --t) {
        pos += kStep[(moves[t >> 5] >> (2 * (t & 31))) & 3u];
        memcpy(traceback + 2 * k, &pos, sizeof pos);
    }
    return SWMI_OK;
}

int swmi_semiglobal_xdrop_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores, void *d_tracebacks,
                                 size_t cap, void *d_lengths, void *stream)
{
    return semiglobal_device(d_seq1s, d_seq2s, n, d_scores, d_tracebacks, cap, d_lengths, stream, nullptr);
}

int swmi_semiglobal_set_mapping(int sweep)
{
    const bool sweep_ok = sweep == -1 || sweep == 1 || sweep == 2 || sweep == 4 || (sweep >= 11 && sweep <= 12) ||
                          (sweep >= 21 && sweep <= 23) || (sweep >= 41 && sweep <= 44);
    if (!sweep_ok) return fail(SWMI_ERR_INVALID_ARGUMENT, "sweep %d: -1, 1, 2, 4, 11..12, 21..23 or 41..44", sweep);
    // (the low half of the word; the high half is swmi_semiglobal_set_exact's)
    uint64_t was = sg_mapping_word().load(std::memory_order_relaxed);
    while (!sg_mapping_word().compare_exchange_weak(was, (was & ~uint64_t(0xffffffffu)) | uint64_t(uint32_t(sweep + 1)))) {}
    return SWMI_OK;
}

int swmi_semiglobal_set_exact(int exact_only)
{
    if (exact_only != 0 && exact_only != 1) return fail(SWMI_ERR_INVALID_ARGUMENT, "exact_only %d: 0 or 1", exact_only);
    uint64_t was = sg_mapping_word().load(std::memory_order_relaxed);
    while (!sg_mapping_word().compare_exchange_weak(was, (was & 0xffffffffu) | (uint64_t(exact_only) << 32))) {}
    return SWMI_OK;
}

int swmi_semiglobal_window_stats(void *stream, uint64_t counts[4])
{
    if (!counts) return fail(SWMI_ERR_INVALID_ARGUMENT, "counts is NULL");
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    Context *ctx = current();
    if (!ctx) return last_status();
    hipStream_t st = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lock(ctx->ws_mu);
    const auto it = ctx->sg_workspaces.find(st);
    if (it == ctx->sg_workspaces.end() || !it->second.ptr)
        return fail(SWMI_ERR_INVALID_ARGUMENT, "no semi-global call has run on this stream");
    uint32_t raw[4] = {0, 0, 0, 0};
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(raw, it->second.ptr, sizeof raw, hipMemcpyDeviceToHost));     // the first words of the workspace (sg_kernels.hip)
    for (int k = 0; k < 4; ++k) counts[k] = raw[k];
    return SWMI_OK;
}

int swmi_semiglobal_release_workspaces(void)
{
    Context *ctx = current();
    if (!ctx) return last_status();
    HIP_TRY(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(ctx->ws_mu);
    for (auto &w : ctx->sg_workspaces)
        if (w.second.ptr) (void)hipFree(w.second.ptr);
    ctx->sg_workspaces.clear();
    return SWMI_OK;
}

int swmi_semiglobal_kernels_for_batch(size_t n, char *sweep, size_t sweep_len, char *traceback, size_t traceback_len)
{
    Context *ctx = current();
    if (!ctx) return last_status();
    swmi::semiglobal_kernel_names(n, ctx->prop.multiProcessorCount, sweep, sweep_len, traceback, traceback_len, sg_tuning());
    return SWMI_OK;
}

int swmi_semiglobal_time_device(const void *d_seq1s, const void *d_seq2s, size_t n, void *d_scores, void *d_tracebacks,
                                size_t cap, void *d_lengths, void *stream, float phase_ms[2])
{
    if (!phase_ms) return fail(SWMI_ERR_INVALID_ARGUMENT, "phase_ms is NULL");
    if (n == 0) return fail(SWMI_ERR_INVALID_ARGUMENT, "n is 0");
    if (!current()) return last_status();
    int rc = SWMI_OK;
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

// Host arrays -> (scores, lengths) + either the positions (tracebacks, `cap` per alignment) or the walk's moves (moves_out,
// SWMI_SG_MOVE_WORDS words per alignment).  Chunks of up to 8192 alignments on two sets of device buffers: while the host is busy
// receiving chunk k (a copy into pageable memory blocks the caller), the GPU works on chunk k + 1.  Only as much of every
// alignment's result crosses the link as the chunk's longest path needs.
static int semiglobal_host(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int32_t *scores, int32_t *tracebacks, size_t cap,
                           uint64_t *moves_out, uint32_t *lengths)
{
    if (n == 0) return SWMI_OK;
    if (!seq1s || !seq2s || !scores || !lengths || (!tracebacks && cap != 0 && !moves_out))
        return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    Context *ctx = current();
    if (!ctx) return last_status();
    std::lock_guard<std::mutex> lock(ctx->mu);
    constexpr size_t kLen = SWMI_SG_LEN;
    const size_t move_words = swmi::semiglobal_move_words();
    // Chunk: 8192 alignments when positions return (2.1 GB of them per chunk: the link is the bound whatever the chunk), 32768
    // when moves return -- a sweep of 8192 alignments takes 9.7 ms, one of 32768 takes 12.6 (one alignment is 32768
    // sequential rounds: small launches are latency bound), and with 8 KB instead of 262 KB per alignment coming back the
    // sweeps, not the link, were what a 65536-alignment call waited for: 77 ms, eight chunks of ~9.6 ms
    const size_t chunk_max = moves_out ? 32768 : 8192;
    const size_t chunk = n < chunk_max ? n : chunk_max;
    using Set = SgSet;
    Set *sets = ctx->sg_sets;
    const int n_sets = n > chunk ? 2 : 1;
    hipStream_t streams[2] = {ctx->slots[0].stream, ctx->slots[1].stream};
    hipError_t e = hipSuccess;
    for (int k = 0; k < n_sets && e == hipSuccess; ++k) {
        Set &s = sets[k];
        s.off = s.m = 0;
        const size_t tb_need = moves_out ? 0 : chunk * (cap ? cap : 1);
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
        if (e == hipSuccess && moves_out && s.move_rows < chunk) {
            (void)hipFree(s.d_moves);
            s.d_moves = nullptr; s.move_rows = 0;
            e = hipMalloc(reinterpret_cast<void **>(&s.d_moves), chunk * move_words * sizeof(uint64_t));
            if (e == hipSuccess) s.move_rows = chunk;
        }
    }
    // results of the chunk a set holds -> host; only as many positions / moves per alignment as the longest path of the chunk has
    auto drain = [&](int which) -> hipError_t {
        Set &s = sets[which];
        hipStream_t st = streams[which];
        if (s.m == 0) return hipSuccess;
        hipError_t r = hipMemcpyAsync(scores + s.off, s.d_scores, s.m * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (r == hipSuccess) r = hipMemcpyAsync(lengths + s.off, s.d_len, s.m * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (r == hipSuccess) r = hipStreamSynchronize(st);
        if (r == hipSuccess && (cap || moves_out)) {
            size_t longest = 0;
            for (size_t k = 0; k < s.m; ++k) longest = lengths[s.off + k] > longest ? lengths[s.off + k] : longest;
            if (moves_out) {
                const size_t pitch = move_words * sizeof(uint64_t);
                const size_t words = longest > 1 ? (longest - 1 + 31) / 32 : 1;              // positions = moves + 1, 32 moves per word
                r = hipMemcpy2DAsync(moves_out + s.off * move_words, pitch, s.d_moves, pitch, words * sizeof(uint64_t), s.m,
                                     hipMemcpyDeviceToHost, st);
            } else {
                if (longest > cap) longest = cap;
                const size_t pitch = cap * 2 * sizeof(int32_t);
                r = hipMemcpy2DAsync(tracebacks + s.off * cap * 2, pitch, s.d_tb, pitch, longest * 2 * sizeof(int32_t), s.m,
                                     hipMemcpyDeviceToHost, st);
            }
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
        if (e == hipSuccess)
            e = swmi::launch_semiglobal(s.d1, s.d2, s.m, s.ws, s.d_scores, moves_out ? nullptr : s.d_tb, moves_out ? 0 : cap, s.d_len, st, nullptr,
                                        ctx->prop.multiProcessorCount, sg_tuning(), moves_out ? s.d_moves : nullptr);
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

int swmi_semiglobal_xdrop(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int32_t *scores, int32_t *tracebacks,
                          size_t cap, uint32_t *lengths)
{
    return semiglobal_host(seq1s, seq2s, n, scores, tracebacks, cap, nullptr, lengths);
}

int swmi_semiglobal_xdrop_moves(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, int32_t *scores, uint64_t *moves, uint32_t *lengths)
{
    if (n && !moves) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    return semiglobal_host(seq1s, seq2s, n, scores, nullptr, 0, moves, lengths);
}

int swmi_unpack(const uint8_t *packed, size_t n_seqs, uint8_t *unpacked)
{
    if (n_seqs == 0) return SWMI_OK;
    if (!packed || !unpacked) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n_seqs = %zu", n_seqs);
    Context *ctx = current();
    if (!ctx) return last_status();
    std::lock_guard<std::mutex> lock(ctx->mu);
    // reuse slot 0: seq1 buffer holds the packed bytes, seq2 buffer the unpacked ones
    const size_t chunk = n_seqs < kChunkPairs ? n_seqs : kChunkPairs;
    int rc = ensure_slot(ctx->slots[0], chunk * kSeq, 0);
    if (rc != SWMI_OK) return rc;
    Slot &s = ctx->slots[0];
    hipError_t e = hipSuccess;
    for (size_t off = 0; off < n_seqs && e == hipSuccess; off += chunk) {
        const size_t m = n_seqs - off < chunk ? n_seqs - off : chunk;
        e = hipMemcpyAsync(s.d_seq1, packed + off * SWMI_PACKED_LEN, m * SWMI_PACKED_LEN, hipMemcpyHostToDevice, s.stream);
        if (e == hipSuccess) e = swmi::launch_unpack(s.d_seq1, s.d_seq2, m, s.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(unpacked + off * kSeq, s.d_seq2, m * kSeq, hipMemcpyDeviceToHost, s.stream);
        const hipError_t es = hipStreamSynchronize(s.stream);              // also after a failure: the caller's buffers are in use
        if (e == hipSuccess) e = es;
    }
    if (e != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_unpack: %s", hipGetErrorString(e));
    return SWMI_OK;
}

size_t swmi_host_granules_for(size_t n, int entry, size_t *granules, size_t cap)
{
    {
        std::lock_guard<std::mutex> lock(g_init_mu);
        if (!g_knobs_read) read_knobs();
    }
    if (entry < kEntryPairs || entry > kEntryOneVsMany) return 0;
    const size_t per_pair = host_entry_bytes(entry), score_group = knobs().score_group;
    size_t count = 0;
    std::vector<size_t> sizes;
    for (size_t group = 0; group < n; group += score_group) {
        const size_t group_n = n - group < score_group ? n - group : score_group;
        granule_list(group_n, per_pair, &sizes);
        for (size_t m : sizes) {
            if (granules && count < cap) granules[count] = m;
            ++count;
        }
    }
    return count;
}

size_t swmi_host_granules(size_t n, size_t *granules, size_t cap) { return swmi_host_granules_for(n, SWMI_ENTRY_PAIRS, granules, cap); }

int swmi_selftest_pk_max3(unsigned long long *checked, unsigned long long *mismatches)
{
    if (!checked || !mismatches) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL output");
    Context *ctx = current();
    if (!ctx) return last_status();
    std::lock_guard<std::mutex> lock(ctx->mu);
    // the pinned, device-visible staging buffer doubles as the two counters (zeroed by the host, added to by the kernel)
    unsigned long long *h = reinterpret_cast<unsigned long long *>(ctx->pin);
    h[0] = h[1] = 0;
    HIP_TRY(swmi::launch_pk_max3_selftest(static_cast<unsigned long long *>(ctx->pin_dev), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *checked = h[0];
    *mismatches = h[1];
    return SWMI_OK;
}

int swmi_generate_pairs_device(void *d_seq1s, void *d_seq2s, size_t n, uint64_t seed, uint64_t first_pair, void *stream)
{
    if (n == 0) return SWMI_OK;
    if (!d_seq1s || !d_seq2s) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL device buffer with n = %zu", n);
    if ((reinterpret_cast<uintptr_t>(d_seq1s) | reinterpret_cast<uintptr_t>(d_seq2s)) & 15)
        return fail(SWMI_ERR_ALIGNMENT, "device pointers must be 16-byte aligned");
    if (!current()) return last_status();
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
    if (!current()) return last_status();
    int rc = SWMI_OK;
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
    Context *ctx = current();
    if (!ctx) return last_status();
    swmi_queue *q = new (std::nothrow) swmi_queue;
    if (!q) return fail(SWMI_ERR_INVALID_ARGUMENT, "out of host memory");
    q->ctx = context_ref(ctx->index);
    q->device = ctx->device;
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
    int rc = check_alive(*q->ctx);
    if (rc != SWMI_OK) return rc;
    rc = queue_ship(q, q->count);
    const hipError_t es = hipStreamSynchronize(q->stream);       // on failure too: shipments in flight read the staging memory
    if (rc != SWMI_OK) return rc;
    if (es != hipSuccess) return fail(SWMI_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
    if (scores) *scores = q->h_scores;
    if (n_scores) *n_scores = q->count;
    return SWMI_OK;
}

int swmi_queue_reset(swmi_queue *q)
{
    if (!q) return fail(SWMI_ERR_INVALID_ARGUMENT, "queue is NULL");
    const int alive = check_alive(*q->ctx);
    if (alive != SWMI_OK) return alive;
    HIP_TRY(hipStreamSynchronize(q->stream));
    q->count = q->shipped = 0;
    return SWMI_OK;
}

int swmi_queue_destroy(swmi_queue *q)
{
    if (!q) return SWMI_OK;
    if (q->device >= 0) (void)hipSetDevice(q->device);     // (valid after swmi_shutdown too: the queue owns its stream and buffers)
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
