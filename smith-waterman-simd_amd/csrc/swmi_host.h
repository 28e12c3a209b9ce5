// swmi_host.h -- host-side state shared by swmi_api.cpp (single-GPU entry points) and swmi_multi.cpp (the batch over
// several GPUs).  Not installed: the public interface is include/swmi.h.
#pragma once
#include "../../include/swmi.h"
#include "swmi_internal.h"

#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace swmi {
namespace host {

constexpr size_t kSeq = SWMI_SEQ_LEN;
constexpr size_t kChunkPairs = size_t(1) << 20;      // largest host-batch pipeline granule: 1M pairs = 128 MiB per input array
constexpr size_t kMinGranule = size_t(1) << 14;      // smallest one (the tail of a tapered schedule)
constexpr size_t kBalancedGranule = size_t(1) << 17; // the steady granule where copy and kernel take the same time (2-bit packed input)
constexpr size_t kMaxTaper1024 = 820;                // steepest granule-to-granule ratio, in 1/1024 (0.8): beyond it a 1M-pair batch
                                                     //   breaks into so many copy commands that their fixed cost outweighs the tail
constexpr size_t kScoreGroup = size_t(1) << 24;      // pairs whose scores return to the host in one copy (64 MiB); Knobs::score_group
constexpr size_t kMaxLaunchPairs = size_t(1) << 30;  // pairs per kernel launch (the kernel indexes pairs with uint32)
constexpr int kSlots = 6;                            // device input buffer sets of the host-batch pipeline: up to three per issuing thread
constexpr int kHostThreads = 2;                      // host threads that issue a host batch's copies and kernels (Knobs::host_threads)
constexpr size_t kPinPairs = 64;                     // host batches up to this size go through the pinned staging buffer

// SWMI_* environment knobs -- experiment / rehearsal switches, none needed in production.  Read ONCE, by swmi_init*
// (under the init mutex) or by the first call that needs one before any init; never on a launch path (getenv is not safe
// against a concurrent setenv, and a launch should not pay for it).
struct Knobs {
    size_t host_granule = 0;        // SWMI_HOST_GRANULE: fixed pipeline granule in pairs (0 = tapered schedule)
    bool host_serial = false;       // SWMI_HOST_SERIAL=1: round 2's pipeline (scores copied back behind every granule), for the A/B
    unsigned host_taper_pct = 0;    // SWMI_HOST_TAPER: granule-to-granule ratio of the tapered schedule in percent (0 = derived from
                                    //   the entry's bytes per pair, next_granule())
    size_t host_min_granule = 0;    // SWMI_HOST_MIN_GRANULE: smallest granule of the tapered schedule (0 = kMinGranule)
    size_t host_schedule[16] = {};      // SWMI_HOST_SCHEDULE="a,b,c,...": explicit granule sizes (the last one repeats), experiments only
    bool host_trace = false;            // SWMI_HOST_TRACE=1: every host batch prints its copy / kernel timeline (HIP events) to stderr
    int host_slots = 0;                 // SWMI_HOST_SLOTS: buffer sets per issuing thread (2 or 3; 0 = the entry's default)
    int host_threads = 0;               // SWMI_HOST_THREADS: issuing threads (1 or 2; 0 = the entry's default): the calling thread alone issues a host batch (round 3's pipeline), for the A/B
    size_t score_group = kScoreGroup;   // SWMI_TEST_SCORE_GROUP: pairs per score copy -- test-only, so that the several-group
                                    //   branch of score_host_batch runs at sizes a test can afford
    unsigned extra_lds = 0;         // SWMI_EXTRA_LDS: unused dynamic LDS per workgroup (occupancy sweep, BASELINE config 3)
    int lanes = 0;                  // SWMI_LANES: initial schedule
    bool banded_no_i16 = false;     // SWMI_BANDED_NO_I16
    bool banded_no_pk = false;      // SWMI_BANDED_NO_PK: never the packed banded-affine kernel (A/B against the int32 cell)
    int sg_sweep = -1;              // SWMI_SG_SWEEP: force a semi-global sweep mapping (sg_kernels.hip choose_sweep)
    int sg_exact = -1;              // SWMI_SG_EXACT: 1 = the sweeps run the X-drop test every round (no calm windows); initial value only
    bool gather_p2p = false;        // SWMI_GATHER_BACKEND=p2p: never RCCL
    size_t gather_piece = 0;        // SWMI_TEST_GATHER_PIECE: ragged RCCL gather even for equal shards, shards broadcast in
                                    //   pieces of this many scores (rehearses the ragged path with one rank)
    char rccl_lib[256] = "";        // SWMI_RCCL_LIB: library to dlopen instead of librccl.so (a missing one rehearses the fallback)
};
const Knobs &knobs();
void read_knobs();                  // (re)reads the environment; callers hold the init mutex

struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *d_seq1 = nullptr, *d_seq2 = nullptr;
    size_t capacity = 0;            // bytes of each of the two input buffers
    int32_t *d_scores = nullptr;    // only the banded-affine host entry keeps per-slot scores (the 128x128 pipeline writes
    size_t score_capacity = 0;      //   into Context::d_scores_all); capacity in scores
};

// What a host-batch entry ships per pair (next_granule derives the pipeline's taper from it)
enum HostEntry { kEntryPairs = 0, kEntryPacked = 1, kEntryOneVsMany = 2 };
size_t host_entry_bytes(int entry);                 // 256 / 64 / 128
size_t next_granule(size_t remaining, size_t bytes_per_pair);
void granule_list(size_t group_n, size_t bytes_per_pair, std::vector<size_t> *out);   // the granules of one score group, in order

// device buffers of the host-buffer semi-global entry (two chunks in flight), kept between calls and grown on demand
struct SgSet {
    uint8_t *d1 = nullptr, *d2 = nullptr;
    void *ws = nullptr;
    int32_t *d_scores = nullptr, *d_tb = nullptr;
    uint32_t *d_len = nullptr;
    unsigned long long *d_moves = nullptr;      // [alignments][SWMI_SG_MOVE_WORDS]: the entry that returns moves instead of positions
    size_t alignments = 0, tb_entries = 0, move_rows = 0;      // capacity
    size_t off = 0, m = 0;                      // chunk in flight
};

// A PERSISTENT host thread that runs one job at a time: a thread's first HIP call costs ~1 ms of runtime set-up, which a
// thread spawned per call would pay every time.  swmi_multi.cpp keeps one per bound GPU beyond the first (the host-array
// entry points over several GPUs), every Context one for the second half of its host-batch pipeline (score_host_batch).
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, stop = false;
    Worker() { th = std::thread([this] { loop(); }); }
    ~Worker() { shut(); }               // also at process exit without swmi_shutdown(): a joinable std::thread must not be destroyed
    Worker(const Worker &) = delete;
    Worker &operator=(const Worker &) = delete;
    void shut()
    {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
    void loop()
    {
        std::unique_lock<std::mutex> l(mu);
        for (;;) {
            cv.wait(l, [this] { return has_job || stop; });
            if (has_job) {              // (a job handed over before the stop still runs: its submitter waits for it)
                l.unlock();
                job();
                l.lock();
                has_job = false;
                cv.notify_all();
                continue;
            }
            return;                     // stop, nothing pending
        }
    }
    void submit(std::function<void()> f)
    {
        std::lock_guard<std::mutex> l(mu);
        job = std::move(f);
        has_job = true;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [this] { return !has_job; });
    }
};

struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
};

// Everything the library owns on ONE bound GPU.  A GPU may be bound twice (swmi_init_devices({0, 0})): two contexts,
// two stream sets, the same hardware.
struct Context {
    int index = 0;                      // position in the bound list = the argument of swmi_use_gpu
    int device = -1;                    // HIP ordinal
    // swmi_shutdown() releases everything below and sets `dead`; the object itself lives as long as a handle (queue,
    // sharded batch) still points at it, so that a late *_destroy or a call on a stale handle finds a flag, not freed memory
    std::atomic<bool> dead{false};
    hipDeviceProp_t prop{};
    hipStream_t stream = nullptr;       // library-owned stream (per-pair path, helpers)
    Slot slots[kSlots];
    int32_t *d_scores_all = nullptr;    // host-batch pipeline: scores of one group of granules (up to kScoreGroup pairs)
    size_t scores_all_capacity = 0;
    hipEvent_t slot_done[kSlots] = {};  // recorded behind a slot's last kernel of a group
    hipEvent_t slot_early[kSlots] = {}; // ... behind its last kernel of the group's EARLY granules (all but the last two)
    std::unique_ptr<Worker> copier;     // second issuing thread of the host-batch pipeline, created by the first batch that uses it
    SgSet sg_sets[2];
    // Semi-global device entry: one workspace per caller stream, so that calls on different streams may be in flight at
    // once; a workspace only grows (after synchronising ITS stream), and is looked up, grown and handed to the launch
    // under ws_mu, so no launch can be enqueued on a workspace that a concurrent call frees.
    std::mutex ws_mu;
    std::map<hipStream_t, Workspace> sg_workspaces;
    // pinned, device-visible staging for tiny host batches (the per-pair call): the kernel reads the pairs from host
    // memory and writes the scores back there, so a call is one launch + one synchronisation, no copies
    uint8_t *pin = nullptr;             // [kPinPairs * 128] seq1s, [kPinPairs * 128] seq2s, [kPinPairs] int32 scores
    void *pin_dev = nullptr;            // the same memory as the device sees it
    unsigned extra_lds = 0;             // SWMI_EXTRA_LDS: occupancy sweep knob (BASELINE config 3)
    std::mutex mu;                      // serialises use of the slots, the sg sets and the pinned buffer
};

// ---- state (swmi_api.cpp) ----
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int last_status();                      // the code the last fail() on this thread returned
Context *current();                     // the calling thread's context, made current on its device; nullptr + last_status()
Context *context_at(int index);         // nullptr if out of range
std::shared_ptr<Context> context_ref(int index);   // what a handle keeps
void stop_workers();                    // swmi_multi.cpp: joins the per-GPU host threads of the *_multi entry points (swmi_shutdown)
int check_alive(const Context &ctx);    // SWMI_OK, or SWMI_ERR_NOT_INITIALIZED when swmi_shutdown() ran after the handle was made
int num_contexts();
std::mutex &init_mutex();

std::atomic<uint64_t> &sg_mapping_word();          // semi-global mapping override (swmi_semiglobal_set_mapping)
int check_params(const int8_t *sm, int gap);
SmRows pack_rows(const int8_t *sm, int add);
LaunchConfig make_config(const Context &ctx, const int8_t *sm, int gap, SmRows *rows, size_t n);
int launch_device(Context &ctx, const void *d1, const void *d2, size_t n, const int8_t *sm, int gap, void *d_out,
                  hipStream_t st, bool packed);
// host arrays -> scores through ctx's buffer sets (the body of swmi_score_batch and its relatives); takes ctx.mu
int score_host_batch(Context &ctx, const uint8_t *s1, const uint8_t *s2, size_t n, const int8_t *sm, int gap,
                     int32_t *out, bool packed, bool one_vs_many);

#define SWMI_HIP_TRY(expr)                                                                                          \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess)                                                                                       \
            return ::swmi::host::fail(SWMI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                                      __LINE__);                                                                    \
    } while (0)

}  // namespace host
}  // namespace swmi
