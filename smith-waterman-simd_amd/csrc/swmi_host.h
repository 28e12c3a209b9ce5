// swmi_host.h -- host-side state shared by swmi_api.cpp (single-GPU entry points) and swmi_multi.cpp (the batch over
// several GPUs).  Not installed: the public interface is include/swmi.h.
#pragma once
#include "../../include/swmi.h"
#include "swmi_internal.h"

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace swmi {
namespace host {

constexpr size_t kSeq = SWMI_SEQ_LEN;
constexpr size_t kChunkPairs = size_t(1) << 20;      // host-batch pipeline granule: 1M pairs = 128 MiB per input array
constexpr size_t kMaxLaunchPairs = size_t(1) << 30;  // pairs per kernel launch (the kernel indexes pairs with uint32)
constexpr int kSlots = 2;
constexpr size_t kPinPairs = 64;                     // host batches up to this size go through the pinned staging buffer

struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *d_seq1 = nullptr, *d_seq2 = nullptr;
    int32_t *d_scores = nullptr;
    size_t capacity = 0;   // pairs
};

// device buffers of the host-buffer semi-global entry (two chunks in flight), kept between calls and grown on demand
struct SgSet {
    uint8_t *d1 = nullptr, *d2 = nullptr;
    void *ws = nullptr;
    int32_t *d_scores = nullptr, *d_tb = nullptr;
    uint32_t *d_len = nullptr;
    size_t alignments = 0, tb_entries = 0;      // capacity
    size_t off = 0, m = 0;                      // chunk in flight
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
    hipDeviceProp_t prop{};
    hipStream_t stream = nullptr;       // library-owned stream (per-pair path, helpers)
    Slot slots[kSlots];
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
int num_contexts();
std::mutex &init_mutex();

int check_params(const int8_t *sm, int gap);
SmRows pack_rows(const int8_t *sm, int add);
LaunchConfig make_config(const Context &ctx, const int8_t *sm, int gap, SmRows *rows, size_t n);
int launch_device(Context &ctx, const void *d1, const void *d2, size_t n, const int8_t *sm, int gap, void *d_out,
                  hipStream_t st, bool packed);
// host arrays -> scores through ctx's two slots (the body of swmi_score_batch); takes ctx.mu
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
