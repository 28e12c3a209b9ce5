// swmi_multi.cpp -- one batch over the G bound GPUs (include/swmi.h, "multi-GPU").
//
// The reference's 1M-call loop (source.cpp:3074-3082) scores independent pairs, so the batch shards trivially: GPU g gets
// the contiguous range shard_bounds(n, g, G), nothing but the final int32 scores ever crosses GPUs.  Two shapes:
//   host arrays      swmi_score_batch[_packed]_multi: one host thread per GPU drives that GPU's two-slot copy / kernel / copy
//                    pipeline (swmi_api.cpp score_host_batch) on its shard and lands the scores in the caller's slice
//   resident shards  swmi_sharded_*: inputs and scores stay in each GPU's HBM; kernels go out on one stream per GPU and the
//                    gather follows on the same streams -- peer-to-peer DMA into GPU 0 (ROOT) or an RCCL all-gather (ALL)
// RCCL is loaded on first use (dlopen), not linked: a process that never asks for SWMI_GATHER_ALL never maps librccl, and
// inside a PyTorch process the librccl torch already loaded serves both.
#include "swmi_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>

using namespace swmi::host;

namespace {

// ---- RCCL through dlopen --------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    bool tried = false;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;                    // why it is unusable, if it is
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// `why` (optional) receives the reason under the same lock that guards g_rccl
bool load_rccl(std::string *why = nullptr)
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    struct Report {                                 // every exit leaves the reason with the caller
        std::string *dst;
        ~Report() { if (dst) *dst = g_rccl.why; }
    } report{why};
    if (g_rccl.tried) return g_rccl.handle != nullptr;
    g_rccl.tried = true;
    const char *override_lib = knobs().rccl_lib[0] ? knobs().rccl_lib : nullptr;       // SWMI_RCCL_LIB
    const char *defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    const char *const *names = override_lib ? &override_lib : defaults;
    const int n_names = override_lib ? 1 : 3;
    for (int k = 0; k < n_names && !g_rccl.handle; ++k) {
        (void)dlerror();
        g_rccl.handle = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl.handle) {
            const char *err = dlerror();            // ONE call: dlerror() clears the state it returns
            g_rccl.why = err ? err : "librccl.so not found";
        }
    }
    if (!g_rccl.handle) return false;
    g_rccl.why.clear();
    auto sym = [&](const char *nm) { return dlsym(g_rccl.handle, nm); };
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.Broadcast = reinterpret_cast<decltype(g_rccl.Broadcast)>(sym("ncclBroadcast"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.Broadcast || !g_rccl.GroupStart ||
        !g_rccl.GroupEnd || !g_rccl.GetErrorString) {
        g_rccl.why = "librccl.so lacks an expected symbol";
        dlclose(g_rccl.handle);
        g_rccl.handle = nullptr;
        return false;
    }
    return true;
}

void shard_bounds(size_t n, int g, int G, size_t *lo, size_t *hi)
{
    const size_t base = n / size_t(G), extra = n % size_t(G);
    const size_t l = size_t(g) * base + (size_t(g) < extra ? size_t(g) : extra);
    *lo = l;
    *hi = l + base + (size_t(g) < extra ? 1 : 0);
}

// One PERSISTENT host thread (swmi_host.h Worker) per bound GPU beyond the first for the host-array entry points -- spawned
// per call, as in round 2, a 1M-pair call took 6.1 ms instead of 4.7: with eight GPUs a shard's whole copy is shorter than a
// thread's first HIP call.  Shard 0 runs on the calling thread, so index 0 stays empty.  Created on first use, joined by
// swmi_shutdown (stop_workers).
std::vector<std::unique_ptr<Worker>> g_workers;
std::mutex g_workers_mu;                // one multi-GPU host batch at a time (they would serialise on the contexts anyway)

void ensure_workers(int G)
{
    while ((int)g_workers.size() < G) g_workers.emplace_back(g_workers.empty() ? nullptr : new Worker);     // [0]: the caller itself
}

int multi_host(const uint8_t *s1, const uint8_t *s2, size_t n, const int8_t *sm, int gap, int32_t *out, bool packed)
{
    int rc = check_params(sm, gap);
    if (rc != SWMI_OK) return rc;
    if (n == 0) return SWMI_OK;
    if (!s1 || !s2 || !out) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL buffer with n = %zu", n);
    const int G = num_contexts();
    if (G == 0) return fail(SWMI_ERR_NOT_INITIALIZED, "swmi_init_all() has not been called (or failed)");
    const size_t stride = packed ? SWMI_PACKED_LEN : kSeq;
    std::vector<int> rcs(G, SWMI_OK);
    std::vector<std::string> errs(G);
    std::lock_guard<std::mutex> pool_lock(g_workers_mu);
    ensure_workers(G);
    auto shard_job = [&, stride](int g, size_t lo, size_t hi) {
        const std::shared_ptr<Context> keep = context_ref(g);      // (a racing swmi_shutdown is the caller's bug, but it must not free this)
        if (!keep) {
            rcs[g] = SWMI_ERR_NOT_INITIALIZED;
            errs[g] = "the context disappeared";
            return;
        }
        Context *ctx = keep.get();
        const hipError_t e = hipSetDevice(ctx->device);          // per host thread, like every HIP "current device"
        if (e != hipSuccess) {
            rcs[g] = SWMI_ERR_HIP;
            errs[g] = std::string("hipSetDevice failed: ") + hipGetErrorString(e);
            return;
        }
        rcs[g] = score_host_batch(*ctx, s1 + lo * stride, s2 + lo * stride, hi - lo, sm, gap, out + lo, packed, false);
        if (rcs[g] != SWMI_OK) errs[g] = swmi_last_error();      // thread-local text: carry it to the caller's thread
    };
    // shard 0 runs on the calling thread (one GPU: no hand-over at all), the others on their GPU's worker
    std::vector<int> busy;
    size_t lo0 = 0, hi0 = 0;
    shard_bounds(n, 0, G, &lo0, &hi0);
    for (int g = 1; g < G; ++g) {
        size_t lo, hi;
        shard_bounds(n, g, G, &lo, &hi);
        if (hi == lo) continue;
        g_workers[g]->submit([=, &shard_job] { shard_job(g, lo, hi); });
        busy.push_back(g);
    }
    if (hi0 > lo0) shard_job(0, lo0, hi0);
    for (int g : busy) g_workers[g]->wait();
    for (int g = 0; g < G; ++g)
        if (rcs[g] != SWMI_OK) return fail(rcs[g], "GPU index %d: %s", g, errs[g].c_str());
    return SWMI_OK;
}

}  // namespace

namespace swmi {
namespace host {
void stop_workers()
{
    std::lock_guard<std::mutex> pool_lock(g_workers_mu);
    for (auto &w : g_workers)
        if (w) w->shut();
    g_workers.clear();
    // a FAILED attempt to load librccl is forgotten, so that a re-init with another SWMI_RCCL_LIB tries again (a loaded
    // library stays: communicators of live batches point into it)
    std::lock_guard<std::mutex> rccl_lock(g_rccl_mu);
    if (!g_rccl.handle) g_rccl.tried = false;
}
}  // namespace host
}  // namespace swmi

struct swmi_sharded_batch {
    size_t n = 0;
    bool packed = false;
    struct Part {
        std::shared_ptr<Context> ctx;   // kept alive past swmi_shutdown (Context::dead); launches read its settings
        int device = -1;                // HIP ordinal: all that swmi_sharded_destroy needs
        size_t lo = 0, hi = 0;
        uint8_t *d1 = nullptr, *d2 = nullptr;
        int32_t *d_scores = nullptr;
        int32_t *d_gathered = nullptr;  // int32[n] on this GPU, allocated on the first gather that needs it
        hipStream_t stream = nullptr;
        std::vector<hipEvent_t> ev;     // timing: 3 per iteration of swmi_sharded_time
    };
    std::vector<Part> parts;
    std::vector<ncclComm_t> comms;      // one per part (single-process RCCL), created on the first GATHER_ALL
    bool rccl_usable = false, rccl_decided = false;
    std::string rccl_why;               // why SWMI_GATHER_ALL runs on peer copies, when it does
    bool equal_shards = false;
    size_t piece = 0;                   // ragged RCCL gather: scores per broadcast (0 = a shard at a time)
};

namespace {

int batch_alive(const swmi_sharded_batch *b)
{
    for (const auto &p : b->parts) {
        const int rc = check_alive(*p.ctx);
        if (rc != SWMI_OK) return rc;
    }
    return SWMI_OK;
}

int ensure_gathered(swmi_sharded_batch *b, int g)
{
    auto &p = b->parts[g];
    if (p.d_gathered) return SWMI_OK;
    SWMI_HIP_TRY(hipSetDevice(p.device));
    SWMI_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p.d_gathered), (b->n ? b->n : 1) * sizeof(int32_t)));
    return SWMI_OK;
}

// RCCL needs one communicator rank per DISTINCT device; a GPU bound twice (rehearsal) falls back to peer copies.
// The reason stays with the batch (swmi_sharded_gather_note) and goes to swmi_last_error() of the deciding call.
void decide_rccl(swmi_sharded_batch *b)
{
    if (b->rccl_decided) return;
    b->rccl_decided = true;
    const int G = (int)b->parts.size();
    if (knobs().gather_p2p) {                                    // SWMI_GATHER_BACKEND=p2p
        b->rccl_why = "SWMI_GATHER_BACKEND=p2p";
        return;
    }
    std::vector<int> devs(G);
    for (int g = 0; g < G; ++g) {
        devs[g] = b->parts[g].device;
        for (int k = 0; k < g; ++k)
            if (devs[k] == devs[g]) {
                b->rccl_why = "device " + std::to_string(devs[g]) + " is bound twice: RCCL needs one rank per distinct device";
                return;
            }
    }
    std::string why;
    if (!load_rccl(&why)) {
        b->rccl_why = "librccl could not be loaded: " + why;
        return;
    }
    b->comms.assign(G, nullptr);
    const ncclResult_t r = g_rccl.CommInitAll(b->comms.data(), G, devs.data());
    if (r != ncclSuccess) {
        b->rccl_why = std::string("ncclCommInitAll failed: ") + g_rccl.GetErrorString(r);
        for (ncclComm_t c : b->comms)                            // whatever part of the clique was created
            if (c) (void)g_rccl.CommDestroy(c);
        b->comms.clear();
        return;
    }
    b->rccl_usable = true;
}

// one score call: kernels on every GPU, then the gather; `ev3` (optional) = this iteration's events per part
int score_once(swmi_sharded_batch *b, const int8_t *sm, int gap, int gather, int iter_for_events)
{
    const int G = (int)b->parts.size();
    for (int g = 0; g < G; ++g) {
        auto &p = b->parts[g];
        const size_t m = p.hi - p.lo;
        SWMI_HIP_TRY(hipSetDevice(p.device));
        hipEvent_t *ev = iter_for_events >= 0 ? &p.ev[3 * size_t(iter_for_events)] : nullptr;
        if (ev) SWMI_HIP_TRY(hipEventRecord(ev[0], p.stream));
        if (m) {
            const int rc = launch_device(*p.ctx, p.d1, p.d2, m, sm, gap, p.d_scores, p.stream, b->packed);
            if (rc != SWMI_OK) return rc;
        }
        if (ev) SWMI_HIP_TRY(hipEventRecord(ev[1], p.stream));
    }
    if (gather == SWMI_GATHER_ROOT || (gather == SWMI_GATHER_ALL && !b->rccl_usable)) {
        const int targets = gather == SWMI_GATHER_ROOT ? 1 : G;
        for (int g = 0; g < G; ++g) {
            auto &p = b->parts[g];
            const size_t m = p.hi - p.lo;
            if (!m) continue;
            SWMI_HIP_TRY(hipSetDevice(p.device));
            for (int r = 0; r < targets; ++r) {                  // push: the copy runs on the SOURCE GPU's stream, behind its kernel
                auto &dst = b->parts[r];
                if (dst.device == p.device)
                    SWMI_HIP_TRY(hipMemcpyAsync(dst.d_gathered + p.lo, p.d_scores, m * sizeof(int32_t), hipMemcpyDeviceToDevice, p.stream));
                else
                    SWMI_HIP_TRY(hipMemcpyPeerAsync(dst.d_gathered + p.lo, dst.device, p.d_scores, p.device,
                                                    m * sizeof(int32_t), p.stream));
            }
        }
    } else if (gather == SWMI_GATHER_ALL) {
        ncclResult_t r = g_rccl.GroupStart();
        for (int g = 0; g < G && r == ncclSuccess; ++g) {
            auto &p = b->parts[g];
            if (b->equal_shards) {
                r = g_rccl.AllGather(p.d_scores, p.d_gathered, p.hi - p.lo, ncclInt32, b->comms[g], p.stream);
            } else {                                             // ragged shards: broadcasts per non-empty shard, same order on every rank
                for (int root = 0; root < G && r == ncclSuccess; ++root) {
                    const auto &src = b->parts[root];
                    const size_t m = src.hi - src.lo, piece = b->piece ? b->piece : m;      // (pieces: SWMI_TEST_GATHER_PIECE)
                    for (size_t off = 0; off < m && r == ncclSuccess; off += piece) {
                        const size_t cnt = m - off < piece ? m - off : piece;
                        // the send buffer is read on the root only; every rank passes its own shard pointer (offset as on the root)
                        r = g_rccl.Broadcast(p.d_scores + (g == root ? off : 0), p.d_gathered + src.lo + off, cnt, ncclInt32, root,
                                             b->comms[g], p.stream);
                    }
                }
            }
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = re;
        if (r != ncclSuccess) return fail(SWMI_ERR_HIP, "RCCL gather failed: %s", g_rccl.GetErrorString(r));
    }
    if (iter_for_events >= 0)
        for (int g = 0; g < G; ++g) {
            auto &p = b->parts[g];
            SWMI_HIP_TRY(hipSetDevice(p.device));
            SWMI_HIP_TRY(hipEventRecord(p.ev[3 * size_t(iter_for_events) + 2], p.stream));
        }
    return SWMI_OK;
}

int prepare_gather(swmi_sharded_batch *b, int gather)
{
    if (gather != SWMI_GATHER_NONE && gather != SWMI_GATHER_ROOT && gather != SWMI_GATHER_ALL)
        return fail(SWMI_ERR_INVALID_ARGUMENT, "unknown gather mode %d", gather);
    if (gather == SWMI_GATHER_ROOT) return ensure_gathered(b, 0);
    if (gather == SWMI_GATHER_ALL) {
        const bool deciding = !b->rccl_decided;
        decide_rccl(b);
        // not an error -- the gather runs on peer copies and gives the same bytes -- but never silent: the text is what
        // swmi_last_error() returns after this call, and swmi_sharded_gather_note() keeps it
        if (deciding && !b->rccl_usable) (void)fail(SWMI_OK, "SWMI_GATHER_ALL runs on peer copies, not RCCL: %s", b->rccl_why.c_str());
        for (int g = 0; g < (int)b->parts.size(); ++g) {
            const int rc = ensure_gathered(b, g);
            if (rc != SWMI_OK) return rc;
        }
    }
    return SWMI_OK;
}

int wait_all(swmi_sharded_batch *b)
{
    hipError_t e = hipSuccess;
    for (auto &p : b->parts) {
        hipError_t es = hipSetDevice(p.device);
        if (es == hipSuccess) es = hipStreamSynchronize(p.stream);
        if (e == hipSuccess) e = es;
    }
    if (e != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_sharded_wait: %s", hipGetErrorString(e));
    return SWMI_OK;
}

}  // namespace

extern "C" {

int swmi_shard_bounds(size_t n, int shard, int n_shards, size_t *lo, size_t *hi)
{
    if (n_shards <= 0 || shard < 0 || shard >= n_shards || !lo || !hi)
        return fail(SWMI_ERR_INVALID_ARGUMENT, "shard %d of %d (or a NULL output)", shard, n_shards);
    shard_bounds(n, shard, n_shards, lo, hi);
    return SWMI_OK;
}

int swmi_score_batch_multi(const uint8_t *seq1s, const uint8_t *seq2s, size_t n, const int8_t score_matrix[16],
                           int8_t gap_penalty, int32_t *scores)
{
    return multi_host(seq1s, seq2s, n, score_matrix, gap_penalty, scores, false);
}

int swmi_score_batch_packed_multi(const uint8_t *seq1s_packed, const uint8_t *seq2s_packed, size_t n,
                                  const int8_t score_matrix[16], int8_t gap_penalty, int32_t *scores)
{
    return multi_host(seq1s_packed, seq2s_packed, n, score_matrix, gap_penalty, scores, true);
}

int swmi_sharded_create(size_t n, int packed, swmi_sharded_batch **out)
{
    if (!out) return fail(SWMI_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    const int G = num_contexts();
    if (G == 0) return fail(SWMI_ERR_NOT_INITIALIZED, "swmi_init_all() has not been called (or failed)");
    if (n / size_t(G) + 1 > kMaxLaunchPairs) return fail(SWMI_ERR_INVALID_ARGUMENT, "n = %zu is too large for %d GPUs", n, G);
    swmi_sharded_batch *b = new (std::nothrow) swmi_sharded_batch;
    if (!b) return fail(SWMI_ERR_INVALID_ARGUMENT, "out of host memory");
    b->n = n;
    b->packed = packed != 0;
    b->piece = knobs().gather_piece;                           // rehearsal: the ragged RCCL path even with equal shards / one rank
    b->equal_shards = n % size_t(G) == 0 && n > 0 && b->piece == 0;
    b->parts.resize(G);
    const size_t stride = b->packed ? SWMI_PACKED_LEN : kSeq;
    hipError_t e = hipSuccess;
    for (int g = 0; g < G && e == hipSuccess; ++g) {
        auto &p = b->parts[g];
        p.ctx = context_ref(g);
        p.device = p.ctx->device;
        shard_bounds(n, g, G, &p.lo, &p.hi);
        const size_t m = p.hi - p.lo ? p.hi - p.lo : 1;
        e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p.d1), m * stride);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p.d2), m * stride);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p.d_scores), m * sizeof(int32_t));
    }
    if (e != hipSuccess) {
        swmi_sharded_destroy(b);
        return fail(SWMI_ERR_HIP, "swmi_sharded_create: %s", hipGetErrorString(e));
    }
    *out = b;
    return SWMI_OK;
}

int swmi_sharded_destroy(swmi_sharded_batch *b)
{
    if (!b) return SWMI_OK;
    // valid after swmi_shutdown() too: the batch owns its streams, buffers and communicators; only the device ordinal is needed
    for (auto &p : b->parts) {
        if (p.device < 0) continue;
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
    }
    for (ncclComm_t c : b->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    for (auto &p : b->parts) {
        if (p.device < 0) continue;
        (void)hipSetDevice(p.device);
        for (hipEvent_t ev : p.ev) (void)hipEventDestroy(ev);
        (void)hipFree(p.d1); (void)hipFree(p.d2); (void)hipFree(p.d_scores); (void)hipFree(p.d_gathered);
        if (p.stream) (void)hipStreamDestroy(p.stream);
    }
    delete b;
    return SWMI_OK;
}

int swmi_sharded_generate(swmi_sharded_batch *b, uint64_t seed, uint64_t first_pair)
{
    if (!b) return fail(SWMI_ERR_INVALID_ARGUMENT, "batch is NULL");
    if (b->packed) return fail(SWMI_ERR_INVALID_ARGUMENT, "the generator writes one base per byte; create the batch with packed = 0");
    if (batch_alive(b) != SWMI_OK) return last_status();
    for (auto &p : b->parts) {
        if (p.hi == p.lo) continue;
        SWMI_HIP_TRY(hipSetDevice(p.device));
        SWMI_HIP_TRY(swmi::launch_generate(p.d1, p.d2, p.hi - p.lo, seed, first_pair + p.lo, p.stream));
    }
    return SWMI_OK;
}

int swmi_sharded_upload(swmi_sharded_batch *b, const uint8_t *seq1s, const uint8_t *seq2s)
{
    if (!b || !seq1s || !seq2s) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL argument");
    if (batch_alive(b) != SWMI_OK) return last_status();
    const size_t stride = b->packed ? SWMI_PACKED_LEN : kSeq;
    hipError_t e = hipSuccess;
    for (auto &p : b->parts) {
        if (p.hi == p.lo || e != hipSuccess) continue;
        e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipMemcpyAsync(p.d1, seq1s + p.lo * stride, (p.hi - p.lo) * stride, hipMemcpyHostToDevice, p.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(p.d2, seq2s + p.lo * stride, (p.hi - p.lo) * stride, hipMemcpyHostToDevice, p.stream);
    }
    const int rc = wait_all(b);         // the caller's arrays are free again on return, also after a failure
    if (e != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_sharded_upload: %s", hipGetErrorString(e));
    return rc;
}

int swmi_sharded_score(swmi_sharded_batch *b, const int8_t score_matrix[16], int8_t gap_penalty, int gather)
{
    if (!b) return fail(SWMI_ERR_INVALID_ARGUMENT, "batch is NULL");
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (batch_alive(b) != SWMI_OK) return last_status();
    rc = prepare_gather(b, gather);
    if (rc != SWMI_OK) return rc;
    return score_once(b, score_matrix, gap_penalty, gather, -1);
}

int swmi_sharded_wait(swmi_sharded_batch *b)
{
    if (!b) return fail(SWMI_ERR_INVALID_ARGUMENT, "batch is NULL");
    if (batch_alive(b) != SWMI_OK) return last_status();
    return wait_all(b);
}

int swmi_sharded_scores_host(swmi_sharded_batch *b, int32_t *scores)
{
    if (!b || (!scores && b->n)) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL argument");
    if (batch_alive(b) != SWMI_OK) return last_status();
    hipError_t e = hipSuccess;
    for (auto &p : b->parts) {
        if (p.hi == p.lo || e != hipSuccess) continue;
        e = hipSetDevice(p.device);
        if (e == hipSuccess) e = hipMemcpyAsync(scores + p.lo, p.d_scores, (p.hi - p.lo) * sizeof(int32_t), hipMemcpyDeviceToHost, p.stream);
    }
    const int rc = wait_all(b);
    if (e != hipSuccess) return fail(SWMI_ERR_HIP, "swmi_sharded_scores_host: %s", hipGetErrorString(e));
    return rc;
}

int swmi_sharded_gathered_device(swmi_sharded_batch *b, int index, void **d_scores)
{
    if (!b || !d_scores || index < 0 || index >= (int)b->parts.size()) return fail(SWMI_ERR_INVALID_ARGUMENT, "bad argument");
    if (!b->parts[index].d_gathered) return fail(SWMI_ERR_INVALID_ARGUMENT, "GPU index %d holds no gathered vector (no gather ran to it)", index);
    *d_scores = b->parts[index].d_gathered;
    return SWMI_OK;
}

int swmi_sharded_gather_backend(swmi_sharded_batch *b)
{
    if (!b) return fail(SWMI_ERR_INVALID_ARGUMENT, "batch is NULL");
    return !b->rccl_decided ? 0 : b->rccl_usable ? 2 : 1;
}

int swmi_sharded_gather_note(swmi_sharded_batch *b, char *text, size_t text_len)
{
    if (!b || !text || text_len == 0) return fail(SWMI_ERR_INVALID_ARGUMENT, "NULL argument");
    snprintf(text, text_len, "%s", b->rccl_why.c_str());
    return SWMI_OK;
}

int swmi_rccl_probe(char *why, size_t why_len)
{
    {
        std::lock_guard<std::mutex> lock(init_mutex());
        if (num_contexts() == 0) read_knobs();          // before any init: take SWMI_RCCL_LIB from the environment now
    }
    std::string reason;
    const bool ok = load_rccl(&reason);
    if (why && why_len) snprintf(why, why_len, "%s", ok ? "" : reason.c_str());
    return ok ? 1 : 0;
}

int swmi_sharded_gathered_host(swmi_sharded_batch *b, int index, int32_t *scores)
{
    void *d = nullptr;
    const int rc = swmi_sharded_gathered_device(b, index, &d);
    if (rc != SWMI_OK) return rc;
    if (!scores && b->n) return fail(SWMI_ERR_INVALID_ARGUMENT, "scores is NULL");
    if (batch_alive(b) != SWMI_OK) return last_status();
    const int rw = wait_all(b);         // the gather into this GPU runs on the OTHER GPUs' streams
    if (rw != SWMI_OK) return rw;
    SWMI_HIP_TRY(hipSetDevice(b->parts[index].device));
    SWMI_HIP_TRY(hipMemcpy(scores, d, b->n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SWMI_OK;
}

int swmi_sharded_time(swmi_sharded_batch *b, const int8_t score_matrix[16], int8_t gap_penalty, int gather, int iters,
                      float *kernel_ms, float *gather_ms, double *wall_ms)
{
    if (!b || iters <= 0 || iters > 100000) return fail(SWMI_ERR_INVALID_ARGUMENT, "batch is NULL or iters outside [1, 100000]");
    int rc = check_params(score_matrix, gap_penalty);
    if (rc != SWMI_OK) return rc;
    if (batch_alive(b) != SWMI_OK) return last_status();
    rc = prepare_gather(b, gather);
    if (rc != SWMI_OK) return rc;
    for (auto &p : b->parts) {
        SWMI_HIP_TRY(hipSetDevice(p.device));
        while (p.ev.size() < 3 * size_t(iters)) {
            hipEvent_t ev = nullptr;
            SWMI_HIP_TRY(hipEventCreate(&ev));
            p.ev.push_back(ev);
        }
    }
    rc = wait_all(b);
    if (rc != SWMI_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters && rc == SWMI_OK; ++it) rc = score_once(b, score_matrix, gap_penalty, gather, it);
    const int rw = wait_all(b);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != SWMI_OK) return rc;
    if (rw != SWMI_OK) return rw;
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count() / iters;
    for (size_t g = 0; g < b->parts.size(); ++g) {
        auto &p = b->parts[g];
        SWMI_HIP_TRY(hipSetDevice(p.device));
        double k = 0, ga = 0;
        for (int it = 0; it < iters; ++it) {
            float a = 0.f, c = 0.f;
            SWMI_HIP_TRY(hipEventElapsedTime(&a, p.ev[3 * size_t(it)], p.ev[3 * size_t(it) + 1]));
            SWMI_HIP_TRY(hipEventElapsedTime(&c, p.ev[3 * size_t(it) + 1], p.ev[3 * size_t(it) + 2]));
            k += a;
            ga += c;
        }
        if (kernel_ms) kernel_ms[g] = float(k / iters);
        if (gather_ms) gather_ms[g] = float(ga / iters);
    }
    return SWMI_OK;
}

}  // extern "C"
