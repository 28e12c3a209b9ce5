// fake_hip.cpp -- TEST INFRASTRUCTURE ONLY: a recording stand-in for the HIP runtime and for the kernel launchers, so that the
// product's HOST code (swmi_api.cpp, swmi_multi.cpp) can run its multi-GPU logic on a machine with no GPU at all:
// FAKE_HIP_DEVICES "gfx950" devices whose memory is host memory, copies that happen at once -- EXCEPT device-to-host copies on
// a stream, which are held back until that stream (or the device, or an event) is synchronised and read the device buffer
// THEN: a host pipeline that lets later kernels overwrite a score buffer before its copy-back has drained hands back wrong
// scores here, as it would on hardware -- streams and events that only carry an id, and swmi::launch_* stand-ins that write, as the "score" of a pair, the 32-bit number found in the first four
// bytes of its seq1 -- tests/native/multi_fake.cpp stores the global pair index there, so a gathered score vector must read
// 0, 1, 2, ... whatever the sharding, the gather backend and the order of the calls.  Nothing here is linked into libswmi.so.
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../smith-waterman-simd_amd/csrc/swmi_internal.h"

namespace {
std::mutex g_mu;
std::vector<std::string> g_log;
thread_local int t_device = 0;
int g_next_id = 1;
struct Pending { void *dst; const void *src; size_t n; };
struct Handle { int id; int device; std::vector<Pending> pending; };       // a stream (with its held-back D2H copies) or an event
std::vector<Handle *> g_streams;
void log(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void log(const char *fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> l(g_mu);
    g_log.emplace_back(buf);
}
int stream_id(hipStream_t s) { return s ? reinterpret_cast<Handle *>(s)->id : 0; }
void drain(Handle *h)                        // callers hold no lock; a stream is driven by one thread at a time
{
    for (auto &p : h->pending) memmove(p.dst, p.src, p.n);
    h->pending.clear();
}
void drain_all()
{
    std::vector<Handle *> all;
    { std::lock_guard<std::mutex> l(g_mu); all = g_streams; }
    for (Handle *h : all) drain(h);
}
int device_count()
{
    const char *e = getenv("FAKE_HIP_DEVICES");
    return e ? atoi(e) : 3;
}
}  // namespace

// the test driver reads and clears the call log through these
extern "C" size_t fake_hip_log_size() { std::lock_guard<std::mutex> l(g_mu); return g_log.size(); }
extern "C" const char *fake_hip_log_at(size_t k) { std::lock_guard<std::mutex> l(g_mu); return k < g_log.size() ? g_log[k].c_str() : ""; }
extern "C" void fake_hip_log_clear() { std::lock_guard<std::mutex> l(g_mu); g_log.clear(); }

extern "C" {
hipError_t hipGetDeviceCount(int *count) { *count = device_count(); return *count > 0 ? hipSuccess : hipErrorNoDevice; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= device_count()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int d)
{
    memset(p, 0, sizeof *p);
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    snprintf(p->name, sizeof p->name, "fake MI355X #%d", d);
    p->multiProcessorCount = 256; p->warpSize = 64; p->clockRate = 2400000; p->totalGlobalMem = size_t(288) << 30;
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake HIP error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{
    std::lock_guard<std::mutex> l(g_mu);
    Handle *h = new Handle{g_next_id++, t_device, {}};
    g_streams.push_back(h);
    *s = reinterpret_cast<hipStream_t>(h);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    Handle *h = reinterpret_cast<Handle *>(s);
    drain(h);
    {
        std::lock_guard<std::mutex> l(g_mu);
        for (size_t k = 0; k < g_streams.size(); ++k)
            if (g_streams[k] == h) { g_streams.erase(g_streams.begin() + k); break; }
    }
    delete h;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s)
{
    log("dev%d stream_sync stream%d", t_device, stream_id(s));
    if (s) drain(reinterpret_cast<Handle *>(s));
    return hipSuccess;
}
hipError_t hipDeviceSynchronize() { drain_all(); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e)
{
    std::lock_guard<std::mutex> l(g_mu);
    *e = reinterpret_cast<hipEvent_t>(new Handle{g_next_id++, t_device, {}});
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<Handle *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { log("dev%d event_record ev%d stream%d", t_device, reinterpret_cast<Handle *>(e)->id, stream_id(s)); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { drain_all(); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { log("dev%d stream_wait stream%d ev%d", t_device, stream_id(s), reinterpret_cast<Handle *>(e)->id); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s)
{
    log("dev%d memcpy kind%d bytes%zu stream%d", t_device, (int)kind, n, stream_id(s));
    if (kind == hipMemcpyDeviceToHost && s) reinterpret_cast<Handle *>(s)->pending.push_back(Pending{dst, src, n});
    else memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind) { return hipMemcpyAsync(dst, src, n, kind, nullptr); }
hipError_t hipMemcpyPeerAsync(void *dst, int dst_dev, const void *src, int src_dev, size_t n, hipStream_t s)
{
    memmove(dst, src, n);
    log("dev%d memcpy_peer dst_dev%d src_dev%d bytes%zu stream%d", t_device, dst_dev, src_dev, n, stream_id(s));
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < height; ++r) memmove(static_cast<char *>(dst) + r * dpitch, static_cast<const char *>(src) + r * spitch, width);
    return hipSuccess;
}
hipError_t hipDeviceCanAccessPeer(int *can, int a, int b) { *can = a != b; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int peer, unsigned) { log("dev%d enable_peer dev%d", t_device, peer); return hipSuccess; }
}  // extern "C"

// ---- stand-ins for the kernel launchers (sw_kernels.hip / sg_kernels.hip) ----------------------------------------------
namespace swmi {
bool schedule_supported(int L) { return L == 64 || L == 32 || L == 16 || L == 8 || L == 4 || L == 2; }
static void fake_scores(const uint8_t *s1, size_t stride, int32_t *out, size_t n)
{
    for (size_t k = 0; k < n; ++k) memcpy(&out[k], s1 + k * stride, 4);       // "score" = the number in the pair's first four bytes
}
hipError_t launch_score(const LaunchConfig &cfg, const uint8_t *s1, const uint8_t *, int32_t *out, size_t n, const SmRows &, int,
                        bool packed, hipStream_t st)
{
    log("dev%d launch_score n%zu lanes%d stream%d", t_device, n, cfg.lanes_per_alignment, stream_id(st));
    fake_scores(s1, packed ? 32 : 128, out, n);
    return hipSuccess;
}
hipError_t launch_score_one_vs_many(const LaunchConfig &, const uint8_t *s1, const uint8_t *, int32_t *out, size_t n, const SmRows &, int, hipStream_t st)
{
    log("dev%d launch_one_vs_many n%zu stream%d", t_device, n, stream_id(st));
    fake_scores(s1, 128, out, n);
    return hipSuccess;
}
hipError_t launch_generate(uint8_t *s1, uint8_t *s2, size_t n, uint64_t, uint64_t first_pair, hipStream_t st)
{
    log("dev%d launch_generate n%zu first%llu stream%d", t_device, n, (unsigned long long)first_pair, stream_id(st));
    for (size_t k = 0; k < n; ++k) {
        const uint32_t id = (uint32_t)(first_pair + k);
        memset(s1 + 128 * k, 0, 128); memset(s2 + 128 * k, 0, 128);
        memcpy(s1 + 128 * k, &id, 4);
    }
    return hipSuccess;
}
hipError_t launch_banded_affine(const uint8_t *, const uint8_t *, int32_t *, size_t, int, const SmRows &, int, int, hipStream_t, bool, bool) { return hipSuccess; }
int banded_affine_kernel_choice(int, const SmRows &, int, int, bool, bool) { return 0; }
hipError_t launch_unpack(const uint8_t *, uint8_t *, size_t, hipStream_t) { return hipSuccess; }
hipError_t launch_pk_max3_selftest(unsigned long long *, hipStream_t) { return hipSuccess; }
size_t semiglobal_workspace_bytes(size_t n) { return 64 * (n + 1); }
hipError_t launch_semiglobal(const uint8_t *, const uint8_t *, size_t, void *, int32_t *, int32_t *, size_t, uint32_t *, hipStream_t, hipEvent_t, int, SgTuning, unsigned long long *) { return hipSuccess; }
size_t semiglobal_move_words() { return 1040; }
void semiglobal_kernel_names(size_t, int, char *a, size_t an, char *b, size_t bn, SgTuning) { if (a && an) a[0] = 0; if (b && bn) b[0] = 0; }
}  // namespace swmi
