// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a single-process stand-in for librccl (loaded through SWMI_RCCL_LIB) that
// (i) records every call, (ii) CHECKS what RCCL requires of a grouped sequence -- every rank issues the same collectives in
// the same order with the same root and count -- and (iii) moves the bytes at ncclGroupEnd, "device" memory being host
// memory under tests/native/fake_hip.cpp.  FAKE_NCCL_FAIL_INIT=1 makes ncclCommInitAll fail.
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct ncclComm { int rank, nranks; };
namespace {
std::vector<std::string> g_log;
struct Op { int kind; const void *send; void *recv; size_t count; int root; };   // kind 0 = broadcast, 1 = all-gather
std::vector<std::vector<Op>> g_ops;          // per rank, inside a group
int g_depth = 0, g_ranks = 0, g_errors = 0;
void log(const std::string &s) { g_log.push_back(s); }
}  // namespace

extern "C" {
size_t fake_rccl_log_size() { return g_log.size(); }
const char *fake_rccl_log_at(size_t k) { return k < g_log.size() ? g_log[k].c_str() : ""; }
void fake_rccl_log_clear() { g_log.clear(); }
int fake_rccl_errors() { return g_errors; }

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs)
{
    std::string s = "comm_init_all";
    for (int k = 0; k < n; ++k) s += " dev" + std::to_string(devs[k]);
    log(s);
    if (getenv("FAKE_NCCL_FAIL_INIT")) return ncclSystemError;
    g_ranks = n;
    g_ops.assign(n, {});
    for (int k = 0; k < n; ++k) comms[k] = new ncclComm{k, n};
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) { log("comm_destroy rank" + std::to_string(c->rank)); delete c; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }
ncclResult_t ncclGroupStart() { if (g_depth++ == 0) for (auto &v : g_ops) v.clear(); log("group_start"); return ncclSuccess; }
ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t, int root, ncclComm_t c, hipStream_t)
{
    log("broadcast rank" + std::to_string(c->rank) + " root" + std::to_string(root) + " count" + std::to_string(count));
    if (!g_depth) { ++g_errors; return ncclInvalidUsage; }      // several ranks in one thread: only legal inside a group
    g_ops[c->rank].push_back(Op{0, send, recv, count, root});
    return ncclSuccess;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t, ncclComm_t c, hipStream_t)
{
    log("all_gather rank" + std::to_string(c->rank) + " count" + std::to_string(count));
    if (!g_depth) { ++g_errors; return ncclInvalidUsage; }
    g_ops[c->rank].push_back(Op{1, send, recv, count, -1});
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
    log("group_end");
    if (--g_depth > 0) return ncclSuccess;
    // every rank must have queued the same sequence of collectives
    const size_t n_ops = g_ops.empty() ? 0 : g_ops[0].size();
    for (int r = 0; r < g_ranks; ++r)
        if (g_ops[r].size() != n_ops) { ++g_errors; log("ERROR rank" + std::to_string(r) + " queued a different number of collectives"); return ncclInvalidUsage; }
    for (size_t i = 0; i < n_ops; ++i) {
        const Op &o0 = g_ops[0][i];
        for (int r = 0; r < g_ranks; ++r) {
            const Op &o = g_ops[r][i];
            if (o.kind != o0.kind || o.count != o0.count || o.root != o0.root) { ++g_errors; log("ERROR collective " + std::to_string(i) + " differs on rank" + std::to_string(r)); return ncclInvalidUsage; }
        }
        if (o0.kind == 0) {
            const void *src = g_ops[o0.root][i].send;
            for (int r = 0; r < g_ranks; ++r) memmove(g_ops[r][i].recv, src, o0.count * 4);
        } else {
            for (int r = 0; r < g_ranks; ++r)
                for (int q = 0; q < g_ranks; ++q) memmove(static_cast<char *>(g_ops[r][i].recv) + (size_t)q * o0.count * 4, g_ops[q][i].send, o0.count * 4);
        }
    }
    return ncclSuccess;
}
}  // extern "C"
