// pbre_comm.hip -- the sharded batch's per-step gather, owned by the context (BASELINE north_star: "shards the env
// batch across the 8 GPUs of one node with a single RCCL gather over xGMI per step to return stacked observations / rewards").
//
// One process per GPU; rank r steps envs [r N / G, (r + 1) N / G) (pbre_config.env_id_base), and pbre_step_gather_device enqueues
//     the step's kernels on the caller's stream  ->  an event  ->  on the ctx's own communication stream: ONE grouped point-to-point
//     exchange (every rank ncclSend's its [n_local, obs_dim + 2] rows to rank 0, rank 0 posts the matching ncclRecv's into the stacked
//     [N, obs_dim + 2] buffer; its own rows are a device copy)  ->  an event the consumer waits for (pbre_gather_wait).
// The caller alternates between two row buffers, so the exchange of step k runs on the xGMI links while the kernels of step k + 1 run on
// the CUs (a buffer is stepped into again only after the exchange that read it: the stream waits for that event).  xGMI is point to
// point -- 7 links per GPU --, so the 7 transfers into rank 0 use 7 different links at once; there is no ring to be bound by.
//
// RCCL is loaded with dlopen when a communicator is first asked for (PBRE_RCCL_LIB, else librccl.so.1 / librccl.so): libpbre.so has no
// link-time dependency on it, and a process that already holds RCCL (torch.distributed's "nccl" backend) passes that library's path and
// shares the one copy.  No torch, no Python in the step loop.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include "../../include/pbre.h"

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    bool load() {
        if (h) return true;
        const char* cand[3] = {getenv("PBRE_RCCL_LIB"), "librccl.so.1", "librccl.so"};
        for (const char* p : cand) { if (p && *p && (h = dlopen(p, RTLD_NOW | RTLD_GLOBAL))) break; }
        if (!h) { err = std::string("RCCL not found (PBRE_RCCL_LIB, librccl.so.1, librccl.so): ") + (dlerror() ? dlerror() : ""); return false; }
#define SYM(f) do { *(void**)&f = dlsym(h, "nccl" #f); if (!f) { err = "librccl lacks nccl" #f; h = nullptr; return false; } } while (0)
        SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(CommCount); SYM(GroupStart); SYM(GroupEnd); SYM(Send); SYM(Recv); SYM(GetVersion); SYM(GetErrorString);
#undef SYM
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_comm_err;
}  // namespace

struct pbre_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0, version = 0, ranks_seen = 0;
    hipStream_t stream = nullptr;                  // the communication stream
    hipEvent_t ev_step[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    long k = 0;
    bool self_p2p = false;                         // PBRE_COMM_SELF_P2P=1: rank 0's own rows also travel through ncclSend / ncclRecv (single-GPU tests)
    std::string err;
};

// one communicator per context ("opaque; owns all device memory, streams, RCCL comm"): kept beside the ctx, which is
// defined in another translation unit, and released by pbre_destroy
static std::map<const pbre_ctx*, pbre_comm*> g_comms;
static std::mutex g_comms_mu;
static pbre_comm* comm_of(const pbre_ctx* c) {
    std::lock_guard<std::mutex> lk(g_comms_mu);
    auto it = g_comms.find(c);
    return it == g_comms.end() ? nullptr : it->second;
}

extern "C" {

const char* pbre_comm_last_error(const pbre_ctx* c) { const pbre_comm* m = comm_of(c); return m ? m->err.c_str() : g_comm_err.c_str(); }

int pbre_comm_unique_id(void* id128) {
    if (!id128) return PBRE_E_ARG;
    static_assert(sizeof(ncclUniqueId) == 128, "the C-ABI hands the id over as 128 opaque bytes");
    if (!g_rccl.load()) { g_comm_err = g_rccl.err; return PBRE_E_UNSUPPORTED; }
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { g_comm_err = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return PBRE_E_DEVICE; }
    std::memcpy(id128, &id, 128);
    return PBRE_OK;
}

static void comm_destroy(pbre_comm* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->comm && g_rccl.h) (void)g_rccl.CommDestroy(m->comm);
    for (auto& e : m->ev_step) if (e) (void)hipEventDestroy(e);
    for (auto& e : m->ev_done) if (e) (void)hipEventDestroy(e);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

__attribute__((visibility("hidden"))) void pbre_comm_release(const pbre_ctx* c) {      // (pbre_destroy; not part of the C-ABI)
    pbre_comm* m = nullptr;
    { std::lock_guard<std::mutex> lk(g_comms_mu); auto it = g_comms.find(c); if (it != g_comms.end()) { m = it->second; g_comms.erase(it); } }
    comm_destroy(m);
}

int pbre_comm_init(pbre_ctx* ctx, const void* id128, int32_t rank, int32_t world) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) { g_comm_err = "pbre_comm_init: bad arguments"; return PBRE_E_ARG; }
    if (comm_of(ctx)) { g_comm_err = "pbre_comm_init: the context already has a communicator"; return PBRE_E_ARG; }
    if (!g_rccl.load()) { g_comm_err = g_rccl.err; return PBRE_E_UNSUPPORTED; }
    int device_id = 0;
    if (pbre_sync(ctx) != PBRE_OK || hipGetDevice(&device_id) != hipSuccess) { g_comm_err = "pbre_comm_init: cannot reach the context's device"; return PBRE_E_DEVICE; }      // (pbre_sync selects it)
    pbre_comm* m = new pbre_comm();
    m->rank = rank; m->world = world; m->device = device_id;
    const char* sp = getenv("PBRE_COMM_SELF_P2P");
    m->self_p2p = sp && sp[0] == '1';
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_comm_err = std::string(#call) + ": " + hipGetErrorString(e_); comm_destroy(m); return PBRE_E_DEVICE; } } while (0)
    CK(hipSetDevice(device_id));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&m->stream, hipStreamNonBlocking, hi));
    for (auto& e : m->ev_step) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : m->ev_done) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
#undef CK
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclResult_t r = g_rccl.CommInitRank(&m->comm, world, id, rank);
    if (r != ncclSuccess) { g_comm_err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); m->comm = nullptr; comm_destroy(m); return PBRE_E_DEVICE; }
    (void)g_rccl.CommCount(m->comm, &m->ranks_seen);
    (void)g_rccl.GetVersion(&m->version);
    { std::lock_guard<std::mutex> lk(g_comms_mu); g_comms[ctx] = m; }
    return PBRE_OK;
}

int pbre_comm_info(const pbre_ctx* ctx, int32_t* info, int32_t n) {
    const pbre_comm* m = comm_of(ctx);
    if (!m || !info) return PBRE_E_ARG;
    const int v[4] = {m->ranks_seen, m->rank, m->version, (int)(m->k & 0x7fffffff)};
    for (int i = 0; i < n; i++) info[i] = i < 4 ? v[i] : 0;
    return PBRE_OK;
}

int pbre_step_gather_device(pbre_ctx* ctx, const float* d_actions, float* d_rows_local, float* d_rows_all, void* stream) {
    pbre_comm* m = comm_of(ctx);
    if (!m) { g_comm_err = "pbre_step_gather_device: call pbre_comm_init first"; return PBRE_E_ARG; }
    if (!d_actions || !d_rows_local || (m->rank == 0 && !d_rows_all)) return PBRE_E_ARG;
    int32_t od = 0, n = 0;
    int rc = pbre_dims(ctx, &od, nullptr, &n);
    if (rc != PBRE_OK) return rc;
    const size_t cnt = (size_t)n * (size_t)(od + 2);
    hipStream_t s = stream == PBRE_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)stream;
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { m->err = std::string(#call) + ": " + hipGetErrorString(e_); return PBRE_E_DEVICE; } } while (0)
#define NK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { m->err = std::string(#call) + ": " + g_rccl.GetErrorString(r_); return PBRE_E_DEVICE; } } while (0)
    HK(hipSetDevice(m->device));
    const int b = (int)(m->k & 1);
    // the exchange that read this parity's row buffers two steps ago must be through before the kernels write them again
    if (m->used[b] && s) HK(hipStreamWaitEvent(s, m->ev_done[b], 0));
    if (m->used[b] && !s) HK(hipEventSynchronize(m->ev_done[b]));
    rc = pbre_step_device(ctx, d_actions, d_rows_local, stream);
    if (rc != PBRE_OK) { m->err = pbre_last_error(ctx); return rc; }
    if (s) { HK(hipEventRecord(m->ev_step[b], s)); HK(hipStreamWaitEvent(m->stream, m->ev_step[b], 0)); }
    else { rc = pbre_sync(ctx); if (rc != PBRE_OK) return rc; }
    // ---- the gather: one grouped point-to-point exchange into rank 0
    const bool own_by_p2p = m->rank == 0 && m->self_p2p;
    if (m->world > 1 || own_by_p2p) {
        NK(g_rccl.GroupStart());
        if (m->rank == 0) {
            for (int r = own_by_p2p ? 0 : 1; r < m->world; r++) NK(g_rccl.Recv(d_rows_all + (size_t)r * cnt, cnt, ncclFloat, r, m->comm, m->stream));
            if (own_by_p2p) NK(g_rccl.Send(d_rows_local, cnt, ncclFloat, 0, m->comm, m->stream));
        } else NK(g_rccl.Send(d_rows_local, cnt, ncclFloat, 0, m->comm, m->stream));
        NK(g_rccl.GroupEnd());
    }
    if (m->rank == 0 && !own_by_p2p && d_rows_all != d_rows_local)
        HK(hipMemcpyAsync(d_rows_all, d_rows_local, cnt * sizeof(float), hipMemcpyDeviceToDevice, m->stream));
    HK(hipEventRecord(m->ev_done[b], m->stream));
    m->used[b] = true;
    m->k++;
#undef HK
#undef NK
    return PBRE_OK;
}

int pbre_gather_wait(pbre_ctx* ctx, void* stream, int32_t host_too) {
    pbre_comm* m = comm_of(ctx);
    if (!m) return PBRE_E_ARG;
    hipStream_t s = stream == PBRE_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)stream;
    (void)hipSetDevice(m->device);
    for (int b = 0; b < 2; b++) {
        if (!m->used[b]) continue;
        hipError_t e = s ? hipStreamWaitEvent(s, m->ev_done[b], 0) : hipSuccess;
        if (e == hipSuccess && (host_too || !s)) e = hipEventSynchronize(m->ev_done[b]);
        if (e != hipSuccess) { m->err = std::string("pbre_gather_wait: ") + hipGetErrorString(e); return PBRE_E_DEVICE; }
    }
    return PBRE_OK;
}

}  // extern "C"
