// pbre_comm_impl.hpp -- the sharded batch's per-step exchanges, owned by the context (BASELINE north_star: "shards the
// env batch across the 8 GPUs of one node with a single RCCL gather over xGMI per step to return stacked observations / rewards").
//
// One process per GPU; rank r steps envs [r N / G, (r + 1) N / G) (pbre_config.env_id_base), and pbre_step_gather_device enqueues
//     the step's kernels on the caller's stream  ->  an event  ->  on the ctx's own communication stream: ONE grouped point-to-point
//     exchange (every rank ncclSend's its [n_local, obs_dim + 2] rows to rank 0, rank 0 posts the matching ncclRecv's into the stacked
//     [N, obs_dim + 2] buffer; its own rows are a device copy)  ->  an event the consumer waits for (pbre_gather_wait).
// The caller alternates between two row buffers, so the exchange of step k runs on the xGMI links while the kernels of step k + 1 run on
// the CUs; a row buffer is stepped into again only after the exchange that read it (tracked per buffer POINTER).  xGMI is point to
// point -- 7 links per GPU --, so the 7 transfers into rank 0 use 7 different links at once; there is no ring to be bound by.
// pbre_scatter_actions_device is the way back of a closed loop: rank 0's policy has the actions of all N envs, every rank needs its
// [n_local, act_dim] slice before it can step -- one grouped exchange in the other direction, on the caller's stream order.
//
// RCCL is loaded with dlopen when a communicator is first asked for (PBRE_RCCL_LIB, else librccl.so.1 / librccl.so): libpbre.so has no
// link-time AND no build-time dependency on it -- the handful of types of its C API this file needs are declared below (checked against
// <rccl/rccl.h> by tests/test_capi.py where the header exists) --, and a process that already holds RCCL (torch.distributed's "nccl"
// backend) passes that library's path and shares the one copy.  No torch, no Python in the step loop.
//
// The logic is a template over a small runtime policy so that the SAME source runs in the product (HipRuntime: streams, events, device
// copies; csrc/pbre_comm.hip) and in the CPU lane emulation of the tests (HostRuntime: host buffers, everything synchronous;
// tests/host_emu/emu_capi.cpp), where tests/fake_rccl supplies the ncclSend / ncclRecv of a world of several processes.
#pragma once
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include "../../include/pbre.h"

namespace pbre_comm_detail {

// ---- the part of RCCL's C API (= NCCL's) this file binds, by its ABI
typedef struct pbreNcclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                 // enum ncclResult_t: ncclSuccess = 0
typedef int ncclDataType_t;               // enum ncclDataType_t
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclFloat = 7;   // ncclFloat32
constexpr ncclDataType_t ncclUint8 = 1;

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, void*) = nullptr;      // (last argument: hipStream_t)
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, void*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    std::mutex mu;
    bool load() {
        std::lock_guard<std::mutex> lk(mu);        // (two threads may create their contexts' communicators at once: MultiEngine)
        if (h) return true;
        const char* cand[3] = {getenv("PBRE_RCCL_LIB"), "librccl.so.1", "librccl.so"};
        void* lib = nullptr;
        for (const char* p : cand) { if (p && *p && (lib = dlopen(p, RTLD_NOW | RTLD_GLOBAL))) break; }
        if (!lib) { const char* de = dlerror(); err = std::string("RCCL not found (PBRE_RCCL_LIB, librccl.so.1, librccl.so): ") + (de ? de : ""); return false; }
#define PBRE_SYM(f) do { *(void**)&f = dlsym(lib, "nccl" #f); if (!f) { err = "the RCCL library lacks nccl" #f; return false; } } while (0)
        PBRE_SYM(GetUniqueId); PBRE_SYM(CommInitRank); PBRE_SYM(CommDestroy); PBRE_SYM(CommCount); PBRE_SYM(GroupStart); PBRE_SYM(GroupEnd);
        PBRE_SYM(Send); PBRE_SYM(Recv); PBRE_SYM(GetVersion); PBRE_SYM(GetErrorString);
#undef PBRE_SYM
        h = lib;
        return true;
    }
};
inline Rccl& rccl() { static Rccl r; return r; }
inline std::string& thread_err() { static thread_local std::string e; return e; }

// RT: the runtime policy --
//   stream_t, event_t; to_stream(void* abi_stream, bool& own) -> stream_t (own: NULL = the ctx's own stream, host-synchronised)
//   raw(stream) -> the void* ncclSend / ncclRecv take; set_device / current_device; stream_create_high_priority / stream_destroy / stream_sync
//   event_create / event_destroy / event_record(ev, s) / stream_wait(s, ev) / event_sync(ev)
//   copy_async(dst, src, bytes, s); step(ctx, actions, rows, abi_stream)
//   each returning "" on success or an error text
template <class RT>
struct Comm {
    typedef typename RT::stream_t stream_t;
    typedef typename RT::event_t event_t;
    static constexpr int NBUF = 4;                 // row buffers remembered (the caller alternates between two)
    struct State {
        ncclComm_t comm = nullptr;
        int rank = 0, world = 1, device = 0, version = 0, ranks_seen = 0;
        stream_t stream = stream_t();              // the communication stream
        bool have_stream = false;
        event_t ev_step = event_t(), ev_act = event_t();
        struct Buf { const void* rows = nullptr; event_t done = event_t(); bool used = false; long last = -1; } buf[NBUF];
        bool have_events = false;
        long k = 0, scatters = 0;
        bool self_p2p = false;                     // PBRE_COMM_SELF_P2P=1: rank 0's own rows also travel through ncclSend / ncclRecv (single-GPU tests)
        bool dead = false;                         // an exchange failed inside its group: the communicator is unusable
        std::string err;
    };
    static std::map<const pbre_ctx*, State*>& table() { static std::map<const pbre_ctx*, State*> t; return t; }
    static std::mutex& table_mu() { static std::mutex m; return m; }
    static State* of(const pbre_ctx* c) {
        std::lock_guard<std::mutex> lk(table_mu());
        auto it = table().find(c);
        return it == table().end() ? nullptr : it->second;
    }
    static const char* last_error(const pbre_ctx* c) { const State* m = of(c); return m ? m->err.c_str() : thread_err().c_str(); }

    // load probe: dlopen + symbol resolution only -- no ncclGetUniqueId, whose bootstrap root thread and listening socket a rank
    // other than 0 would start for nothing
    static int probe() {
        Rccl& R = rccl();
        if (!R.load()) { thread_err() = R.err; return PBRE_E_UNSUPPORTED; }
        return PBRE_OK;
    }
    static int unique_id(void* id128) {
        if (!id128) { thread_err() = "pbre_comm_unique_id: null argument"; return PBRE_E_ARG; }
        Rccl& R = rccl();
        if (!R.load()) { thread_err() = R.err; return PBRE_E_UNSUPPORTED; }
        ncclUniqueId id;
        const ncclResult_t r = R.GetUniqueId(&id);
        if (r != ncclSuccess) { thread_err() = std::string("ncclGetUniqueId: ") + R.GetErrorString(r); return PBRE_E_DEVICE; }
        std::memcpy(id128, &id, 128);
        return PBRE_OK;
    }
    static void destroy(State* m) {
        if (!m) return;
        (void)RT::set_device(m->device);
        if (m->have_stream) (void)RT::stream_sync(m->stream);
        if (m->comm && rccl().h) (void)rccl().CommDestroy(m->comm);
        if (m->have_events) { RT::event_destroy(m->ev_step); RT::event_destroy(m->ev_act); for (auto& b : m->buf) RT::event_destroy(b.done); }
        if (m->have_stream) RT::stream_destroy(m->stream);
        delete m;
    }
    static void release(const pbre_ctx* c) {       // pbre_destroy
        State* m = nullptr;
        { std::lock_guard<std::mutex> lk(table_mu()); auto it = table().find(c); if (it != table().end()) { m = it->second; table().erase(it); } }
        destroy(m);
    }
    static int init(pbre_ctx* ctx, const void* id128, int32_t rank, int32_t world) {
        std::string& te = thread_err();
        if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) { te = "pbre_comm_init: bad arguments"; return PBRE_E_ARG; }
        if (of(ctx)) { te = "pbre_comm_init: the context already has a communicator"; return PBRE_E_ARG; }
        Rccl& R = rccl();
        if (!R.load()) { te = R.err; return PBRE_E_UNSUPPORTED; }
        int device_id = 0;
        if (pbre_sync(ctx) != PBRE_OK || !RT::current_device(&device_id).empty()) { te = "pbre_comm_init: cannot reach the context's device"; return PBRE_E_DEVICE; }      // (pbre_sync selects it)
        State* m = new State();
        m->rank = rank; m->world = world; m->device = device_id;
        const char* sp = getenv("PBRE_COMM_SELF_P2P");
        m->self_p2p = sp && sp[0] == '1';
        std::string e = RT::set_device(device_id);
        if (e.empty()) { e = RT::stream_create_high_priority(&m->stream); m->have_stream = e.empty(); }
        if (e.empty()) {
            e = RT::event_create(&m->ev_step);
            if (e.empty()) e = RT::event_create(&m->ev_act);
            for (auto& b : m->buf) if (e.empty()) e = RT::event_create(&b.done);
            m->have_events = e.empty();
        }
        if (!e.empty()) { te = e; m->have_events = false; destroy(m); return PBRE_E_DEVICE; }
        ncclUniqueId id;
        std::memcpy(&id, id128, 128);
        const ncclResult_t r = R.CommInitRank(&m->comm, world, id, rank);
        if (r != ncclSuccess) { te = std::string("ncclCommInitRank: ") + R.GetErrorString(r); m->comm = nullptr; destroy(m); return PBRE_E_DEVICE; }
        (void)R.CommCount(m->comm, &m->ranks_seen);
        (void)R.GetVersion(&m->version);
        { std::lock_guard<std::mutex> lk(table_mu()); table()[ctx] = m; }
        return PBRE_OK;
    }
    static int info(const pbre_ctx* ctx, int32_t* out, int32_t n) {
        const State* m = of(ctx);
        if (!m || !out) { thread_err() = "pbre_comm_info: no communicator / null argument"; return PBRE_E_ARG; }
        const int v[5] = {m->ranks_seen, m->rank, m->version, (int)(m->k & 0x7fffffff), (int)(m->scatters & 0x7fffffff)};
        for (int i = 0; i < n; i++) out[i] = i < 5 ? v[i] : 0;
        return PBRE_OK;
    }
    static int fail(State* m, int code, const std::string& what) { m->err = what; return code; }

    // One grouped exchange on stream `cs`: `post` issues the ncclSend / ncclRecv calls.  A failure INSIDE the group still closes it
    // (an open group would swallow every later call), and marks the communicator dead.
    template <class Post>
    static int grouped(State* m, Post&& post) {
        Rccl& R = rccl();
        ncclResult_t r = R.GroupStart();
        if (r != ncclSuccess) return fail(m, PBRE_E_DEVICE, std::string("ncclGroupStart: ") + R.GetErrorString(r));
        std::string what;
        r = post(what);
        const ncclResult_t re = R.GroupEnd();
        if (r != ncclSuccess) { m->dead = true; return fail(m, PBRE_E_DEVICE, what + ": " + R.GetErrorString(r)); }
        if (re != ncclSuccess) { m->dead = true; return fail(m, PBRE_E_DEVICE, std::string("ncclGroupEnd: ") + R.GetErrorString(re)); }
        return PBRE_OK;
    }

    static int step_gather(pbre_ctx* ctx, const float* d_actions, float* d_rows_local, float* d_rows_all, void* stream) {
        State* m = of(ctx);
        if (!m) { thread_err() = "pbre_step_gather_device: call pbre_comm_init first"; return PBRE_E_ARG; }
        if (m->dead) return fail(m, PBRE_E_DEVICE, "the communicator is unusable after a failed exchange: " + m->err);
        if (!d_actions || !d_rows_local || (m->rank == 0 && !d_rows_all)) return fail(m, PBRE_E_ARG, "pbre_step_gather_device: null buffer");
        int32_t od = 0, n = 0;
        int rc = pbre_dims(ctx, &od, nullptr, &n);
        if (rc != PBRE_OK) return fail(m, rc, "pbre_dims failed");
        const size_t cnt = (size_t)n * (size_t)(od + 2);
        bool own = false;
        const stream_t s = RT::to_stream(stream, own);      // own: NULL = the ctx's own non-blocking stream, ordered through the host
        Rccl& R = rccl();
        std::string e = RT::set_device(m->device);
        if (!e.empty()) return fail(m, PBRE_E_DEVICE, e);
        // the exchange that last read THIS row buffer must be through before the kernels write it again (tracked per pointer, so a caller
        // that interleaves plain pbre_step_device calls or changes its alternation is still safe as long as it goes through here)
        typename State::Buf* slot = nullptr;
        for (auto& b : m->buf) if (b.used && b.rows == d_rows_local) slot = &b;
        if (slot) { e = own ? RT::event_sync(slot->done) : RT::stream_wait(s, slot->done); if (!e.empty()) return fail(m, PBRE_E_DEVICE, e); }
        else {        // a free slot, else the least recently used one (its exchange is waited for first: its event is about to be re-recorded)
            for (auto& b : m->buf) if (!b.used) { slot = &b; break; }
            if (!slot) { slot = &m->buf[0]; for (auto& b : m->buf) if (b.last < slot->last) slot = &b; e = RT::event_sync(slot->done); if (!e.empty()) return fail(m, PBRE_E_DEVICE, e); }
        }
        rc = RT::step(ctx, d_actions, d_rows_local, stream);
        if (rc != PBRE_OK) return fail(m, rc, pbre_last_error(ctx));
        if (!own) { e = RT::event_record(m->ev_step, s); if (e.empty()) e = RT::stream_wait(m->stream, m->ev_step); if (!e.empty()) return fail(m, PBRE_E_DEVICE, e); }
        else { rc = pbre_sync(ctx); if (rc != PBRE_OK) return fail(m, rc, pbre_last_error(ctx)); }
        // ---- the gather: one grouped point-to-point exchange into rank 0
        const bool own_by_p2p = m->rank == 0 && m->self_p2p;
        if (m->world > 1 || own_by_p2p) {
            rc = grouped(m, [&](std::string& what) -> ncclResult_t {
                ncclResult_t r = ncclSuccess;
                if (m->rank == 0) {
                    for (int q = own_by_p2p ? 0 : 1; q < m->world && r == ncclSuccess; q++) { r = R.Recv(d_rows_all + (size_t)q * cnt, cnt, ncclFloat, q, m->comm, RT::raw(m->stream)); what = "ncclRecv"; }
                    if (own_by_p2p && r == ncclSuccess) { r = R.Send(d_rows_local, cnt, ncclFloat, 0, m->comm, RT::raw(m->stream)); what = "ncclSend"; }
                } else { r = R.Send(d_rows_local, cnt, ncclFloat, 0, m->comm, RT::raw(m->stream)); what = "ncclSend"; }
                return r;
            });
            if (rc != PBRE_OK) return rc;
        }
        if (m->rank == 0 && !own_by_p2p && d_rows_all != d_rows_local) {
            e = RT::copy_async(d_rows_all, d_rows_local, cnt * sizeof(float), m->stream);
            if (!e.empty()) return fail(m, PBRE_E_DEVICE, e);
        }
        e = RT::event_record(slot->done, m->stream);
        if (!e.empty()) return fail(m, PBRE_E_DEVICE, e);
        slot->rows = d_rows_local; slot->used = true; slot->last = m->k;
        m->k++;
        return PBRE_OK;
    }

    // make `stream` (host_too: and the host) wait for every exchange enqueued so far
    static int gather_wait(pbre_ctx* ctx, void* stream, int32_t host_too) {
        State* m = of(ctx);
        if (!m) { thread_err() = "pbre_gather_wait: no communicator"; return PBRE_E_ARG; }
        bool own = false;
        const stream_t s = RT::to_stream(stream, own);
        std::string e = RT::set_device(m->device);
        for (auto& b : m->buf) {
            if (!b.used || !e.empty()) continue;
            if (!own) e = RT::stream_wait(s, b.done);
            if (e.empty() && (host_too || own)) e = RT::event_sync(b.done);
        }
        if (!e.empty()) return fail(m, PBRE_E_DEVICE, "pbre_gather_wait: " + e);
        return PBRE_OK;
    }

    // rank 0: d_actions_all [world * n_local, act_dim]; every rank: d_actions_local [n_local, act_dim].  In `stream` order on both ends.
    static int scatter_actions(pbre_ctx* ctx, const float* d_actions_all, float* d_actions_local, void* stream) {
        State* m = of(ctx);
        if (!m) { thread_err() = "pbre_scatter_actions_device: call pbre_comm_init first"; return PBRE_E_ARG; }
        if (m->dead) return fail(m, PBRE_E_DEVICE, "the communicator is unusable after a failed exchange: " + m->err);
        if (!d_actions_local || (m->rank == 0 && !d_actions_all)) return fail(m, PBRE_E_ARG, "pbre_scatter_actions_device: null buffer");
        int32_t ad = 0, n = 0;
        int rc = pbre_dims(ctx, nullptr, &ad, &n);
        if (rc != PBRE_OK) return fail(m, rc, "pbre_dims failed");
        const size_t cnt = (size_t)n * (size_t)ad;
        bool own = false;
        const stream_t s = RT::to_stream(stream, own);
        if (own) return fail(m, PBRE_E_ARG, "pbre_scatter_actions_device: pass the stream the actions are produced / consumed on (or PBRE_STREAM_LEGACY)");
        Rccl& R = rccl();
        std::string e = RT::set_device(m->device);
        // communication stream after the producer of the actions (rank 0: the policy) and after earlier readers of the local buffer
        if (e.empty()) e = RT::event_record(m->ev_act, s);
        if (e.empty()) e = RT::stream_wait(m->stream, m->ev_act);
        if (!e.empty()) return fail(m, PBRE_E_DEVICE, e);
        const bool own_by_p2p = m->rank == 0 && m->self_p2p;
        if (m->world > 1 || own_by_p2p) {
            rc = grouped(m, [&](std::string& what) -> ncclResult_t {
                ncclResult_t r = ncclSuccess;
                if (m->rank == 0) {
                    for (int q = own_by_p2p ? 0 : 1; q < m->world && r == ncclSuccess; q++) { r = R.Send(d_actions_all + (size_t)q * cnt, cnt, ncclFloat, q, m->comm, RT::raw(m->stream)); what = "ncclSend"; }
                    if (own_by_p2p && r == ncclSuccess) { r = R.Recv(d_actions_local, cnt, ncclFloat, 0, m->comm, RT::raw(m->stream)); what = "ncclRecv"; }
                } else { r = R.Recv(d_actions_local, cnt, ncclFloat, 0, m->comm, RT::raw(m->stream)); what = "ncclRecv"; }
                return r;
            });
            if (rc != PBRE_OK) return rc;
        }
        if (m->rank == 0 && !own_by_p2p && d_actions_all != d_actions_local) {
            e = RT::copy_async(d_actions_local, d_actions_all, cnt * sizeof(float), m->stream);
            if (!e.empty()) return fail(m, PBRE_E_DEVICE, e);
        }
        // ... and the caller's stream (the step that consumes the actions) after the exchange
        e = RT::event_record(m->ev_act, m->stream);
        if (e.empty()) e = RT::stream_wait(s, m->ev_act);
        if (!e.empty()) return fail(m, PBRE_E_DEVICE, e);
        m->scatters++;
        return PBRE_OK;
    }
};

}  // namespace pbre_comm_detail
