// pbre_comm.hip -- the context-owned RCCL exchanges of the sharded batch (include/pbre.h: pbre_comm_*, pbre_step_gather_device,
// pbre_gather_wait, pbre_scatter_actions_device) on the HIP runtime.  The logic -- dlopen'ed RCCL, one grouped ncclSend / ncclRecv
// exchange per step on the ctx's communication stream, per-buffer reuse protection, error paths -- is pbre_comm_impl.hpp's; this file
// supplies the runtime policy (streams, events, device copies) and the C-ABI entry points.
#include <hip/hip_runtime.h>
#include <string>
#include "pbre_comm_impl.hpp"

namespace {
struct HipRuntime {
    typedef hipStream_t stream_t;
    typedef hipEvent_t event_t;
    static std::string ck(hipError_t e, const char* what) { return e == hipSuccess ? std::string() : std::string(what) + ": " + hipGetErrorString(e); }
    // The ABI's stream argument: a hipStream_t; PBRE_STREAM_LEGACY = HIP's legacy default stream, a real stream as far as events go
    // (hipEventRecord(ev, nullptr) and hipStreamWaitEvent(nullptr, ev) work on it: torch's default stream keeps the asynchronous,
    // overlapped path); NULL = the ctx's own non-blocking stream, which only the ctx can order against: host-synchronised (own).
    static stream_t to_stream(void* abi, bool& own) { own = abi == nullptr; return abi == PBRE_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)abi; }
    static void* raw(stream_t s) { return (void*)s; }      // what ncclSend / ncclRecv take
    static std::string set_device(int d) { return ck(hipSetDevice(d), "hipSetDevice"); }
    static std::string current_device(int* d) { return ck(hipGetDevice(d), "hipGetDevice"); }
    static std::string stream_create_high_priority(stream_t* s) {
        int lo = 0, hi = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(s, hipStreamNonBlocking, hi);
        return ck(e, "hipStreamCreateWithPriority");
    }
    static void stream_destroy(stream_t s) { (void)hipStreamDestroy(s); }
    static std::string stream_sync(stream_t s) { return ck(hipStreamSynchronize(s), "hipStreamSynchronize"); }
    static std::string event_create(event_t* e) { return ck(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreateWithFlags"); }
    static void event_destroy(event_t e) { if (e) (void)hipEventDestroy(e); }
    static std::string event_record(event_t e, stream_t s) { return ck(hipEventRecord(e, s), "hipEventRecord"); }
    static std::string stream_wait(stream_t s, event_t e) { return ck(hipStreamWaitEvent(s, e, 0), "hipStreamWaitEvent"); }
    static std::string event_sync(event_t e) { return ck(hipEventSynchronize(e), "hipEventSynchronize"); }
    static std::string copy_async(void* dst, const void* src, size_t bytes, stream_t s) { return ck(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync"); }
    static int step(pbre_ctx* ctx, const float* act, float* rows, void* abi_stream) { return pbre_step_device(ctx, act, rows, abi_stream); }
};
typedef pbre_comm_detail::Comm<HipRuntime> CommD;
}  // namespace

extern "C" {
__attribute__((visibility("hidden"))) void pbre_comm_release(const pbre_ctx* c) { CommD::release(c); }      // (pbre_destroy; not part of the C-ABI)
const char* pbre_comm_last_error(const pbre_ctx* c) { return CommD::last_error(c); }
int pbre_comm_probe(void) { return CommD::probe(); }
int pbre_comm_unique_id(void* id128) { return CommD::unique_id(id128); }
int pbre_comm_init(pbre_ctx* ctx, const void* id128, int32_t rank, int32_t world) { return CommD::init(ctx, id128, rank, world); }
int pbre_comm_info(const pbre_ctx* ctx, int32_t* info, int32_t n) { return CommD::info(ctx, info, n); }
int pbre_step_gather_device(pbre_ctx* ctx, const float* d_actions, float* d_rows_local, float* d_rows_all, void* stream) {
    return CommD::step_gather(ctx, d_actions, d_rows_local, d_rows_all, stream);
}
int pbre_gather_wait(pbre_ctx* ctx, void* stream, int32_t host_too) { return CommD::gather_wait(ctx, stream, host_too); }
int pbre_scatter_actions_device(pbre_ctx* ctx, const float* d_actions_all, float* d_actions_local, void* stream) {
    return CommD::scatter_actions(ctx, d_actions_all, d_actions_local, stream);
}
}  // extern "C"
