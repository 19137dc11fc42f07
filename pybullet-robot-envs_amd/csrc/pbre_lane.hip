// pbre_lane.hip -- the Shape32 engine (iCub as simulated, 20 DoF): the lane-group engine of pbre_wide.hip / pbre_wide_impl.hpp with the
// lane-per-env kernels of pbre_lane.hpp on top for the steps of the task envs.  Its own translation unit: the fully unrolled 20-link
// code needs -mllvm -pragma-unroll-threshold far above the default (build.sh), which the other kernels are not compiled with.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <string>

#include "pbre_wide_impl.hpp"

// lane-per-env path of the 20-DoF iCub (pbre_lane.hpp): device definitions of its hooks
#define PBRE_ANY(x) (__any((int)(x)) != 0)
#define PBRE_REG_BARRIER() asm volatile("" ::: "memory")
#define PBRE_LAUNDER(p) asm volatile("" : "+s"(p))
#define PBRE_OPAQUE_I(x) asm volatile("" : "+v"(x))
#define PBRE_OPAQUE_F(x) asm volatile("" : "+v"(x))
#define PBRE_NOUNROLL _Pragma("nounroll")
#define PBRE_LANE_MSTRIDE 64         // M^-1 in wave-private LDS, [entry][lane]
#ifndef PBRE_LANE_MREG
#define PBRE_LANE_MREG 50            // 160 of the 210 entries in LDS (40 KB per wave: four waves per CU), the rest in registers
#endif
#include "pbre_lane.hpp"

namespace pbre {

// ------------------------------------------------------------------ lane-per-env kernels (Shape32 records, TopoICub)
using LaneD = Lane<TopoICub, Shape32>;
using CoreW = Core<DevLanes32, Shape32>;
constexpr int LTPB = 64;             // one wave per block
static_assert((int)CoreW::M_INNER == (int)LaneD::M_INNER && (int)CoreW::M_TGT == (int)LaneD::M_TGT && (int)CoreW::M_ACTION == (int)LaneD::M_ACTION, "mode bits shared by the lane-group and lane-per-env kernels");

__device__ __forceinline__ void wpublish(int env, int c, signed char* __restrict__ cls, int* __restrict__ next_list, int* __restrict__ next_count) {
    cls[env] = (signed char)c;
    if (c) next_list[atomicAdd(next_count, 1)] = env;
}
// Simple envs (no robot collision sphere near the object / table): every env of the batch in natural order, lanes of complex envs idle.
// One wave per SIMD (the kernel may use the whole register file); LDS: the wave's M^-1 entries.
// (the mode bits are a kernel argument: wave-uniform branches, one instantiation of this large kernel)
__global__ __launch_bounds__(LTPB, 1) void kw_lane(const TablesT<Shape32>* __restrict__ T, const Params P, float* __restrict__ state,
                                                   const float* __restrict__ actions, float* __restrict__ out, int n, int act_dim, int ow, int flags, int MODE,
                                                   const float* __restrict__ tgt, const signed char* __restrict__ cls_cur, signed char* __restrict__ cls,
                                                   int* __restrict__ next_list, int* __restrict__ next_count, int* __restrict__ zero_count) {
    __shared__ float lds_m[LaneD::MLDS * LTPB];
    const int env = blockIdx.x * LTPB + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *zero_count = 0;          // the counter the step after this one appends to (idle now)
    if (env >= n || cls_cur[env] != 0) return;
    const int c = LaneD::step(*T, P, state + (size_t)env * Shape32::STATE, (MODE & LaneD::M_ACTION) ? actions + (size_t)env * act_dim : nullptr,
                              (MODE & LaneD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags, P.env_id_base + (unsigned long long)env,
                              (MODE & LaneD::M_TGT) ? tgt + (size_t)env * Shape32::TGT : nullptr, lds_m + threadIdx.x);
    wpublish(env, c, cls, next_list, next_count);
}
// Complex envs over the compacted list: physics by the lane-group kernel (Core::step: all row types, one env per half-wave),
// observation / reward / termination / auto-reset / class of the new state by Lane::finish on the group's first lane.
__global__ __launch_bounds__(WTPB, 3) void kw_list(const TablesT<Shape32>* __restrict__ T, const Params P, float* __restrict__ state,
                                                   const float* __restrict__ actions, float* __restrict__ out, int act_dim, int ow, int flags, int MODE,
                                                   const float* __restrict__ tgt, const int* __restrict__ cur_list, const int* __restrict__ cur_count,
                                                   signed char* __restrict__ cls, int* __restrict__ next_list, int* __restrict__ next_count) {
    constexpr int EPB = WTPB / 32;
    const int PHYS = MODE & (CoreW::M_ACTION | CoreW::M_TGT);
    const int total = *cur_count;
    for (int base = blockIdx.x * EPB; base < total; base += gridDim.x * EPB) {
        const int i = base + (int)(threadIdx.x / 32);
        if (i >= total) break;                      // whole lane group; the wave's other group keeps going
        const int env = cur_list[i];
        float* st = state + (size_t)env * Shape32::STATE;
        CoreW::step(*T, P, st, (MODE & CoreW::M_ACTION) ? actions + (size_t)env * act_dim : nullptr, nullptr, PHYS, flags,
                    (MODE & CoreW::M_TGT) ? tgt + (size_t)env * Shape32::TGT : nullptr, P.env_id_base + (unsigned long long)env, nullptr);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);    // the group's stores are read back by its first lane below
        if ((threadIdx.x & 31u) == 0) {
            float q[LaneD::ND], qd[LaneD::ND];
            PBRE_UNROLL for (int j = 0; j < LaneD::ND; j++) { q[j] = st[j]; qd[j] = st[Shape32::W + j]; }
            LaneD::V3 op; op.x = st[Shape32::LC]; op.y = st[Shape32::LC + 1]; op.z = st[Shape32::LC + 2];
            LaneD::Q4 oq; oq.x = st[Shape32::LC + 3]; oq.y = st[Shape32::LC + 4]; oq.z = st[Shape32::LC + 5]; oq.w = st[Shape32::LC + 6];
            const int c = LaneD::finish(*T, P, st, q, qd, op, oq, (MODE & LaneD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags,
                                        P.env_id_base + (unsigned long long)env);
            wpublish(env, c, cls, next_list, next_count);
        }
    }
}
// Cartesian control: hand-pose update + inverse kinematics -> joint targets, one thread per env
__global__ __launch_bounds__(LTPB) void kw_lane_ik(const TablesT<Shape32>* __restrict__ T, const Params P, float* __restrict__ state,
                                                   const float* __restrict__ actions, float* __restrict__ tgt, int n, int act_dim) {
    const int env = blockIdx.x * LTPB + threadIdx.x;
    if (env >= n) return;
    LaneD::ik_targets(*T, P, state + (size_t)env * Shape32::STATE, actions + (size_t)env * act_dim, tgt + (size_t)env * Shape32::TGT);
}
// class of every env's current state (after reset / set_state / settle steps)
__global__ __launch_bounds__(LTPB) void kw_lane_classify(const TablesT<Shape32>* __restrict__ T, const Params P, const float* __restrict__ state, int n, int flags,
                                                         signed char* __restrict__ cls, int* __restrict__ list, int* __restrict__ count) {
    const int env = blockIdx.x * LTPB + threadIdx.x;
    if (env >= n) return;
    wpublish(env, LaneD::classify_state(*T, P, state + (size_t)env * Shape32::STATE, flags), cls, list, count);
}

// The Shape32 engine with the lane-per-env path on top: class bookkeeping as in pbre_capi.hip (EnvBuf) -- cls[cur] / list[cur] /
// count[ccur] describe the current states, a step writes the other halves; the counters rotate over three ints so that the one the
// step after next appends to is zeroed by a kernel of the current step.
struct WideLane : WideImpl<Shape32, DevLanes32> {
    signed char* cls = nullptr;       // [2][n]
    int* list = nullptr;              // [2][n]
    int* count = nullptr;             // [3]
    int cur = 0, ccur = 0;
    bool cls_valid = false, topo_ok = false, enabled = true;
    int n_simd = 1024;
    ~WideLane() override { for (void* p : {(void*)cls, (void*)list, (void*)count}) if (p) (void)hipFree(p); }
    bool lane_ok() const override { return enabled && topo_ok && cls != nullptr; }
    void lane_invalidate() override { cls_valid = false; }
    hipError_t lane_alloc() override {
        const char* knob = getenv("PBRE_ICUB_LANE");
        enabled = knob && knob[0] == '1';            // off by default: see DESIGN.md (measured slower than the lane-group kernel so far)
        topo_ok = lane_topo_matches<TopoICub, Shape32>(T);
        if (!enabled || !topo_ok) return hipSuccess;
        hipError_t e;
        if ((e = hipMalloc(&cls, 2 * (size_t)n)) != hipSuccess) return e;
        if ((e = hipMalloc(&list, 2 * (size_t)n * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMalloc(&count, 3 * sizeof(int))) != hipSuccess) return e;
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, device) == hipSuccess) n_simd = pr.multiProcessorCount * 4;
        return hipSuccess;
    }
    void launch_lane_ik(const float* act, hipStream_t s) override {
        hipLaunchKernelGGL(kw_lane_ik, dim3((n + LTPB - 1) / LTPB), dim3(LTPB), 0, s, dT, P, state, act, tgt, n, act_dim);
    }
    void lane_t(int MODE, const float* act, float* out, int flags, hipStream_t s, hipEvent_t* ek) {
        signed char* c_cur = cls + (size_t)cur * n; signed char* c_nxt = cls + (size_t)(cur ^ 1) * n;
        int* l_cur = list + (size_t)cur * n; int* l_nxt = list + (size_t)(cur ^ 1) * n;
        int* k_cur = count + ccur; int* k_nxt = count + (ccur + 1) % 3; int* k_zero = count + (ccur + 2) % 3;
        const int gl = std::min((n + 7) / 8, n_simd);      // persistent blocks of 8 groups; blocks without work exit at once
        hipLaunchKernelGGL(kw_list, dim3(gl), dim3(WTPB), 0, s, dT, P, state, act, out, act_dim, ow, flags, MODE, tgt, l_cur, k_cur, c_nxt, l_nxt, k_nxt);
        if (ek) (void)hipEventRecord(ek[0], s);
        hipLaunchKernelGGL(kw_lane, dim3((n + LTPB - 1) / LTPB), dim3(LTPB), 0, s, dT, P, state, act, out, n, act_dim, ow, flags, MODE, tgt,
                           c_cur, c_nxt, l_nxt, k_nxt, k_zero);
        if (ek) (void)hipEventRecord(ek[1], s);
        cur ^= 1; ccur = (ccur + 1) % 3;
    }
    hipError_t launch_lane_step(int kind, const float* act, float* out, int flags, hipStream_t s, bool timed) override {
        if (!cls_valid) {
            cur = 0; ccur = 0;
            hipError_t e = hipMemsetAsync(count, 0, 3 * sizeof(int), s);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kw_lane_classify, dim3((n + LTPB - 1) / LTPB), dim3(LTPB), 0, s, dT, P, state, n, flags, cls, list, count);
            cls_valid = true;
        }
        hipEvent_t* ek = timed ? ev_k[k_steps % KRING] : nullptr;
        constexpr int OT = LaneD::M_OBS | LaneD::M_TASK;
        switch (kind) {
            case K_STEP_ACT: lane_t(LaneD::M_ACTION | OT, act, out, flags, s, ek); break;
            case K_INNER_ACT: lane_t(LaneD::M_ACTION | LaneD::M_TASK | LaneD::M_INNER, act, out, flags, s, ek); break;
            case K_INNER_TGT: lane_t(LaneD::M_TGT | LaneD::M_TASK | LaneD::M_INNER, act, out, flags, s, ek); break;
            default: lane_t(LaneD::M_TGT | OT, act, out, flags, s, ek); break;
        }
        if (timed) k_steps++;
        return hipGetLastError();
    }
    int lane_info(int* vg, int* complex_now) override {
        hipFuncAttributes fa;
        *vg = hipFuncGetAttributes(&fa, (const void*)kw_lane) == hipSuccess ? fa.numRegs : -1;
        *complex_now = 0;
        if (cls_valid) { (void)hipDeviceSynchronize(); (void)hipMemcpy(complex_now, count + ccur, sizeof(int), hipMemcpyDeviceToHost); }
        return 1;
    }
};

WideEngine* make_lane_engine() { return new WideLane(); }

}  // namespace pbre
