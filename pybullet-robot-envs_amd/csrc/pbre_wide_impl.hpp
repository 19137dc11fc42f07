// pbre_wide_impl.hpp -- kernels and the shape-specific half of the wide lane-group engine (pbre_wide.hip), as templates over
// the lane-group shape S and the lane backend L.  Included by the translation units that instantiate them: pbre_wide.hip
// (Shape32 / Shape64) and pbre_hands.hip (Shape128, the iCub with hands).
#pragma once
#include <hip/hip_runtime.h>
#include <string>

#define PBRE_HD __device__ __forceinline__
#define PBRE_UNROLL _Pragma("unroll")
#define PBRE_OPAQUE(p) asm volatile("" : "+s"(p))
#define PBRE_COUNT_BAD(p) atomicAdd((p), 1)
#include "pbre_host.hpp"
#include "lanes_device.hpp"
#include "pbre_core.hpp"
#include "pbre_objstep.hpp"
#include "pbre_wide.hpp"

namespace pbre {

constexpr int WTPB = 256;                        // 4 independent waves per block
static_assert(WTPB == 64 * DevLanes128::WPB, "DevLanes128 sizes its per-wave LDS regions for this block size");
template <class S> constexpr int phys_lanes() { return S::W > 64 ? 64 : S::W; }
static_assert(Shape32::LC >= 16, "DevLanes32::sum_obj assumes the object lanes lie in the upper 16-lane row of the half-wave");   // physical lanes of one env group (Shape128: two virtual lanes each)
struct MotorCmd { int n; int dof[64]; float target[64]; float kp, fscale, vmax; };     // pbre_set_motors, by value

// RT: pbre_physics.solver_residual_threshold > 0 (Core::step<RT>; objv is null then: the exit test is over all rows of an env)
template <class S, class L, int MODE, bool RT = false>
__global__ __launch_bounds__(WTPB, S::W > 64 ? 2 : 3) void kw_step(const TablesT<S>* __restrict__ T, const Params P, float* __restrict__ state,
                                                   const float* __restrict__ actions, float* __restrict__ out, int n, int act_dim, int ow,
                                                   int flags, const float* __restrict__ tgt, const float* __restrict__ objv) {
    using C = Core<L, S>;
    constexpr int EPB = WTPB / phys_lanes<S>();
    const int env = blockIdx.x * EPB + (int)(threadIdx.x / phys_lanes<S>());
    if (env >= n) return;                           // whole lane group; a partially filled wave keeps running its other group
    C::template step<RT>(*T, P, state + (size_t)env * S::STATE, (MODE & C::M_ACTION) ? actions + (size_t)env * act_dim : nullptr,
            (MODE & C::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags, ((MODE & C::M_TGT) || S::MREC) ? tgt + (size_t)env * S::TGT : nullptr,
            P.env_id_base + (unsigned long long)env, objv ? objv + (size_t)env * S::W : nullptr, nullptr, (RT && P.sweeps) ? P.sweeps + env : nullptr);
}
// The object's half of the step for every env, one thread per env (pbre_objstep.hpp): twist after a step without robot-object contact,
// into the object lanes of the env's side record.  kw_step takes it where its collision detection finds no such contact.
template <class S>
PBRE_HD void obj_env(const Params& P, const float* __restrict__ state, float* __restrict__ objv, int e) {
    const float* st = state + (size_t)e * S::STATE;
    float pose[7], tw[6], o[6];
    PBRE_UNROLL for (int k = 0; k < 7; k++) pose[k] = st[S::LC + k];
    PBRE_UNROLL for (int k = 0; k < 6; k++) tw[k] = st[S::W + S::LC + k];
    ObjStep::run(P, pose, tw, o);
    PBRE_UNROLL for (int k = 0; k < 6; k++) objv[(size_t)e * S::W + S::LC + k] = o[k];
}
template <class S>
__global__ __launch_bounds__(64) void kw_obj(const Params P, const float* __restrict__ state, float* __restrict__ objv, int n) {
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e < n) obj_env<S>(P, state, objv, e);
}
// Cartesian control: joint targets of the step (lane groups, blocks [0, ik_blocks)).  The object solve of the same step does not
// depend on them, so it rides along as extra blocks (one thread per env) instead of a launch of its own ahead of kw_step: its 45 us
// dependency chain disappears under this kernel's 100 us.
template <class S, class L, bool RESET>
__global__ __launch_bounds__(WTPB) void kw_ik(const TablesT<S>* __restrict__ T, const Params P, float* __restrict__ state,
                                              const float* __restrict__ actions, float* __restrict__ tgt, int n, int act_dim,
                                              int ik_blocks, float* __restrict__ objv) {
    if ((int)blockIdx.x >= ik_blocks) {
        const int e = ((int)blockIdx.x - ik_blocks) * WTPB + (int)threadIdx.x;
        if (e < n) obj_env<S>(P, state, objv, e);
        return;
    }
    constexpr int EPB = WTPB / phys_lanes<S>();
    const int env = blockIdx.x * EPB + (int)(threadIdx.x / phys_lanes<S>());
    if (env >= n) return;
    Core<L, S>::ik_targets(*T, P, state + (size_t)env * S::STATE, RESET ? nullptr : actions + (size_t)env * act_dim, tgt + (size_t)env * S::TGT, RESET);
}
template <class S, class L, int MODE>
__global__ __launch_bounds__(WTPB) void kw_observe(const TablesT<S>* __restrict__ T, const Params P, float* __restrict__ state,
                                                   float* __restrict__ out, int n, int ow) {
    using C = Core<L, S>;
    constexpr int EPB = WTPB / phys_lanes<S>();
    const int env = blockIdx.x * EPB + (int)(threadIdx.x / phys_lanes<S>());
    if (env >= n) return;
    float* st = state + (size_t)env * S::STATE;
    const auto Q = L::load(st), V = L::load(st + S::W), X = L::loadm(st + 2 * S::W, L::lti(L::lane(), 16));
    C::observe(*T, P, st, Q, V, X, (MODE & C::M_OBS) ? out + (size_t)env * ow : nullptr, MODE);
}
template <class S, class L>
__global__ void kw_init(const TablesT<S>* __restrict__ T, const Params P, float* __restrict__ state,
                        const unsigned long long* __restrict__ ids, const unsigned* __restrict__ ep, int cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) Core<L, S>::init_state(*T, P, ids[i], ep[i], state + (size_t)i * S::STATE);
}
// motor record of a freshly reset env (iCubHandsEnv.reset, icub_env_with_hands.py:108-121): initial positions, gain 0.2, default force
template <class S>
__global__ void kw_mrec_init(const TablesT<S>* __restrict__ T, float* __restrict__ tgt, int cnt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / S::W, l = t % S::W;
    if (i >= cnt) return;
    float* m = tgt + (size_t)i * S::TGT;
    m[l] = T->home[l]; m[S::W + l] = T->kp_hold[l]; m[2 * S::W + l] = 1.f; m[3 * S::W + l] = 0.f;
}
// joint-control half of apply_action without the simulation step (icub_env.py:341-361): clipped absolute targets, gain 0.5, default force
template <class S>
__global__ void kw_cmd_joints(const TablesT<S>* __restrict__ T, float* __restrict__ tgt, const float* __restrict__ actions, int n, int act_dim, float vmax) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t / 64, l = t % 64;
    if (e >= n || l >= S::W) return;              // (64 threads per env cover the <= 64 DoF lanes; the Panda's shape has 32 lanes)
    const int k = T->act_idx[l];
    if (k < 0) return;
    float* m = tgt + (size_t)e * S::TGT;
    m[l] = fminf(fmaxf(actions[(size_t)e * act_dim + k], T->lower[l]), T->upper[l]); m[S::W + l] = T->kp_act[l]; m[2 * S::W + l] = 1.f; m[3 * S::W + l] = vmax;
}
template <class S>
__global__ void kw_set_motors(float* __restrict__ tgt, int n, const MotorCmd cmd, const unsigned char* __restrict__ mask) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = t / 64, k = t % 64;
    if (e >= n || k >= cmd.n || (mask && !mask[e])) return;
    float* m = tgt + (size_t)e * S::TGT;
    m[cmd.dof[k]] = cmd.target[k]; m[S::W + cmd.dof[k]] = cmd.kp; m[2 * S::W + cmd.dof[k]] = cmd.fscale; m[3 * S::W + cmd.dof[k]] = cmd.vmax;
}
// pbre_reset_snapshot (see pbre_capi.hip k_snapshot_reset); iCub push: the initial distances X[12], X[13] of the re-initialised envs are
// taken from `initd`, a copy of the batch on which the M_INITD observation was run
template <class S, class L>
__global__ void kw_snapshot_reset(const TablesT<S>* __restrict__ T, const Params P, float* __restrict__ state, const unsigned char* __restrict__ mask, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && mask[e]) Core<L, S>::snapshot_reset(*T, P, P.env_id_base + (unsigned long long)e, state + (size_t)e * S::STATE);
}
template <class S>
__global__ void kw_take_initd(float* __restrict__ state, const float* __restrict__ initd, const unsigned char* __restrict__ mask, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && mask[e]) { const size_t o = (size_t)e * S::STATE + 2 * S::W; state[o + 12] = initd[o + 12]; state[o + 13] = initd[o + 13]; }
}
template <class S, class L>
__global__ void kw_target(const Params P, float* __restrict__ state, const unsigned long long* __restrict__ ids,
                          const unsigned* __restrict__ ep, int cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) Core<L, S>::sample_target(P, ids[i], ep[i], state + (size_t)i * S::STATE);
}
// shape-independent part of an engine + the launches that depend on the lane-group shape
struct WideEngine {
    pbre_config cfg;
    Params P;
    int n = 0, obs_dim = 0, act_dim = 0, ow = 0, device = 0, sf = 0, nj = 0, lc = 0, tgs = 0;   // tgs: floats per env of the motor-target buffer
    bool mrec = false;                        // the shape keeps persistent motor records (iCub with hands)
    unsigned char* d_mask = nullptr;
    float *state = nullptr, *tmp = nullptr, *tgt = nullptr, *tgt_tmp = nullptr;
    float *d_act = nullptr, *d_out = nullptr;
    int* d_bad = nullptr;                     // NaN / Inf guard counter (Params::bad_count)
    float* d_hull = nullptr;                  // PBRE_SHAPE_HULL: the object's vertex / face table (Params::hull; pbre_set_object_hull)
    int* d_sweeps = nullptr;                  // [n] sweeps every env's solver ran in the last step (Params::sweeps; pbre_physics.solver_residual_threshold > 0)
    float* objv = nullptr;                    // [n][W] side records of kw_obj, or nullptr: object rows always solved in kw_step
    const float* obj_done = nullptr;          // state buffer whose object solve rode along with the last kw_ik launch (consumed by the next step)
    unsigned long long* d_ids = nullptr; unsigned* d_ep = nullptr; int* d_idx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    static constexpr int KRING = 64;
    hipEvent_t ev_k[KRING][2] = {};
    long k_steps = 0;
    double ms[3] = {0, 0, 0};
    bool ext_dirty = false;                   // a step was enqueued on a caller-supplied stream since the last wquiesce()
    std::string err;
    enum { K_SETTLE, K_SETTLE_TGT, K_STEP_ACT, K_STEP_TGT, K_INNER_ACT, K_INNER_TGT };
    virtual ~WideEngine() {}
    virtual std::string tables(const pbre_config& c) = 0;
    virtual hipError_t upload_tables() = 0;
    virtual void free_tables() = 0;
    virtual void launch_step(int kind, float* st, float* tg, int cnt, const float* act, float* out, int flags, hipStream_t s) = 0;
    virtual void launch_ik(bool reset, float* st, const float* act, float* tg, int cnt, hipStream_t s, bool step_follows = false) = 0;   // step_follows: the next launch on s is launch_step on st
    virtual void launch_observe(bool initd, float* st, float* out, int cnt, hipStream_t s) = 0;
    virtual void launch_init(float* st, int cnt, hipStream_t s) = 0;
    virtual void launch_snapshot_reset(const unsigned char* mask, hipStream_t s) = 0;
    bool have_snapshot = false, stale_snapshot = false;   // stale: pbre_set_physics changed the scene the snapshot was recorded in
    virtual void launch_target(float* st, int cnt, hipStream_t s) = 0;
    virtual void launch_mrec_init(float* tg, int cnt, hipStream_t s) = 0;
    virtual void launch_set_motors(const MotorCmd& cmd, const unsigned char* mask, hipStream_t s) = 0;
    virtual void launch_cmd_joints(const float* act, float vmax, hipStream_t s) = 0;
    virtual int ndof() const = 0;
    virtual void snapshot(const float* rec) = 0;
    virtual void limits(float* lo, float* hi) const = 0;
    virtual int vgprs() const = 0;
    // lane-per-env path (pbre_lane.hpp; the Shape32 engine of pbre_wide.hip overrides these): steps of the whole batch on `state`
    virtual bool lane_ok() const { return false; }
    virtual hipError_t lane_alloc() { return hipSuccess; }   // after the shape-independent buffers exist
    virtual void lane_invalidate() {}                  // the records were changed by something other than a lane step: classes are stale
    virtual void launch_lane_ik(const float* act, hipStream_t s) {}
    virtual hipError_t launch_lane_step(int kind, const float* act, float* out, int flags, hipStream_t s, bool timed) { return hipSuccess; }
    virtual int lane_info(int* vg, int* complex_now) { return 0; }
};

template <class S, class L>
struct WideImpl : WideEngine {
    using C = Core<L, S>;
    static constexpr int EPB = WTPB / phys_lanes<S>();
    TablesT<S> T;
    TablesT<S>* dT = nullptr;
    static int blocks_of(int cnt) { return (cnt + EPB - 1) / EPB; }
    std::string tables(const pbre_config& c) override {
        std::string e = make_tables<S>(c, T, P);
        if (e.empty()) { obs_dim = obs_dim_of(T, P); sf = S::STATE; nj = S::NJ; lc = S::LC; tgs = S::TGT; mrec = S::MREC; }
        return e;
    }
    hipError_t upload_tables() override {
        if (!dT) { hipError_t e = hipMalloc(&dT, sizeof(TablesT<S>)); if (e != hipSuccess) return e; }
        return hipMemcpy(dT, &T, sizeof(TablesT<S>), hipMemcpyHostToDevice);
    }
    void free_tables() override { if (dT) (void)hipFree(dT); dT = nullptr; }
    template <int MODE>
    void step_t(float* st, float* tg, int cnt, const float* act, float* out, int flags, hipStream_t s) {
        if (P.res_lim > 0.f) {         // Bullet's residual exit: one solve over all rows of an env, no side solve of the object
            obj_done = nullptr;
            hipLaunchKernelGGL((kw_step<S, L, MODE, true>), dim3(blocks_of(cnt)), dim3(WTPB), 0, s, dT, P, st, act, out, cnt, act_dim, ow, flags, tg, (const float*)nullptr);
            return;
        }
        // (a convex-hull object's rows stay in kw_step: the per-env object solver kw_obj is compiled for the primitives)
        const float* ov = (objv && !(flags & 1) && P.obj_shape != PBRE_SHAPE_HULL) ? objv : nullptr;
        const bool done = ov && obj_done == st;
        obj_done = nullptr;
        if (ov && !done) hipLaunchKernelGGL((kw_obj<S>), dim3((cnt + 63) / 64), dim3(64), 0, s, P, st, objv, cnt);
        hipLaunchKernelGGL((kw_step<S, L, MODE>), dim3(blocks_of(cnt)), dim3(WTPB), 0, s, dT, P, st, act, out, cnt, act_dim, ow, flags, tg, ov);
    }
    void launch_step(int kind, float* st, float* tg, int cnt, const float* act, float* out, int flags, hipStream_t s) override {
        constexpr int OT = C::M_OBS | C::M_TASK;
        if (S::MREC && kind == K_SETTLE) kind = K_SETTLE_TGT;      // the motor record is always the source of the targets
        switch (kind) {
            case K_SETTLE: step_t<0>(st, tg, cnt, act, out, flags, s); break;
            case K_SETTLE_TGT: step_t<C::M_TGT>(st, tg, cnt, act, out, flags, s); break;
            case K_STEP_ACT: step_t<C::M_ACTION | OT>(st, tg, cnt, act, out, flags, s); break;
            case K_INNER_ACT: step_t<C::M_ACTION | C::M_TASK | C::M_INNER>(st, tg, cnt, act, out, flags, s); break;
            case K_INNER_TGT: step_t<C::M_TGT | C::M_TASK | C::M_INNER>(st, tg, cnt, act, out, flags, s); break;
            default: step_t<C::M_TGT | OT>(st, tg, cnt, act, out, flags, s); break;
        }
    }
    void launch_ik(bool reset, float* st, const float* act, float* tg, int cnt, hipStream_t s, bool step_follows = false) override {
        const int ikb = blocks_of(cnt);
        if (reset) hipLaunchKernelGGL((kw_ik<S, L, true>), dim3(ikb), dim3(WTPB), 0, s, dT, P, st, act, tg, cnt, act_dim, ikb, (float*)nullptr);
        else {
            const bool ride = step_follows && objv != nullptr && !(cfg.flags & PBRE_F_NO_OBJECT) && !(P.res_lim > 0.f) && P.obj_shape != PBRE_SHAPE_HULL;
            hipLaunchKernelGGL((kw_ik<S, L, false>), dim3(ikb + (ride ? (cnt + WTPB - 1) / WTPB : 0)), dim3(WTPB), 0, s, dT, P, st, act, tg, cnt, act_dim,
                               ikb, ride ? objv : nullptr);
            obj_done = ride ? st : nullptr;
        }
    }
    void launch_observe(bool initd, float* st, float* out, int cnt, hipStream_t s) override {
        if (initd) hipLaunchKernelGGL((kw_observe<S, L, C::M_INITD>), dim3(blocks_of(cnt)), dim3(WTPB), 0, s, dT, P, st, out, cnt, ow);
        else hipLaunchKernelGGL((kw_observe<S, L, C::M_OBS>), dim3(blocks_of(cnt)), dim3(WTPB), 0, s, dT, P, st, out, cnt, ow);
    }
    void launch_snapshot_reset(const unsigned char* mask, hipStream_t s) override {
        hipLaunchKernelGGL((kw_snapshot_reset<S, L>), dim3((n + 127) / 128), dim3(128), 0, s, dT, P, state, mask, n);
        if (P.robot >= 1 && P.task >= 1) {          // icub_push_gym_env.py:124-127: distances of the new episode's first state
            (void)hipMemcpyAsync(tmp, state, (size_t)n * S::STATE * sizeof(float), hipMemcpyDeviceToDevice, s);
            launch_observe(true, tmp, nullptr, n, s);
            hipLaunchKernelGGL((kw_take_initd<S>), dim3((n + 127) / 128), dim3(128), 0, s, state, tmp, mask, n);
        }
    }
    void launch_init(float* st, int cnt, hipStream_t s) override {
        hipLaunchKernelGGL((kw_init<S, L>), dim3((cnt + 127) / 128), dim3(128), 0, s, dT, P, st, d_ids, d_ep, cnt);
    }
    void launch_target(float* st, int cnt, hipStream_t s) override {
        hipLaunchKernelGGL((kw_target<S, L>), dim3((cnt + 127) / 128), dim3(128), 0, s, P, st, d_ids, d_ep, cnt);
    }
    void launch_mrec_init(float* tg, int cnt, hipStream_t s) override {
        if (S::MREC) hipLaunchKernelGGL((kw_mrec_init<S>), dim3((cnt * S::W + 255) / 256), dim3(256), 0, s, dT, tg, cnt);
    }
    void launch_set_motors(const MotorCmd& cmd, const unsigned char* mask, hipStream_t s) override {
        if (S::MREC) hipLaunchKernelGGL((kw_set_motors<S>), dim3((n * 64 + 255) / 256), dim3(256), 0, s, tgt, n, cmd, mask);
    }
    void launch_cmd_joints(const float* act, float vmax, hipStream_t s) override {
        if (S::MREC) hipLaunchKernelGGL((kw_cmd_joints<S>), dim3((n * 64 + 255) / 256), dim3(256), 0, s, dT, tgt, act, n, act_dim, vmax);
    }
    int ndof() const override { return T.ndof; }
    void snapshot(const float* rec) override {
        for (int k = 0; k < S::NJ; k++) { T.rst_q[k] = rec[k]; P.rst_q[k] = rec[k]; }
        P.rst_objz = rec[S::LC + 2];
    }
    void limits(float* lo, float* hi) const override { obs_limits(cfg, T, lo, hi); }
    int vgprs() const override {
        hipFuncAttributes fa;
        constexpr int M = C::M_ACTION | C::M_OBS | C::M_TASK;
        return hipFuncGetAttributes(&fa, (const void*)kw_step<S, L, M>) == hipSuccess ? fa.numRegs : -1;
    }
};


WideEngine* make_hands_engine();       // pbre_hands.hip
WideEngine* make_lane_engine();        // pbre_lane.hip
WideEngine* make_icub_arm_engine();    // pbre_icub_arm.hip

}  // namespace pbre
