// pbre_wide.hip -- the wide lane-group engine: robots with more than 9 DoF (iCub) are stepped by the lane-group core
// (pbre_core.hpp) with one env per half-wave (Shape32: <= 20 DoF, 2 envs per wavefront -- the iCub as the engine simulates
// it, i.e. without the legs, model/table.py prune_base_branches) or one env per wavefront (Shape64: <= 32 DoF).  Lane k of
// the group owns DoF k (robot lanes, 6 object lanes, the constant lane); M^-1 rows, constraint rows and the 150-iteration
// PGS state live in VGPRs; cross-lane traffic is DPP / ds_swizzle (all-reduce), ds_bpermute (gathers, broadcasts inside a
// half-wave) and v_readlane (broadcasts inside a whole wave).  No LDS memory, no barriers; a block is 4 independent waves.
// State: Q[W] | V[W] | X[16] floats per env (80 for Shape32, 144 for Shape64).  The kernels and the shape-specific engine half
// are templates in pbre_wide_impl.hpp; the iCub with hands (Shape128, 60 DoF) is instantiated in pbre_hands.hip.
//
// Replaces, per env (reference file:line): iCubReachGymEnv / iCubPushGymEnv / iCubPushGymGoalEnv .step and .reset
// (icub_reach_gym_env.py:114-259, icub_push_gym_env.py:116-282, icub_push_gym_goal_env.py:69-139), iCubEnv.apply_action /
// get_observation (icub_env.py:202-361) and the p.stepSimulation / p.calculateInverseKinematics calls inside them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "pbre_wide_impl.hpp"

namespace pbre {

__global__ void kw_next_episode(const float* __restrict__ state, const int* __restrict__ idx, int cnt, unsigned* __restrict__ ep, int sf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) ep[i] = (unsigned)((int)state[(size_t)idx[i] * sf + (sf - 16) + 5] + 1);
}
__global__ void kw_scatter(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ idx, int cnt, int sf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / sf, k = t % sf;
    if (i < cnt) dst[(size_t)idx[i] * sf + k] = src[(size_t)i * sf + k];
}

#define WCHK(call)                                                                  \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            w->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
            return PBRE_E_DEVICE;                                                   \
        }                                                                           \
    } while (0)

// all work of the engine complete on return (see quiesce() in pbre_capi.hip)
static hipError_t wquiesce(WideEngine* w) {
    if (w->ext_dirty) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
        w->ext_dirty = false;
        return hipSuccess;
    }
    return hipStreamSynchronize(w->stream);
}
static hipError_t wstep(WideEngine* w, int kind, float* st, float* tg, int cnt, const float* act, float* out, int flags, hipStream_t s, bool timed = false) {
    hipEvent_t* ek = w->ev_k[w->k_steps % WideEngine::KRING];
    if (timed) (void)hipEventRecord(ek[0], s);
    w->launch_step(kind, st, tg, cnt, act, out, flags, s);
    if (timed) { (void)hipEventRecord(ek[1], s); w->k_steps++; }
    return hipGetLastError();
}
static hipError_t wsettle(WideEngine* w, float* st, float* tg, int cnt, int count, int flags, hipStream_t s) {
    const int kind = (w->P.use_ik || w->mrec) ? WideEngine::K_SETTLE_TGT : WideEngine::K_SETTLE;
    // settle steps of the whole batch in place: through the lane-per-env pipeline where it is the step path (pbre_lane.hip); the object's
    // presence decides the classes, so they are recomputed for this run of steps and left invalid after it
    const bool lane = w->lane_ok() && st == w->state && tg == w->tgt && cnt == w->n;
    if (lane) w->lane_invalidate();
    for (int i = 0; i < count; i++) {
        hipError_t e = lane ? w->launch_lane_step(kind, nullptr, nullptr, flags, s, false) : wstep(w, kind, st, tg, cnt, nullptr, nullptr, flags, s);
        if (e != hipSuccess) return e;
    }
    if (lane) w->lane_invalidate();
    return hipSuccess;
}
static hipError_t wfull_step(WideEngine* w, const float* d_act, float* d_out, hipStream_t s) {
    const int flags = w->cfg.flags & (PBRE_F_NO_OBJECT | PBRE_F_AUTO_RESET);
    const int reps = w->cfg.action_repeat > 1 ? w->cfg.action_repeat : 1;
    const Params P0 = w->P;
    hipError_t e = hipSuccess;
    for (int r = 0; r < reps && e == hipSuccess; r++) {
        // apply_action loop (icub_reach_gym_env.py:200-246): the reference scales the action in place in every iteration, so
        // iteration r applies action * scale^(r+1); all but the last iteration only simulate, test termination and count
        w->P.act_scale = (r ? w->P.act_scale : 1.f) * P0.act_scale; w->P.ik_ps = (r ? w->P.ik_ps : 1.f) * P0.ik_ps; w->P.ik_rs = (r ? w->P.ik_rs : 1.f) * P0.ik_rs;
        const bool last = r + 1 == reps;
        if (w->lane_ok()) {     // lane-per-env path (pbre_lane.hpp)
            if (w->P.use_ik) w->launch_lane_ik(d_act, s);
            e = w->launch_lane_step(w->P.use_ik ? (last ? WideEngine::K_STEP_TGT : WideEngine::K_INNER_TGT) : (last ? WideEngine::K_STEP_ACT : WideEngine::K_INNER_ACT),
                                    d_act, last ? d_out : nullptr, flags, s, last);
        } else
        if (!w->P.use_ik) e = wstep(w, last ? WideEngine::K_STEP_ACT : WideEngine::K_INNER_ACT, w->state, w->tgt, w->n, d_act, last ? d_out : nullptr, flags, s, last);
        else {
            w->launch_ik(false, w->state, d_act, w->tgt, w->n, s, true);
            if ((e = hipGetLastError()) != hipSuccess) { w->obj_done = nullptr; break; }
            e = wstep(w, last ? WideEngine::K_STEP_TGT : WideEngine::K_INNER_TGT, w->state, w->tgt, w->n, nullptr, last ? d_out : nullptr, flags, s, last);
        }
    }
    w->P = P0;
    return e;
}

void wide_destroy(WideEngine* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    w->free_tables();
    for (void* p : {(void*)w->state, (void*)w->tmp, (void*)w->tgt, (void*)w->tgt_tmp, (void*)w->d_act, (void*)w->d_out,
                    (void*)w->d_ids, (void*)w->d_ep, (void*)w->d_idx, (void*)w->d_mask, (void*)w->objv, (void*)w->d_bad, (void*)w->d_sweeps, (void*)w->d_hull})
        if (p) (void)hipFree(p);
    for (auto& e : w->ev) if (e) (void)hipEventDestroy(e);
    for (auto& pr : w->ev_k) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
}

int wide_create(const pbre_config* cfg, WideEngine** out, std::string& err) {
    const int nd = table_ndof(*cfg);
    WideEngine* w = nd > Shape64::NJ ? make_hands_engine()
                  : (cfg->robot_level && nd <= ShapePA::NJ ? static_cast<WideEngine*>(new WideImpl<ShapePA, DevLanes32>())     // pandaEnv alone
                  : (cfg->robot_level && nd <= ShapeIA::NJ ? make_icub_arm_engine()                                             // iCubEnv alone
                  : (nd <= Shape32::NJ ? make_lane_engine()
                                       : static_cast<WideEngine*>(new WideImpl<Shape64, DevLanes64>()))));
    w->cfg = *cfg;
    std::string e = w->tables(*cfg);
    if (!e.empty()) {
        err = e; delete w;
        return e.find("robot_table") == 0 ? PBRE_E_TABLE : (e.find("not implemented") != std::string::npos ? PBRE_E_UNSUPPORTED : PBRE_E_ARG);
    }
    w->cfg.robot_table = nullptr;
    w->n = cfg->num_envs; w->act_dim = act_dim_of(*cfg); w->ow = w->obs_dim + 2; w->device = cfg->device_id;
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) { err = std::string("no HIP device available (") + hipGetErrorString(he) + "); libpbre has no CPU fallback"; delete w; return PBRE_E_DEVICE; }
    if (w->device < 0 || w->device >= ndev) { err = "device_id out of range"; delete w; return PBRE_E_ARG; }
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); wide_destroy(w); return PBRE_E_DEVICE; } } while (0)
    CK(hipSetDevice(w->device));
    CK(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    for (auto& ev : w->ev) CK(hipEventCreate(&ev));
    // (timing-only events around the dominant kernel: no system-scope fence at the markers)
    for (auto& pr : w->ev_k) for (auto& ev : pr) CK(hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
    const size_t n = (size_t)w->n, sf = (size_t)w->sf, nj = (size_t)w->tgs;
    CK(w->upload_tables());
    CK(hipMalloc(&w->state, n * sf * sizeof(float)));
    CK(hipMalloc(&w->tmp, n * sf * sizeof(float)));
    CK(hipMalloc(&w->tgt, n * nj * sizeof(float)));
    CK(hipMalloc(&w->tgt_tmp, n * nj * sizeof(float)));
    CK(hipMemset(w->tgt, 0, n * nj * sizeof(float)));
    CK(hipMemset(w->tgt_tmp, 0, n * nj * sizeof(float)));
    {   // side records of the per-env object solve (pbre_objstep.hpp); PBRE_OBJ_SPLIT=0 keeps every object row in kw_step (A/B runs)
        const char* knob = getenv("PBRE_OBJ_SPLIT");
        if (!(knob && knob[0] == '0')) {
            const size_t wl = (sf - 16) / 2;
            CK(hipMalloc(&w->objv, n * wl * sizeof(float)));
            CK(hipMemset(w->objv, 0, n * wl * sizeof(float)));
        }
    }
    CK(hipMalloc(&w->d_act, n * w->act_dim * sizeof(float)));
    CK(hipMalloc(&w->d_out, n * w->ow * sizeof(float)));
    CK(hipMalloc(&w->d_bad, 2 * sizeof(int)));
    CK(hipMemset(w->d_bad, 0, 2 * sizeof(int)));
    w->P.bad_count = w->d_bad;
    CK(hipMalloc(&w->d_sweeps, (size_t)w->n * sizeof(int)));
    CK(hipMemset(w->d_sweeps, 0, (size_t)w->n * sizeof(int)));
    w->P.sweeps = w->d_sweeps;
    CK(w->lane_alloc());
    CK(hipMalloc(&w->d_ids, n * sizeof(unsigned long long)));
    CK(hipMalloc(&w->d_ep, n * sizeof(unsigned)));
    CK(hipMalloc(&w->d_idx, n * sizeof(int)));
    {   // every record holds a valid (un-settled) state with episode -1
        std::vector<unsigned long long> ids(n, w->P.env_id_base); std::vector<unsigned> ep(n, 0xFFFFFFFFu);
        CK(hipMemcpy(w->d_ids, ids.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(w->d_ep, ep.data(), n * 4, hipMemcpyHostToDevice));
        w->launch_init(w->state, w->n, w->stream);
        w->launch_init(w->tmp, w->n, w->stream);
        w->launch_mrec_init(w->tgt, w->n, w->stream);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(w->stream));
    }
#undef CK
    *out = w;
    return PBRE_OK;
}

const char* wide_error(const WideEngine* w) { return w->err.c_str(); }
void wide_dims(const WideEngine* w, int32_t* od, int32_t* ad, int32_t* n, int32_t* sf) {
    if (od) *od = w->obs_dim;
    if (ad) *ad = w->act_dim;
    if (n) *n = w->n;
    if (sf) *sf = w->sf;
}
int wide_sync(WideEngine* w) {
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    return PBRE_OK;
}
int wide_observe(WideEngine* w, float* obs) {
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    w->launch_observe(false, w->state, w->d_out, w->n, w->stream);
    WCHK(hipGetLastError());
    WCHK(hipMemcpy2DAsync(obs, (size_t)w->obs_dim * 4, w->d_out, (size_t)w->ow * 4, (size_t)w->obs_dim * 4, w->n, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    return PBRE_OK;
}
int wide_settle(WideEngine* w, int32_t n, int32_t flags) {
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    WCHK(wsettle(w, w->state, w->tgt, w->n, n, flags & PBRE_F_NO_OBJECT, w->stream));
    w->lane_invalidate();
    WCHK(hipStreamSynchronize(w->stream));
    return PBRE_OK;
}
int wide_reset(WideEngine* w, const uint8_t* mask, float* obs) {
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    w->lane_invalidate();
    std::vector<int> idx;
    for (int e = 0; e < w->n; e++) if (!mask || mask[e]) idx.push_back(e);
    const int cnt = (int)idx.size();
    if (cnt > 0) {
        std::vector<unsigned long long> ids(cnt);
        for (int i = 0; i < cnt; i++) ids[i] = w->P.env_id_base + (unsigned long long)idx[i];
        hipStream_t s = w->stream;
        WCHK(hipMemcpyAsync(w->d_ids, ids.data(), (size_t)cnt * 8, hipMemcpyHostToDevice, s));
        WCHK(hipMemcpyAsync(w->d_idx, idx.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kw_next_episode, dim3((cnt + 127) / 128), dim3(128), 0, s, w->state, w->d_idx, cnt, w->d_ep, w->sf);
        WCHK(hipGetLastError());
        WCHK(hipStreamSynchronize(s));                      // host vectors go out of scope below
        const bool full = cnt == w->n;
        float* st = full ? w->state : w->tmp;               // a partial reset settles a compacted copy
        float* tg = full ? w->tgt : w->tgt_tmp;
        const int f0 = w->cfg.flags & PBRE_F_NO_OBJECT;
        w->launch_init(st, cnt, s);
        w->launch_mrec_init(tg, cnt, s);
        WCHK(hipGetLastError());
        // iCubEnv.reset (icub_env.py:88-151): joints at their initial positions, IK targets of the home hand pose when
        // use_IK, one stepSimulation; then reset_simulation (icub_reach_gym_env.py:135-148): 100 steps robot alone,
        // world loaded, 100 + 1 steps
        if (w->P.use_ik) {
            w->launch_ik(true, st, nullptr, tg, cnt, s);
            WCHK(hipGetLastError());
        }
        WCHK(wsettle(w, st, tg, cnt, (w->P.use_ik || w->P.robot != PBRE_ROBOT_PANDA ? 1 : 0) + 100, PBRE_F_NO_OBJECT, s));
        WCHK(wsettle(w, st, tg, cnt, 101, f0, s));
        w->launch_target(st, cnt, s);
        WCHK(hipGetLastError());
        if (w->P.task >= 1) {   // iCubPushGymEnv.reset (icub_push_gym_env.py:124-127): distances the normalised reward divides by
            w->launch_observe(true, st, nullptr, cnt, s);
            WCHK(hipGetLastError());
        }
        if (!full) {
            // the IK targets of the reset envs are only needed while settling; the next step recomputes them
            hipLaunchKernelGGL(kw_scatter, dim3((cnt * w->sf + 255) / 256), dim3(256), 0, s, w->state, st, w->d_idx, cnt, w->sf);
            if (w->mrec) hipLaunchKernelGGL(kw_scatter, dim3((cnt * w->tgs + 255) / 256), dim3(256), 0, s, w->tgt, tg, w->d_idx, cnt, w->tgs);   // motors persist
            WCHK(hipGetLastError());
        }
        WCHK(hipStreamSynchronize(s));
        if (full) {   // snapshot for PBRE_F_AUTO_RESET: settled robot pose and object height (identical in every env)
            std::vector<float> rec(w->sf);
            WCHK(hipMemcpy(rec.data(), w->state, (size_t)w->sf * sizeof(float), hipMemcpyDeviceToHost));
            w->snapshot(rec.data());
            WCHK(w->upload_tables());
            w->have_snapshot = true; w->stale_snapshot = false;
            // end-effector pose of the settled robot (the first 6 observation entries of env 0) for the lane-per-env pipeline's in-kernel
            // restart (Lane::finish): valid while the settled state is in the simple class (no robot sphere at the object)
            w->P.rst_ok = 0;
            if (w->lane_ok() && !w->mrec) {
                w->launch_observe(false, w->state, w->d_out, w->n, s);
                WCHK(hipGetLastError());
                WCHK(hipStreamSynchronize(s));
                float row[6]; int vg = 0, cn = 1;
                WCHK(hipMemcpy(row, w->d_out, sizeof row, hipMemcpyDeviceToHost));
                for (int k = 0; k < 6; k++) w->P.rst_ee[k] = row[k];
                if (w->lane_info(&vg, &cn) && cn == 0) w->P.rst_ok = 1;
            }
        }
    }
    if (obs) return wide_observe(w, obs);
    return PBRE_OK;
}
int wide_reset_snapshot(WideEngine* w, const uint8_t* mask, float* obs) {
    if (w->mrec) { w->err = "pbre_reset_snapshot: task envs only (the robot-level interfaces have no episodes)"; return PBRE_E_UNSUPPORTED; }
    if (!w->have_snapshot) { w->err = w->stale_snapshot ? stale_snapshot_msg() : "pbre_reset_snapshot: no settled snapshot yet (call pbre_reset for the whole batch first)"; return PBRE_E_ARG; }
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    if (!w->d_mask) WCHK(hipMalloc(&w->d_mask, (size_t)w->n));
    WCHK(hipMemcpyAsync(w->d_mask, mask, (size_t)w->n, hipMemcpyHostToDevice, w->stream));
    w->lane_invalidate();
    w->launch_snapshot_reset(w->d_mask, w->stream);
    WCHK(hipGetLastError());
    WCHK(hipStreamSynchronize(w->stream));
    if (obs) return wide_observe(w, obs);
    return PBRE_OK;
}
int wide_step_device(WideEngine* w, const float* d_actions, float* d_out, void* stream) {
    WCHK(hipSetDevice(w->device));
    if (stream) w->ext_dirty = true;
    if (w->stale_snapshot && (w->cfg.flags & PBRE_F_AUTO_RESET)) { w->err = stale_snapshot_msg(); return PBRE_E_ARG; }
    WCHK(wfull_step(w, d_actions, d_out, stream == PBRE_STREAM_LEGACY ? (hipStream_t) nullptr : (stream ? (hipStream_t)stream : w->stream)));
    return PBRE_OK;
}
int wide_step(WideEngine* w, const float* actions, float* out) {
    WCHK(hipSetDevice(w->device));
    if (w->ext_dirty) WCHK(wquiesce(w));
    if (w->stale_snapshot && (w->cfg.flags & PBRE_F_AUTO_RESET)) { w->err = stale_snapshot_msg(); return PBRE_E_ARG; }
    hipStream_t s = w->stream;
    WCHK(hipEventRecord(w->ev[0], s));
    WCHK(hipMemcpyAsync(w->d_act, actions, (size_t)w->n * w->act_dim * 4, hipMemcpyHostToDevice, s));
    WCHK(hipEventRecord(w->ev[1], s));
    WCHK(wfull_step(w, w->d_act, w->d_out, s));
    WCHK(hipEventRecord(w->ev[2], s));
    WCHK(hipMemcpyAsync(out, w->d_out, (size_t)w->n * w->ow * 4, hipMemcpyDeviceToHost, s));
    WCHK(hipEventRecord(w->ev[3], s));
    WCHK(hipStreamSynchronize(s));
    for (int i = 0; i < 3; i++) { float t = 0; WCHK(hipEventElapsedTime(&t, w->ev[i], w->ev[i + 1])); w->ms[i] = t; }
    return PBRE_OK;
}
int wide_get_state(WideEngine* w, float* s) {
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    WCHK(hipMemcpy(s, w->state, (size_t)w->n * w->sf * 4, hipMemcpyDeviceToHost));
    return PBRE_OK;
}
int wide_get_state_cols(WideEngine* w, int32_t first, int32_t count, float* out) {
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    WCHK(hipMemcpy2D(out, (size_t)count * 4, w->state + first, (size_t)w->sf * 4, (size_t)count * 4, w->n, hipMemcpyDeviceToHost));
    return PBRE_OK;
}
int wide_set_state(WideEngine* w, const float* s) {
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    WCHK(hipMemcpy(w->state, s, (size_t)w->n * w->sf * 4, hipMemcpyHostToDevice));
    w->lane_invalidate();
    return PBRE_OK;
}
int wide_set_motors(WideEngine* w, int32_t cnt, const int32_t* dofs, const float* targets, double kp, double max_force, double max_vel, const uint8_t* mask) {
    if (!w->mrec) { w->err = "pbre_set_motors: only the robot-level engines keep a motor record"; return PBRE_E_UNSUPPORTED; }
    if (cnt > 64) { w->err = "pbre_set_motors: more than 64 joints"; return PBRE_E_ARG; }
    MotorCmd cmd;
    cmd.n = cnt; cmd.kp = (float)kp;
    cmd.fscale = max_force > 0 ? (float)(max_force * w->cfg.phys.dt / w->cfg.phys.max_motor_impulse) : 1.f;
    cmd.vmax = max_vel > 0 ? (float)max_vel : 0.f;
    for (int k = 0; k < cnt; k++) {
        if (dofs[k] < 0 || dofs[k] >= w->ndof()) { w->err = "pbre_set_motors: bad DoF index"; return PBRE_E_ARG; }
        cmd.dof[k] = dofs[k]; cmd.target[k] = targets[k];
    }
    if (cnt == 0) return PBRE_OK;
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    if (mask) {
        if (!w->d_mask) WCHK(hipMalloc(&w->d_mask, (size_t)w->n));
        WCHK(hipMemcpyAsync(w->d_mask, mask, (size_t)w->n, hipMemcpyHostToDevice, w->stream));
    }
    w->launch_set_motors(cmd, mask ? w->d_mask : nullptr, w->stream);
    WCHK(hipGetLastError());
    WCHK(hipStreamSynchronize(w->stream));
    return PBRE_OK;
}
int wide_apply_action(WideEngine* w, const float* actions, double max_vel) {
    if (!w->mrec) { w->err = "pbre_apply_action: only the robot-level engines keep a motor record"; return PBRE_E_UNSUPPORTED; }
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    hipStream_t s = w->stream;
    WCHK(hipMemcpyAsync(w->d_act, actions, (size_t)w->n * w->act_dim * 4, hipMemcpyHostToDevice, s));
    const bool panda = w->P.robot == PBRE_ROBOT_PANDA;
    const float vm = max_vel > 0 ? (float)max_vel : 0.f;
    if (w->P.use_ik) {
        // with max_vel the iCub commands every joint (positionGain 0.2, icub_env.py:338-346), the Panda its 7 arm joints with
        // PyBullet's default positionGain 0.1 [EXT-UNVERIFIED] (panda_env.py:284-290)
        const Params P0 = w->P;
        w->P.cmd_vmax = vm;
        if (panda && vm > 0.f) { w->P.cmd_kp = 0.1f; w->P.cmd_nj = 7; }
        w->launch_ik(false, w->state, w->d_act, w->tgt, w->n, s);
        w->P = P0;
    } else w->launch_cmd_joints(w->d_act, panda ? 0.f : vm, s);      // the joint branch passes maxVelocity on the iCub only (icub_env.py:353-360)
    WCHK(hipGetLastError());
    WCHK(hipStreamSynchronize(s));
    return PBRE_OK;
}
int wide_motor_state(WideEngine* w, float* out, const float* in) {
    if (!w->mrec) { w->err = "pbre_get/set_motor_state: only the iCub-with-hands engine keeps a motor record"; return PBRE_E_UNSUPPORTED; }
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    if (out) WCHK(hipMemcpy(out, w->tgt, (size_t)w->n * w->tgs * 4, hipMemcpyDeviceToHost));
    if (in) WCHK(hipMemcpy(w->tgt, in, (size_t)w->n * w->tgs * 4, hipMemcpyHostToDevice));
    return PBRE_OK;
}
int wide_get_physics(const WideEngine* w, pbre_physics* p) { *p = w->cfg.phys; return PBRE_OK; }
int wide_set_physics(WideEngine* w, const pbre_physics* p) {
    Params P2 = w->P;
    if (!apply_physics(*p, P2)) { w->err = "bad physics parameters"; return PBRE_E_ARG; }
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    if (snapshot_relevant_change(w->cfg.phys, *p)) { w->stale_snapshot = w->stale_snapshot || w->have_snapshot; w->have_snapshot = false; P2.rst_ok = 0; }
    w->cfg.phys = *p; w->P = P2;
    w->lane_invalidate();          // the contact margin may have changed
    return PBRE_OK;
}
int wide_set_object_hull(WideEngine* w, const double* verts, int32_t n_verts) {
    HullTable H;
    const std::string e = build_hull(verts, n_verts, H);
    if (!e.empty()) { w->err = e; return PBRE_E_ARG; }
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    if (!w->d_hull) WCHK(hipMalloc(&w->d_hull, sizeof H.data));
    WCHK(hipMemcpy(w->d_hull, H.data, sizeof H.data, hipMemcpyHostToDevice));
    w->P.hull = w->d_hull; w->P.hull_nv = H.nv; w->P.hull_nf = H.nf; w->P.hull_rb = H.rb; w->P.obj_shape = PBRE_SHAPE_HULL;
    w->cfg.phys.obj_shape = PBRE_SHAPE_HULL;
    for (int k = 0; k < 3; k++) { w->cfg.phys.obj_h[k] = H.half[k]; w->P.obj_h[k] = (float)H.half[k]; }
    w->P.rst_objz = (float)(w->cfg.h_table + H.half[2]);
    w->stale_snapshot = w->stale_snapshot || w->have_snapshot; w->have_snapshot = false; w->P.rst_ok = 0;
    w->lane_invalidate();
    return PBRE_OK;
}
int wide_get_sweeps(WideEngine* w, int32_t* sweeps) {
    if (!(w->P.res_lim > 0.f)) { w->err = "pbre_get_sweeps: pbre_physics.solver_residual_threshold is 0 (every env runs all solver_iters sweeps)"; return PBRE_E_UNSUPPORTED; }
    WCHK(hipSetDevice(w->device));
    WCHK(wquiesce(w));
    WCHK(hipMemcpy(sweeps, w->d_sweeps, (size_t)w->n * sizeof(int), hipMemcpyDeviceToHost));
    return PBRE_OK;
}
int wide_obs_limits(const WideEngine* w, float* lo, float* hi) { w->limits(lo, hi); return PBRE_OK; }
int wide_timing(const WideEngine* w, double* ms, int32_t n) {
    double kd = 0.0;
    if (n > 3 && w->k_steps > 0) {
        (void)hipSetDevice(w->device);
        (void)hipDeviceSynchronize();
        const long cnt = std::min<long>(w->k_steps, WideEngine::KRING);
        int ok = 0;
        for (long i = 0; i < cnt; i++) {
            float t = 0.f;
            hipEvent_t* ek = const_cast<WideEngine*>(w)->ev_k[(w->k_steps - 1 - i) % WideEngine::KRING];
            if (hipEventElapsedTime(&t, ek[0], ek[1]) == hipSuccess) { kd += t; ok++; }
        }
        kd = ok ? kd / ok : 0.0;
    }
    for (int i = 0; i < n; i++) ms[i] = i < 3 ? w->ms[i] : (i == 3 ? kd : 0.0);
    return PBRE_OK;
}
int wide_kernel_info(const WideEngine* w, int32_t* info, int32_t n) {
    int lv = -1, cn = 0;
    const bool lane = w->lane_ok() && const_cast<WideEngine*>(w)->lane_info(&lv, &cn);
    // same slots as the Panda engine: [0] VGPRs of the lane-per-env kernel, [1] of the lane-group kernel, [2] lane-per-env path in use,
    // [3] envs in the simple class, [4] envs the lane-group kernel steps when the lane path is off, [5] complex envs
    int bad = 0;                      // [12] env-steps that met a non-finite state (NaN / Inf guard)
    if (w->d_bad) { (void)hipSetDevice(w->device); (void)hipDeviceSynchronize(); (void)hipMemcpy(&bad, w->d_bad, sizeof(int), hipMemcpyDeviceToHost); }
    const int v[13] = {lane ? lv : -1, w->vgprs(), lane ? 1 : 0, lane ? w->n - cn : 0, lane ? 0 : w->n, lane ? cn : 0, -1, 0, 0, 0, 0, 0, bad};
    for (int i = 0; i < n; i++) info[i] = i < 13 ? v[i] : 0;
    return PBRE_OK;
}

}  // namespace pbre
