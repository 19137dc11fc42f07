// pbre_wide.hip -- the 64-lane engine: robots with up to 32 DoF (iCub) are stepped one env per wavefront by the lane-group
// core (pbre_core.hpp instantiated for Shape64).  Lane k of the wave owns DoF k (32 robot lanes, 6 object lanes, the
// constant lane); M^-1 rows, constraint rows and the 150-iteration PGS state live in VGPRs, cross-lane traffic is
// v_readlane (broadcast of a row's owner), DPP (all-reduce) and ds_bpermute (tree gathers).  No LDS, no barriers; a
// block is 4 independent waves.  State: 144 floats per env (Q[64] | V[64] | X[16]).
//
// Replaces, per env (reference file:line): iCubReachGymEnv / iCubPushGymEnv / iCubPushGymGoalEnv .step and .reset
// (icub_reach_gym_env.py:114-259, icub_push_gym_env.py:116-282, icub_push_gym_goal_env.py:69-139), iCubEnv.apply_action /
// get_observation (icub_env.py:202-361) and the p.stepSimulation / p.calculateInverseKinematics calls inside them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#define PBRE_HD __device__ __forceinline__
#define PBRE_UNROLL _Pragma("unroll")
#include "pbre_host.hpp"
#include "lanes_device.hpp"
#include "pbre_core.hpp"
#include "pbre_wide.hpp"

namespace pbre {

using SW = Shape64;
using CoreW = Core<DevLanes64, SW>;
using TablesW = TablesT<SW>;
constexpr int WST = SW::STATE, WNJ = SW::NJ, WW = SW::W;
constexpr int WPB = 4, WTPB = WPB * 64;          // envs (waves) per block

template <int MODE>
__global__ __launch_bounds__(WTPB, 3) void kw_step(const TablesW* __restrict__ T, const Params P, float* __restrict__ state,
                                                const float* __restrict__ actions, float* __restrict__ out, int n, int act_dim, int ow,
                                                int flags, const float* __restrict__ tgt) {
    const int env = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (env >= n) return;                           // whole wave
    CoreW::step(*T, P, state + (size_t)env * WST, (MODE & CoreW::M_ACTION) ? actions + (size_t)env * act_dim : nullptr,
                (MODE & CoreW::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags, (MODE & CoreW::M_TGT) ? tgt + (size_t)env * WNJ : nullptr,
                P.env_id_base + (unsigned long long)env);
}
template <bool RESET>
__global__ __launch_bounds__(WTPB) void kw_ik(const TablesW* __restrict__ T, const Params P, float* __restrict__ state,
                                              const float* __restrict__ actions, float* __restrict__ tgt, int n, int act_dim) {
    const int env = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (env >= n) return;
    CoreW::ik_targets(*T, P, state + (size_t)env * WST, RESET ? nullptr : actions + (size_t)env * act_dim, tgt + (size_t)env * WNJ, RESET);
}
template <int MODE>
__global__ __launch_bounds__(WTPB) void kw_observe(const TablesW* __restrict__ T, const Params P, float* __restrict__ state,
                                                   float* __restrict__ out, int n, int ow) {
    const int env = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (env >= n) return;
    float* st = state + (size_t)env * WST;
    float Q = DevLanes64::load(st), V = DevLanes64::load(st + WW), X = DevLanes64::loadm(st + 2 * WW, DevLanes64::lane() < 16);
    CoreW::observe(*T, P, st, Q, V, X, (MODE & CoreW::M_OBS) ? out + (size_t)env * ow : nullptr, MODE);
}
__global__ void kw_init(const TablesW* __restrict__ T, const Params P, float* __restrict__ state,
                        const unsigned long long* __restrict__ ids, const unsigned* __restrict__ ep, int cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) CoreW::init_state(*T, P, ids[i], ep[i], state + (size_t)i * WST);
}
__global__ void kw_target(const Params P, float* __restrict__ state, const unsigned long long* __restrict__ ids,
                          const unsigned* __restrict__ ep, int cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) CoreW::sample_target(P, ids[i], ep[i], state + (size_t)i * WST);
}
__global__ void kw_next_episode(const float* __restrict__ state, const int* __restrict__ idx, int cnt, unsigned* __restrict__ ep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) ep[i] = (unsigned)((int)state[(size_t)idx[i] * WST + 2 * WW + 5] + 1);
}
__global__ void kw_scatter(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ idx, int cnt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / WST, k = t % WST;
    if (i < cnt) dst[(size_t)idx[i] * WST + k] = src[(size_t)i * WST + k];
}

struct WideEngine {
    pbre_config cfg;
    TablesW T; Params P;
    int n = 0, obs_dim = 0, act_dim = 0, ow = 0, device = 0;
    TablesW* dT = nullptr;
    float *state = nullptr, *tmp = nullptr, *tgt = nullptr, *tgt_tmp = nullptr;
    float *d_act = nullptr, *d_out = nullptr;
    unsigned long long* d_ids = nullptr; unsigned* d_ep = nullptr; int* d_idx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    static constexpr int KRING = 64;
    hipEvent_t ev_k[KRING][2] = {};
    long k_steps = 0;
    double ms[3] = {0, 0, 0};
    std::string err;
};

#define WCHK(call)                                                                  \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            w->err = std::string(#call) + ": " + hipGetErrorString(e_);             \
            return PBRE_E_DEVICE;                                                   \
        }                                                                           \
    } while (0)

static int blocks_of(int n) { return (n + WPB - 1) / WPB; }

template <int MODE>
static hipError_t wstep(WideEngine* w, float* st, float* tg, int n, const float* act, float* out, int flags, hipStream_t s, bool timed = false) {
    hipEvent_t* ek = w->ev_k[w->k_steps % WideEngine::KRING];
    if (timed) (void)hipEventRecord(ek[0], s);
    hipLaunchKernelGGL(kw_step<MODE>, dim3(blocks_of(n)), dim3(WTPB), 0, s, w->dT, w->P, st, act, out, n, w->act_dim, w->ow, flags, tg);
    if (timed) { (void)hipEventRecord(ek[1], s); w->k_steps++; }
    return hipGetLastError();
}
static hipError_t wsettle(WideEngine* w, float* st, float* tg, int n, int count, int flags, hipStream_t s) {
    for (int i = 0; i < count; i++) {
        hipError_t e = w->P.use_ik ? wstep<CoreW::M_TGT>(w, st, tg, n, nullptr, nullptr, flags, s) : wstep<0>(w, st, tg, n, nullptr, nullptr, flags, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
static hipError_t wfull_step(WideEngine* w, const float* d_act, float* d_out, hipStream_t s) {
    const int flags = w->cfg.flags & (PBRE_F_NO_OBJECT | PBRE_F_AUTO_RESET);
    constexpr int OT = CoreW::M_OBS | CoreW::M_TASK;
    if (!w->P.use_ik) return wstep<CoreW::M_ACTION | OT>(w, w->state, w->tgt, w->n, d_act, d_out, flags, s, true);
    hipLaunchKernelGGL(kw_ik<false>, dim3(blocks_of(w->n)), dim3(WTPB), 0, s, w->dT, w->P, w->state, d_act, w->tgt, w->n, w->act_dim);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return wstep<CoreW::M_TGT | OT>(w, w->state, w->tgt, w->n, nullptr, d_out, flags, s, true);
}

void wide_destroy(WideEngine* w) {
    if (!w) return;
    (void)hipSetDevice(w->device);
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    for (void* p : {(void*)w->dT, (void*)w->state, (void*)w->tmp, (void*)w->tgt, (void*)w->tgt_tmp, (void*)w->d_act, (void*)w->d_out,
                    (void*)w->d_ids, (void*)w->d_ep, (void*)w->d_idx})
        if (p) (void)hipFree(p);
    for (auto& e : w->ev) if (e) (void)hipEventDestroy(e);
    for (auto& pr : w->ev_k) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    delete w;
}

int wide_create(const pbre_config* cfg, WideEngine** out, std::string& err) {
    WideEngine* w = new WideEngine();
    w->cfg = *cfg;
    std::string e = make_tables<SW>(*cfg, w->T, w->P);
    if (!e.empty()) {
        err = e; delete w;
        return e.find("robot_table") == 0 ? PBRE_E_TABLE : (e.find("not implemented") != std::string::npos ? PBRE_E_UNSUPPORTED : PBRE_E_ARG);
    }
    w->cfg.robot_table = nullptr;
    w->n = cfg->num_envs; w->obs_dim = obs_dim_of(w->T, w->P); w->act_dim = act_dim_of(*cfg); w->ow = w->obs_dim + 2; w->device = cfg->device_id;
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) { err = std::string("no HIP device available (") + hipGetErrorString(he) + "); libpbre has no CPU fallback"; delete w; return PBRE_E_DEVICE; }
    if (w->device < 0 || w->device >= ndev) { err = "device_id out of range"; delete w; return PBRE_E_ARG; }
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); wide_destroy(w); return PBRE_E_DEVICE; } } while (0)
    CK(hipSetDevice(w->device));
    CK(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    for (auto& ev : w->ev) CK(hipEventCreate(&ev));
    for (auto& pr : w->ev_k) for (auto& ev : pr) CK(hipEventCreate(&ev));
    const size_t n = (size_t)w->n;
    CK(hipMalloc(&w->dT, sizeof(TablesW)));
    CK(hipMemcpy(w->dT, &w->T, sizeof(TablesW), hipMemcpyHostToDevice));
    CK(hipMalloc(&w->state, n * WST * sizeof(float)));
    CK(hipMalloc(&w->tmp, n * WST * sizeof(float)));
    CK(hipMalloc(&w->tgt, n * WNJ * sizeof(float)));
    CK(hipMalloc(&w->tgt_tmp, n * WNJ * sizeof(float)));
    CK(hipMemset(w->tgt, 0, n * WNJ * sizeof(float)));
    CK(hipMemset(w->tgt_tmp, 0, n * WNJ * sizeof(float)));
    CK(hipMalloc(&w->d_act, n * w->act_dim * sizeof(float)));
    CK(hipMalloc(&w->d_out, n * w->ow * sizeof(float)));
    CK(hipMalloc(&w->d_ids, n * sizeof(unsigned long long)));
    CK(hipMalloc(&w->d_ep, n * sizeof(unsigned)));
    CK(hipMalloc(&w->d_idx, n * sizeof(int)));
    {   // every record holds a valid (un-settled) state with episode -1
        std::vector<unsigned long long> ids(n, w->P.env_id_base); std::vector<unsigned> ep(n, 0xFFFFFFFFu);
        CK(hipMemcpy(w->d_ids, ids.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(w->d_ep, ep.data(), n * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(kw_init, dim3((w->n + 127) / 128), dim3(128), 0, w->stream, w->dT, w->P, w->state, w->d_ids, w->d_ep, w->n);
        hipLaunchKernelGGL(kw_init, dim3((w->n + 127) / 128), dim3(128), 0, w->stream, w->dT, w->P, w->tmp, w->d_ids, w->d_ep, w->n);
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
    if (sf) *sf = WST;
}
int wide_sync(WideEngine* w) {
    WCHK(hipSetDevice(w->device));
    WCHK(hipStreamSynchronize(w->stream));
    return PBRE_OK;
}
int wide_observe(WideEngine* w, float* obs) {
    WCHK(hipSetDevice(w->device));
    hipLaunchKernelGGL(kw_observe<CoreW::M_OBS>, dim3(blocks_of(w->n)), dim3(WTPB), 0, w->stream, w->dT, w->P, w->state, w->d_out, w->n, w->ow);
    WCHK(hipGetLastError());
    WCHK(hipMemcpy2DAsync(obs, (size_t)w->obs_dim * 4, w->d_out, (size_t)w->ow * 4, (size_t)w->obs_dim * 4, w->n, hipMemcpyDeviceToHost, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    return PBRE_OK;
}
int wide_settle(WideEngine* w, int32_t n, int32_t flags) {
    WCHK(hipSetDevice(w->device));
    WCHK(wsettle(w, w->state, w->tgt, w->n, n, flags & PBRE_F_NO_OBJECT, w->stream));
    WCHK(hipStreamSynchronize(w->stream));
    return PBRE_OK;
}
int wide_reset(WideEngine* w, const uint8_t* mask, float* obs) {
    WCHK(hipSetDevice(w->device));
    std::vector<int> idx;
    for (int e = 0; e < w->n; e++) if (!mask || mask[e]) idx.push_back(e);
    const int cnt = (int)idx.size();
    if (cnt > 0) {
        std::vector<unsigned long long> ids(cnt);
        for (int i = 0; i < cnt; i++) ids[i] = w->P.env_id_base + (unsigned long long)idx[i];
        hipStream_t s = w->stream;
        WCHK(hipMemcpyAsync(w->d_ids, ids.data(), (size_t)cnt * 8, hipMemcpyHostToDevice, s));
        WCHK(hipMemcpyAsync(w->d_idx, idx.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kw_next_episode, dim3((cnt + 127) / 128), dim3(128), 0, s, w->state, w->d_idx, cnt, w->d_ep);
        WCHK(hipGetLastError());
        WCHK(hipStreamSynchronize(s));                      // host vectors go out of scope below
        const bool full = cnt == w->n;
        float* st = full ? w->state : w->tmp;               // a partial reset settles a compacted copy
        float* tg = full ? w->tgt : w->tgt_tmp;
        const int f0 = w->cfg.flags & PBRE_F_NO_OBJECT;
        hipLaunchKernelGGL(kw_init, dim3((cnt + 127) / 128), dim3(128), 0, s, w->dT, w->P, st, w->d_ids, w->d_ep, cnt);
        WCHK(hipGetLastError());
        // iCubEnv.reset (icub_env.py:88-151): joints at their initial positions, IK targets of the home hand pose when
        // use_IK, one stepSimulation; then reset_simulation (icub_reach_gym_env.py:135-148): 100 steps robot alone,
        // world loaded, 100 + 1 steps
        if (w->P.use_ik) {
            hipLaunchKernelGGL(kw_ik<true>, dim3(blocks_of(cnt)), dim3(WTPB), 0, s, w->dT, w->P, st, (const float*)nullptr, tg, cnt, w->act_dim);
            WCHK(hipGetLastError());
        }
        WCHK(wsettle(w, st, tg, cnt, (w->P.use_ik || w->P.robot == PBRE_ROBOT_ICUB ? 1 : 0) + 100, PBRE_F_NO_OBJECT, s));
        WCHK(wsettle(w, st, tg, cnt, 101, f0, s));
        hipLaunchKernelGGL(kw_target, dim3((cnt + 127) / 128), dim3(128), 0, s, w->P, st, w->d_ids, w->d_ep, cnt);
        WCHK(hipGetLastError());
        if (w->P.task >= 1) {   // iCubPushGymEnv.reset (icub_push_gym_env.py:124-127): distances the normalised reward divides by
            hipLaunchKernelGGL(kw_observe<CoreW::M_INITD>, dim3(blocks_of(cnt)), dim3(WTPB), 0, s, w->dT, w->P, st, (float*)nullptr, cnt, w->ow);
            WCHK(hipGetLastError());
        }
        if (!full) {
            hipLaunchKernelGGL(kw_scatter, dim3((cnt * WST + 255) / 256), dim3(256), 0, s, w->state, st, w->d_idx, cnt);
            WCHK(hipGetLastError());
            // the IK targets of the reset envs are only needed while settling; the next step recomputes them
        }
        WCHK(hipStreamSynchronize(s));
        if (full) {   // snapshot for PBRE_F_AUTO_RESET: settled robot pose and object height (identical in every env)
            std::vector<float> rec(WST);
            WCHK(hipMemcpy(rec.data(), w->state, WST * sizeof(float), hipMemcpyDeviceToHost));
            for (int k = 0; k < WNJ; k++) { w->T.rst_q[k] = rec[k]; w->P.rst_q[k] = rec[k]; }
            w->P.rst_objz = rec[SW::LC + 2];
            WCHK(hipMemcpy(w->dT, &w->T, sizeof(TablesW), hipMemcpyHostToDevice));
        }
    }
    if (obs) return wide_observe(w, obs);
    return PBRE_OK;
}
int wide_step_device(WideEngine* w, const float* d_actions, float* d_out, void* stream) {
    WCHK(hipSetDevice(w->device));
    WCHK(wfull_step(w, d_actions, d_out, stream ? (hipStream_t)stream : w->stream));
    return PBRE_OK;
}
int wide_step(WideEngine* w, const float* actions, float* out) {
    WCHK(hipSetDevice(w->device));
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
    WCHK(hipStreamSynchronize(w->stream));
    WCHK(hipMemcpy(s, w->state, (size_t)w->n * WST * 4, hipMemcpyDeviceToHost));
    return PBRE_OK;
}
int wide_set_state(WideEngine* w, const float* s) {
    WCHK(hipSetDevice(w->device));
    WCHK(hipStreamSynchronize(w->stream));
    WCHK(hipMemcpy(w->state, s, (size_t)w->n * WST * 4, hipMemcpyHostToDevice));
    return PBRE_OK;
}
int wide_get_physics(const WideEngine* w, pbre_physics* p) { *p = w->cfg.phys; return PBRE_OK; }
int wide_set_physics(WideEngine* w, const pbre_physics* p) {
    Params P2 = w->P;
    if (!apply_physics(*p, P2)) { w->err = "bad physics parameters"; return PBRE_E_ARG; }
    WCHK(hipSetDevice(w->device));
    WCHK(hipStreamSynchronize(w->stream));
    w->cfg.phys = *p; w->P = P2;
    return PBRE_OK;
}
int wide_obs_limits(const WideEngine* w, float* lo, float* hi) { obs_limits(w->cfg, w->T, lo, hi); return PBRE_OK; }
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
    hipFuncAttributes fa;
    int rg = -1;
    constexpr int M = CoreW::M_ACTION | CoreW::M_OBS | CoreW::M_TASK;
    if (hipFuncGetAttributes(&fa, (const void*)kw_step<M>) == hipSuccess) rg = fa.numRegs;
    const int v[7] = {-1, rg, 0, 0, w->n, 0, -1};      // same slots as the Panda engine: [1] VGPRs of the lane-group kernel, [4] envs it steps
    for (int i = 0; i < n; i++) info[i] = i < 7 ? v[i] : 0;
    return PBRE_OK;
}

}  // namespace pbre
