// pbre_capi.hip -- libpbre.so: HIP kernels (gfx950) + the C-ABI of include/pbre.h.
//
// Kernels.  k_step<MODE> instantiates the lane-generic step (pbre_core.hpp) with the
// device lane backend: block = 256 threads = 4 waves = 16 envs, one env per 16-lane DPP
// row, no LDS, no scratch.  State lives in HBM as one 192-byte record per env (three
// 64-byte lane records), so a wave's loads/stores are 256 contiguous bytes per record.
// Grid = ceil(num_envs/16) blocks; the block index -> env mapping is linear, so the 8 XCDs
// (block b runs on XCD b%8) each stream disjoint 3 KB slices -- there is no inter-block
// reuse to be XCD-aware about, and every block is independent (no barriers, no atomics).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#define PBRE_HD __device__ __forceinline__
#define PBRE_UNROLL _Pragma("unroll")
#define PBRE_ANY(x) (__any((int)(x)) != 0)
#define PBRE_REG_BARRIER() asm volatile("" ::: "memory")
#define PBRE_LAUNDER(p) asm volatile("" : "+s"(p))
#include "pbre_host.hpp"
#include "lanes_device.hpp"
#include "pbre_core.hpp"
#include "pbre_fast.hpp"

using namespace pbre;
using CoreD = Core<DevLanes>;
using FastD = Fast<TopoPanda>;

constexpr int EPB = 16;              // envs per block
constexpr int TPB = EPB * W;         // 256 threads

// ------------------------------------------------------------------ kernels
// General row kernel.  MODE: CoreD::M_* bits.  n = real env count; state has ceil16(n) + 16 records (the last 16 are
// valid dummy records for the padding rows of a partially filled block).  With list == nullptr the kernel steps envs
// [0, n); otherwise it steps the *count envs named in list (those the fast path declined); the grid is sized for the
// worst case and surplus blocks exit immediately.  actions/out rows of padding rows are redirected to env 0 / a scratch row.
template <int MODE>
__global__ __launch_bounds__(TPB) void k_step(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                              const float* __restrict__ actions, float* __restrict__ out,
                                              float* __restrict__ scratch_row, int n, int dummy_base, int act_dim, int ow, int flags,
                                              const int* __restrict__ list, const int* __restrict__ count) {
    const int row = threadIdx.x >> 4;
    const int total = list ? *count : n;
    const int i = blockIdx.x * EPB + row;
    if ((int)blockIdx.x * EPB >= total) return;          // block-uniform: blocks beyond the list exit at once
    const bool real = i < total;
    const int env = real ? (list ? list[i] : i) : dummy_base + row;
    float* st = state + (size_t)env * STATE;
    const float* a = nullptr;
    float* o = nullptr;
    if (MODE & CoreD::M_ACTION) a = actions + (size_t)(real ? env : 0) * act_dim;
    if (MODE & CoreD::M_OBS) o = real ? out + (size_t)env * ow : scratch_row;
    CoreD::step(*T, P, st, a, o, MODE, flags);
}

// Lane-per-env fast path: one thread = one env.  Envs it cannot handle (robot contact or limit row) are appended to
// `list` for the general kernel.
constexpr int FTPB = 64;
#ifndef PBRE_FAST_WAVES
#define PBRE_FAST_WAVES 2      // waves per SIMD the fast kernel is register-limited to (tuned on MI355X)
#endif
template <int MODE>
__global__ __launch_bounds__(FTPB, PBRE_FAST_WAVES) void k_fast(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                               const float* __restrict__ actions, float* __restrict__ out,
                                               int n, int act_dim, int ow, int flags, int* __restrict__ list, int* __restrict__ count) {
    const int env = blockIdx.x * FTPB + threadIdx.x;
    if (env >= n) return;
    const bool ok = FastD::step(*T, P, state + (size_t)env * STATE, (MODE & FastD::M_ACTION) ? actions + (size_t)env * act_dim : nullptr,
                                (MODE & FastD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags);
    if (!ok) list[atomicAdd(count, 1)] = env;
}

__global__ __launch_bounds__(TPB) void k_observe(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                                 float* __restrict__ out, float* __restrict__ scratch_row, int n, int ow) {
    const int env = blockIdx.x * EPB + (threadIdx.x >> 4);
    float* st = state + (size_t)env * STATE;
    float Q = DevLanes::load(st), V = DevLanes::load(st + 16), X = DevLanes::load(st + 32);
    CoreD::observe(*T, P, st, Q, V, X, env < n ? out + (size_t)env * ow : scratch_row, CoreD::M_OBS);
}

// robot.reset + WorldEnv._sample_pose: one thread per env.  ids: global env id per record, ep: episode per record
__global__ void k_init(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                       const unsigned long long* __restrict__ ids, const unsigned* __restrict__ ep, int cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) CoreD::init_state(*T, P, ids[i], ep[i], state + (size_t)i * STATE);
}
__global__ void k_target(const Params P, float* __restrict__ state, const unsigned long long* __restrict__ ids,
                         const unsigned* __restrict__ ep, int cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) CoreD::sample_target(P, ids[i], ep[i], state + (size_t)i * STATE);
}
// dst[idx[i]] <- src[i] (scatter) or dst[i] <- src[idx[i]] (gather), 48 floats per record
__global__ void k_move(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ idx, int cnt, int scatter) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / STATE, k = t % STATE;
    if (i >= cnt) return;
    if (scatter) dst[(size_t)idx[i] * STATE + k] = src[(size_t)i * STATE + k];
    else dst[(size_t)i * STATE + k] = src[(size_t)idx[i] * STATE + k];
}

// ------------------------------------------------------------------ context
struct pbre_ctx {
    pbre_config cfg;
    Tables T; Params P;
    int n = 0, npad = 0, obs_dim = 0, act_dim = 0, ow = 0, device = 0;
    Tables* dT = nullptr;
    float *d_state = nullptr, *d_act = nullptr, *d_out = nullptr, *d_scratch = nullptr, *d_tmp = nullptr;
    unsigned long long* d_ids = nullptr; unsigned* d_ep = nullptr; int* d_idx = nullptr;
    int *d_list = nullptr, *d_count = nullptr;     // envs declined by the fast path in the current step
    bool fast_ok = false;
    int grid_general = 0;
    std::vector<unsigned> episode;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double ms[3] = {0, 0, 0};
    std::string err;
};
static std::string g_err;

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            c->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
            return PBRE_E_DEVICE;                                                                      \
        }                                                                                              \
    } while (0)

static int ceil16(int n) { return (n + EPB - 1) / EPB * EPB; }

template <int MODE>
static hipError_t launch_step(pbre_ctx* c, float* state, int n, const float* act, float* out, int flags, hipStream_t s) {
    const int dummy = ceil16(n);          // first of the 16 dummy records behind this buffer's real records
    if (c->fast_ok && !(c->cfg.flags & PBRE_F_FORCE_GENERAL)) {
        hipError_t e = hipMemsetAsync(c->d_count, 0, sizeof(int), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_fast<MODE>, dim3((n + FTPB - 1) / FTPB), dim3(FTPB), 0, s, c->dT, c->P, state, act, out, n,
                           c->act_dim, c->ow, flags, c->d_list, c->d_count);
        const int blocks = (n + EPB - 1) / EPB;
        hipLaunchKernelGGL(k_step<MODE>, dim3(blocks), dim3(TPB), 0, s, c->dT, c->P, state, act, out, c->d_scratch, n, dummy,
                           c->act_dim, c->ow, flags, c->d_list, c->d_count);
    } else {
        hipLaunchKernelGGL(k_step<MODE>, dim3(ceil16(n) / EPB), dim3(TPB), 0, s, c->dT, c->P, state, act, out, c->d_scratch, n, dummy,
                           c->act_dim, c->ow, flags, (const int*)nullptr, (const int*)nullptr);
    }
    return hipGetLastError();
}

extern "C" {

int pbre_default_config(pbre_config* cfg, int32_t robot, int32_t task) { return default_config(cfg, robot, task); }

void pbre_destroy(pbre_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : {(void*)c->dT, (void*)c->d_state, (void*)c->d_act, (void*)c->d_out, (void*)c->d_scratch,
                    (void*)c->d_tmp, (void*)c->d_ids, (void*)c->d_ep, (void*)c->d_idx, (void*)c->d_list, (void*)c->d_count})
        if (p) (void)hipFree(p);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int pbre_create(const pbre_config* cfg, pbre_ctx** out) {
    if (!cfg || !out) { g_err = "null argument"; return PBRE_E_ARG; }
    *out = nullptr;
    pbre_ctx* c = new pbre_ctx();
    c->cfg = *cfg;
    std::string e = make_tables(*cfg, c->T, c->P);
    if (!e.empty()) {
        g_err = e; delete c;
        return e.find("robot_table") == 0 ? PBRE_E_TABLE : (e.find("not implemented") != std::string::npos ? PBRE_E_UNSUPPORTED : PBRE_E_ARG);
    }
    c->cfg.robot_table = nullptr;
    c->n = cfg->num_envs; c->npad = ceil16(c->n); c->obs_dim = obs_dim_of(c->T, c->P); c->act_dim = cfg->num_controlled_joints;
    c->ow = c->obs_dim + 2; c->device = cfg->device_id;
    c->episode.assign(c->n, 0u);
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) {
        g_err = std::string("no HIP device available (") + hipGetErrorString(he) + "); libpbre has no CPU fallback";
        delete c; return PBRE_E_DEVICE;
    }
    if (c->device < 0 || c->device >= ndev) { g_err = "device_id out of range"; delete c; return PBRE_E_ARG; }
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_err = std::string(#call) + ": " + hipGetErrorString(e_); pbre_destroy(c); return PBRE_E_DEVICE; } } while (0)
    CK(hipSetDevice(c->device));
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& ev : c->ev) CK(hipEventCreate(&ev));
    CK(hipMalloc(&c->dT, sizeof(Tables)));
    CK(hipMemcpy(c->dT, &c->T, sizeof(Tables), hipMemcpyHostToDevice));
    const size_t sb = (size_t)(c->npad + EPB) * STATE * sizeof(float);     // + 16 dummy records for padding rows
    CK(hipMalloc(&c->d_state, sb)); CK(hipMemset(c->d_state, 0, sb));
    CK(hipMalloc(&c->d_tmp, sb));
    CK(hipMalloc(&c->d_act, (size_t)c->npad * c->act_dim * sizeof(float)));
    CK(hipMalloc(&c->d_out, (size_t)c->npad * c->ow * sizeof(float)));
    CK(hipMalloc(&c->d_scratch, 64 * sizeof(float)));
    CK(hipMalloc(&c->d_ids, (size_t)c->npad * sizeof(unsigned long long)));
    CK(hipMalloc(&c->d_ep, (size_t)c->npad * sizeof(unsigned)));
    CK(hipMalloc(&c->d_idx, (size_t)c->npad * sizeof(int)));
    CK(hipMalloc(&c->d_list, (size_t)c->npad * sizeof(int)));
    CK(hipMalloc(&c->d_count, sizeof(int)));
    CK(hipMemset(c->d_count, 0, sizeof(int)));
    c->fast_ok = topo_matches<TopoPanda>(c->T) && fast_scene_ok(c->P);
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, c->device)); c->grid_general = pr.multiProcessorCount * 2; }
#undef CK
    // padding and dummy records must hold a valid state: initialise every record of both buffers to the un-settled reset pose
    {
        const int tot = c->npad + EPB;
        std::vector<unsigned long long> ids(tot, c->P.env_id_base); std::vector<unsigned> ep(tot, 0u);
        unsigned long long* d_i = nullptr; unsigned* d_e = nullptr;
        bool ok = hipMalloc(&d_i, tot * 8) == hipSuccess && hipMalloc(&d_e, tot * 4) == hipSuccess &&
                  hipMemcpy(d_i, ids.data(), tot * 8, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(d_e, ep.data(), tot * 4, hipMemcpyHostToDevice) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_init, dim3((tot + 127) / 128), dim3(128), 0, c->stream, c->dT, c->P, c->d_state, d_i, d_e, tot);
            hipLaunchKernelGGL(k_init, dim3((tot + 127) / 128), dim3(128), 0, c->stream, c->dT, c->P, c->d_tmp, d_i, d_e, tot);
            ok = hipStreamSynchronize(c->stream) == hipSuccess;
        }
        if (d_i) (void)hipFree(d_i);
        if (d_e) (void)hipFree(d_e);
        if (!ok) { g_err = "initialising the state records failed"; pbre_destroy(c); return PBRE_E_DEVICE; }
    }
    *out = c;
    return PBRE_OK;
}

const char* pbre_last_error(const pbre_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

int pbre_dims(const pbre_ctx* c, int32_t* od, int32_t* ad, int32_t* n) {
    if (!c) return PBRE_E_ARG;
    if (od) *od = c->obs_dim;
    if (ad) *ad = c->act_dim;
    if (n) *n = c->n;
    return PBRE_OK;
}

int pbre_sync(pbre_ctx* c) {
    if (!c) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRE_OK;
}

int pbre_observe(pbre_ctx* c, float* obs) {
    if (!c || !obs) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_observe, dim3(c->npad / EPB), dim3(TPB), 0, c->stream, c->dT, c->P, c->d_state, c->d_out, c->d_scratch, c->n, c->ow);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy2DAsync(obs, (size_t)c->obs_dim * 4, c->d_out, (size_t)c->ow * 4, (size_t)c->obs_dim * 4, c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRE_OK;
}

int pbre_settle(pbre_ctx* c, int32_t n, int32_t flags) {
    if (!c || n < 0) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i < n; i++) HIPCHK(launch_step<0>(c, c->d_state, c->n, nullptr, nullptr, flags & PBRE_F_NO_OBJECT, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRE_OK;
}

int pbre_reset(pbre_ctx* c, const uint8_t* mask, float* obs) {
    if (!c) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    std::vector<int> idx;
    for (int e = 0; e < c->n; e++) if (!mask || mask[e]) idx.push_back(e);
    const int cnt = (int)idx.size();
    if (cnt > 0) {
        const int cpad = ceil16(cnt);
        std::vector<unsigned long long> ids(cpad); std::vector<unsigned> ep(cpad);
        for (int i = 0; i < cpad; i++) {
            const int e = idx[i < cnt ? i : cnt - 1];
            ids[i] = c->P.env_id_base + (unsigned long long)e;
            ep[i] = c->episode[e];
        }
        for (int i = 0; i < cnt; i++) c->episode[idx[i]]++;
        HIPCHK(hipMemcpyAsync(c->d_ids, ids.data(), (size_t)cpad * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_ep, ep.data(), (size_t)cpad * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_idx, idx.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));          // host vectors go out of scope below
        float* work = (cnt == c->n) ? c->d_state : c->d_tmp;   // a partial reset settles a compacted copy
        hipLaunchKernelGGL(k_init, dim3((cpad + 127) / 128), dim3(128), 0, c->stream, c->dT, c->P, work, c->d_ids, c->d_ep, cpad);
        HIPCHK(hipGetLastError());
        // reset_simulation (panda_push_gym_env.py:117-148): 100 steps robot alone, then world loaded: 100 + 1 steps
        for (int i = 0; i < 100; i++) HIPCHK(launch_step<0>(c, work, cnt, nullptr, nullptr, PBRE_F_NO_OBJECT, c->stream));
        for (int i = 0; i < 101; i++) HIPCHK(launch_step<0>(c, work, cnt, nullptr, nullptr, c->cfg.flags & PBRE_F_NO_OBJECT, c->stream));
        hipLaunchKernelGGL(k_target, dim3((cpad + 127) / 128), dim3(128), 0, c->stream, c->P, work, c->d_ids, c->d_ep, cpad);
        HIPCHK(hipGetLastError());
        if (work != c->d_state) {
            hipLaunchKernelGGL(k_move, dim3((cnt * STATE + 255) / 256), dim3(256), 0, c->stream, c->d_state, work, c->d_idx, cnt, 1);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (obs) return pbre_observe(c, obs);
    return PBRE_OK;
}

int pbre_step_device(pbre_ctx* c, const float* d_actions, float* d_out, void* stream) {
    if (!c || !d_actions || !d_out) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK((launch_step<CoreD::M_ACTION | CoreD::M_OBS | CoreD::M_TASK>(c, c->d_state, c->n, d_actions, d_out, c->cfg.flags & PBRE_F_NO_OBJECT, s)));
    return PBRE_OK;
}

int pbre_step(pbre_ctx* c, const float* actions, float* out) {
    if (!c || !actions || !out) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    HIPCHK(hipMemcpyAsync(c->d_act, actions, (size_t)c->n * c->act_dim * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK((launch_step<CoreD::M_ACTION | CoreD::M_OBS | CoreD::M_TASK>(c, c->d_state, c->n, c->d_act, c->d_out, c->cfg.flags & PBRE_F_NO_OBJECT, c->stream)));
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    HIPCHK(hipMemcpyAsync(out, c->d_out, (size_t)c->n * c->ow * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) { float t = 0; HIPCHK(hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1])); c->ms[i] = t; }
    return PBRE_OK;
}

int pbre_get_state(pbre_ctx* c, float* s) {
    if (!c || !s) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(s, c->d_state, (size_t)c->n * STATE * 4, hipMemcpyDeviceToHost));
    return PBRE_OK;
}
int pbre_set_state(pbre_ctx* c, const float* s) {
    if (!c || !s) return PBRE_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(c->d_state, s, (size_t)c->n * STATE * 4, hipMemcpyHostToDevice));
    return PBRE_OK;
}

int pbre_obs_limits(const pbre_ctx* c, float* lo, float* hi) {
    if (!c || !lo || !hi) return PBRE_E_ARG;
    obs_limits(c->cfg, c->T, lo, hi);
    return PBRE_OK;
}
int pbre_timing(const pbre_ctx* c, double* ms, int32_t n) {
    if (!c || !ms) return PBRE_E_ARG;
    for (int i = 0; i < n; i++) ms[i] = i < 3 ? c->ms[i] : 0.0;
    return PBRE_OK;
}
int pbre_kernel_info(const pbre_ctx* c, int32_t* info, int32_t n) {
    if (!c || !info) return PBRE_E_ARG;
    hipFuncAttributes fa;
    int rf = -1, rg = -1;
    if (hipFuncGetAttributes(&fa, (const void*)k_fast<FastD::M_ACTION | FastD::M_OBS | FastD::M_TASK>) == hipSuccess) rf = fa.numRegs;
    if (hipFuncGetAttributes(&fa, (const void*)k_step<CoreD::M_ACTION | CoreD::M_OBS | CoreD::M_TASK>) == hipSuccess) rg = fa.numRegs;
    int last = 0;       // envs the fast path declined in the most recent step (reads the device counter)
    if (c->d_count) { (void)hipSetDevice(c->device); (void)hipMemcpy(&last, c->d_count, sizeof(int), hipMemcpyDeviceToHost); }
    const int v[5] = {c->fast_ok ? rf : -1, rg, c->fast_ok ? 1 : 0, c->n - last, last};
    for (int i = 0; i < n; i++) info[i] = i < 5 ? v[i] : 0;
    return PBRE_OK;
}

}  // extern "C"
