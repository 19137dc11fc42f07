// pbre_capi.hip -- libpbre.so: HIP kernels (gfx950) + the C-ABI of include/pbre.h.
//
// Four stepping kernels:
//   k_fast      lane-per-env (one thread = one env, 64 envs per wave), everything in VGPRs.  Steps the "simple" envs (class 0: no robot
//               contact, no joint-limit row): 9 motor rows (in closed form) + <= 4 object-table contacts.  Two builds: 256 VGPRs / 2 waves
//               per SIMD, and 168 VGPRs / 3 per SIMD for the steps in which the complex envs' waves need room (launch_step picks).
//   k_row_list  complex envs (class 1), few of them: the 16-lane row physics of k_step over the compacted list + Fast::finish,
//               concurrently with k_fast on a second stream.
//   k_fast_rc   complex envs, many of them: lane-per-env with dense robot-contact rows and limit rows, whole register file (1 wave/SIMD).
//   k_step      general 16-lane-row kernel (pbre_core.hpp): one env per DPP row, 4 envs per wave.  Any robot the
//               RobotTable describes (<= 9 DoF); used when the table does not match the compiled-in Panda topology
//               or when PBRE_F_FORCE_GENERAL is set (validation).
// Every step kernel ends by classifying the state it produced and appends complex envs to the list the next step's complex-env
// kernel consumes, so there is no classification pre-pass on the hot path.
//
// State lives in HBM as one 192-byte record per env (three 64-byte lane records).  No kernel uses LDS or barriers; blocks
// are independent, so the block -> XCD mapping is irrelevant (there is no inter-block reuse to be XCD-aware about).
#include "pbre_panda.hpp"

#ifdef PBRE_UNITY
PBRE_STEP_INST_ALL()
#else
PBRE_STEP_INST_ALL(extern)      // pbre_step_inst.hip, one translation unit per (MODE, RT)
#endif

// use_IK = 1: hand-pose update + inverse kinematics -> joint targets (one thread per env).  RESET: targets of the home hand pose.
template <bool RESET>
__global__ __launch_bounds__(FTPB) void k_ik(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                             const float* __restrict__ actions, float* __restrict__ tgt, int n, int act_dim) {
    const int env = blockIdx.x * FTPB + threadIdx.x;
    if (env >= n) return;
    FastD::ik_targets(*T, P, state + (size_t)env * STATE, RESET ? nullptr : actions + (size_t)env * act_dim, tgt + (size_t)env * NJ, RESET);
}

// Class of every env's current state (after reset / set_state / a change of the NO_OBJECT flag).
__global__ __launch_bounds__(FTPB) void k_classify(const Tables* __restrict__ T, const Params P, const float* __restrict__ state, int n, int flags,
                                                   signed char* __restrict__ cls, int* __restrict__ list, int* __restrict__ count, int cap) {
    const int env = blockIdx.x * FTPB + threadIdx.x;
    if (env >= n) return;
    publish_class(env, FastD::classify_state(*T, P, state + (size_t)env * STATE, flags), cls, list, count, cap);
}

__global__ void k_total(const int* __restrict__ count, int* __restrict__ host_total, int* __restrict__ recent) {
    int t = 0;
    for (int b = 0; b < NB; b++) t += count[b];
    *recent = 0;
    const int acc = recent[1];
    report_hint(t, recent, host_total);
    recent[1] = acc;                    // a (re)classification is not a step
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
// pbre_reset_snapshot: the selected envs start their next episode from the settled snapshot (what PBRE_F_AUTO_RESET does in-kernel)
__global__ void k_snapshot_reset(const Tables* __restrict__ T, const Params P, float* __restrict__ state, const unsigned char* __restrict__ mask, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n && mask[e]) CoreD::snapshot_reset(*T, P, P.env_id_base + (unsigned long long)e, state + (size_t)e * STATE);
}
// episode number of the next reset of env idx[i]: one more than the episode stored in its record (-1 = never reset)
// (also hands the env's per-env object parameters X[12], X[13], X[15] -- which a reset keeps -- to the record it is re-initialised in)
__global__ void k_next_episode(const float* __restrict__ state, const int* __restrict__ idx, int cnt, int cpad, unsigned* __restrict__ ep,
                               float* __restrict__ work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cpad) return;
    const float* src = state + (size_t)idx[i < cnt ? i : cnt - 1] * STATE;
    ep[i] = (unsigned)((int)src[37] + 1);
    if (work != state) { float* dst = work + (size_t)i * STATE; dst[44] = src[44]; dst[45] = src[45]; dst[47] = src[47]; dst[31] = src[31]; }
}
// dst[idx[i]] <- src[i], 48 floats per record
__global__ void k_scatter(float* __restrict__ dst, const float* __restrict__ src, const int* __restrict__ idx, int cnt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t / STATE, k = t % STATE;
    if (i < cnt) dst[(size_t)idx[i] * STATE + k] = src[(size_t)i * STATE + k];
}

static thread_local std::string g_err;      // errors without a ctx (pbre_create): per calling thread (MultiEngine creates its shards from one thread per device)

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            c->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
            return PBRE_E_DEVICE;                                                                      \
        }                                                                                              \
    } while (0)

// Every host-synchronous entry point starts here: all work the ctx has in flight is complete on return.  Steps enqueued on a
// caller-supplied stream (pbre_step_device) are not ordered against the ctx's own non-blocking streams, so after one of those the
// whole device is drained (these entry points are not on the hot path).
static hipError_t quiesce(pbre_ctx* c) {
    hipError_t e;
    if (c->ext_dirty) {
        if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
        c->ext_dirty = false;
        return hipSuccess;
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return e;
    return hipStreamSynchronize(c->side);      // (also the pipelined host path's downloads: pbre_step_wait then finds their events complete)
}

static hipError_t alloc_buf(EnvBuf& b, int cap) {
    b.cap = cap;
    hipError_t e;
    // cap records + EPB pristine dummy records (read by the idle rows of the row kernels) + EPB scratch records (written by them)
    if ((e = hipMalloc(&b.state, (size_t)(cap + 2 * EPB) * STATE * sizeof(float))) != hipSuccess) return e;
    if ((e = hipMemset(b.state, 0, (size_t)(cap + 2 * EPB) * STATE * sizeof(float))) != hipSuccess) return e;     // k_init keeps X[12], X[13], X[15]
    if ((e = hipMalloc(&b.cls, (size_t)2 * cap)) != hipSuccess) return e;
    if ((e = hipMalloc(&b.tgt, (size_t)(cap + EPB) * NJ * sizeof(float))) != hipSuccess) return e;
    if ((e = hipMemset(b.tgt, 0, (size_t)(cap + EPB) * NJ * sizeof(float))) != hipSuccess) return e;
    if ((e = hipMemset(b.cls, 0, (size_t)2 * cap)) != hipSuccess) return e;
    for (int k = 0; k < 2; k++) if ((e = hipMalloc(&b.list[k], (size_t)NB * cap * sizeof(int))) != hipSuccess) return e;
    if ((e = hipMalloc(&b.count, (3 * NB + 2) * sizeof(int))) != hipSuccess) return e;      // + the device copy of the "recent" hint
    if ((e = hipMalloc(&b.objv_g, (size_t)(cap + 32) * W * sizeof(float))) != hipSuccess) return e;
    if ((e = hipMemset(b.objv_g, 0, (size_t)(cap + 32) * W * sizeof(float))) != hipSuccess) return e;
    if ((e = hipMalloc(&b.pair_g, (size_t)((cap + FTPB - 1) / FTPB) * PBRE_PAIR_G_BYTES)) != hipSuccess) return e;
    if ((e = hipMemset(b.pair_g, 0, (size_t)((cap + FTPB - 1) / FTPB) * PBRE_PAIR_G_BYTES)) != hipSuccess) return e;
    if ((e = hipHostMalloc(&b.h_total, 2 * sizeof(int), hipHostMallocDefault)) != hipSuccess) return e;
    b.h_total[0] = 1; b.h_total[1] = 16;   // unknown until the first step has run
    return hipMemset(b.count, 0, (3 * NB + 2) * sizeof(int));
}
static void free_buf(EnvBuf& b) {
    for (void* p : {(void*)b.state, (void*)b.cls, (void*)b.tgt, (void*)b.list[0], (void*)b.list[1], (void*)b.count, (void*)b.objv_g, (void*)b.pair_g}) if (p) (void)hipFree(p);
    if (b.h_total) (void)hipHostFree(b.h_total);
}

// (re)build class array and current list of the first n envs of b
static hipError_t classify(pbre_ctx* c, EnvBuf& b, int n, int flags, hipStream_t s) {
    if (!lane_per_env(c)) return hipSuccess;
    hipError_t e = hipMemsetAsync(b.count + b.ccur * NB, 0, NB * sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_classify, dim3((n + FTPB - 1) / FTPB), dim3(FTPB), 0, s, c->dT, c->P, b.state, n, flags, b.cls + (size_t)b.cur * b.cap, b.list[b.cur], b.count + b.ccur * NB, b.cap);
    hipLaunchKernelGGL(k_total, dim3(1), dim3(1), 0, s, b.count + b.ccur * NB, b.h_total, b.count + 3 * NB);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return le;
    // The complex-env count k_total just wrote is what the next launch_step sizes its complex-env kernel by.  Every caller is a
    // host-synchronous entry point off the hot path (reset, set_state, settle, set_physics), so wait for it: launched against the count
    // of an EARLIER state, a batch that has just become entirely complex (IK control: the home hand pose's IK solution lies beyond
    // joint 4's limit) was walked by an 8-block row kernel -- 11 ms per launch at 16384 envs, 19 s per reset at 131072.
    return hipStreamSynchronize(s);
}

template <int MODE>
static hipError_t launch_step(pbre_ctx* c, EnvBuf& b, int n, const float* act, float* out, int flags, hipStream_t s) {
    return c->P.res_lim > 0.f ? launch_step_t<MODE, true>(c, b, n, act, out, flags, s) : launch_step_t<MODE, false>(c, b, n, act, out, flags, s);
}

// settle steps (hold motors; IK mode: hold the IK targets)
static hipError_t settle_steps(pbre_ctx* c, EnvBuf& b, int n, int count, int flags, hipStream_t s) {
    for (int i = 0; i < count; i++) {
        hipError_t e = c->P.use_ik ? launch_step<MODE_SETTLE_IK>(c, b, n, nullptr, nullptr, flags, s) : launch_step<0>(c, b, n, nullptr, nullptr, flags, s);
        if (e != hipSuccess) return e;
        // (a settle loop is host-synchronous anyway: every 16 launches let the device catch up, so that the complex-env count the next
        // launches are sized and scheduled by is at most 16 steps old)
        if ((i & 15) == 15 && (e = hipStreamSynchronize(s)) != hipSuccess) return e;
    }
    return hipSuccess;
}
static hipError_t full_step(pbre_ctx* c, const float* d_act, float* d_out, hipStream_t s) {
    const int flags = c->cfg.flags & (PBRE_F_NO_OBJECT | PBRE_F_AUTO_RESET);
    const int reps = c->cfg.action_repeat > 1 ? c->cfg.action_repeat : 1;
    const Params P0 = c->P;
    hipError_t e = hipSuccess;
    for (int r = 0; r < reps && e == hipSuccess; r++) {
        // apply_action loop (panda_push_gym_env.py:193-242): the reference scales the action in place in every iteration, so
        // iteration r applies action * scale^(r+1); all but the last iteration only simulate, test termination and count
        c->P.act_scale = (r ? c->P.act_scale : 1.f) * P0.act_scale; c->P.ik_ps = (r ? c->P.ik_ps : 1.f) * P0.ik_ps; c->P.ik_rs = (r ? c->P.ik_rs : 1.f) * P0.ik_rs;
        const bool last = r + 1 == reps;
        if (!c->P.use_ik) {
            e = last ? launch_step<MODE_STEP>(c, c->main, c->n, d_act, d_out, flags, s) : launch_step<MODE_INNER>(c, c->main, c->n, d_act, nullptr, flags, s);
        } else {
            hipLaunchKernelGGL(k_ik<false>, dim3((c->n + FTPB - 1) / FTPB), dim3(FTPB), 0, s, c->dT, c->P, c->main.state, d_act, c->main.tgt, c->n, c->act_dim);
            if ((e = hipGetLastError()) != hipSuccess) break;
            e = last ? launch_step<MODE_STEP_IK>(c, c->main, c->n, nullptr, d_out, flags, s) : launch_step<MODE_INNER_IK>(c, c->main, c->n, nullptr, nullptr, flags, s);
        }
    }
    c->P = P0;
    return e;
}

extern "C" {

int pbre_default_config(pbre_config* cfg, int32_t robot, int32_t task) { return default_config(cfg, robot, task); }

__attribute__((visibility("hidden"))) void pbre_comm_release(const pbre_ctx* c);      // pbre_comm.hip: the ctx's RCCL communicator, if any

void pbre_destroy(pbre_ctx* c) {
    if (!c) return;
    pbre_comm_release(c);
    if (c->wide) { wide_destroy(c->wide); delete c; return; }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    free_buf(c->main); free_buf(c->tmp);
    for (void* p : {(void*)c->dT, (void*)c->d_act, (void*)c->d_out, (void*)c->d_scratch, (void*)c->d_ids, (void*)c->d_ep, (void*)c->d_idx, (void*)c->d_mask, (void*)c->d_bad, (void*)c->d_sweeps, (void*)c->d_hull})
        if (p) (void)hipFree(p);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->ap.ready) {
        for (int b = 0; b < 2; b++) {
            if (c->ap.d_act[b]) (void)hipFree(c->ap.d_act[b]);
            if (c->ap.d_rows[b]) (void)hipFree(c->ap.d_rows[b]);
            for (hipEvent_t e : {c->ap.ev_step[b], c->ap.ev_out[b]}) if (e) (void)hipEventDestroy(e);
        }
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_join_sys) (void)hipEventDestroy(c->ev_join_sys);
    for (auto& pr : c->ev_k) for (auto& e : pr) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->sp.destroy(); c->side = nullptr;
    delete c;
}

int pbre_create(const pbre_config* cfg, pbre_ctx** out) {
    if (!cfg || !out) { g_err = "null argument"; return PBRE_E_ARG; }
    *out = nullptr;
    pbre_ctx* c = new pbre_ctx();
    if (table_ndof(*cfg) > NJ || cfg->robot_level) {      // iCub shapes, and the robot-level interface (motor records) of either robot
        const int rc = wide_create(cfg, &c->wide, g_err);
        if (rc != PBRE_OK) { delete c; return rc; }
        *out = c;
        return PBRE_OK;
    }
    c->cfg = *cfg;
    std::string e = make_tables<Shape16>(*cfg, c->T, c->P);
    if (!e.empty()) {
        g_err = e; delete c;
        return e.find("robot_table") == 0 ? PBRE_E_TABLE : (e.find("not implemented") != std::string::npos ? PBRE_E_UNSUPPORTED : PBRE_E_ARG);
    }
    c->cfg.robot_table = nullptr;
    c->n = cfg->num_envs; c->npad = ceil16(c->n); c->obs_dim = obs_dim_of(c->T, c->P); c->act_dim = act_dim_of(*cfg);
    c->ow = c->obs_dim + 2; c->device = cfg->device_id;
    c->fast_ok = topo_matches<TopoPanda>(c->T) && fast_scene_ok(c->P);
    if (const char* ev = getenv("PBRE_RC_FIRST_MIN")) c->rc_first_min = atoi(ev);       // A/B knobs
    if (const char* ev = getenv("PBRE_IDLE_SINGLE")) c->idle_single = atoi(ev);
    if (const char* ev = getenv("PBRE_KSAMPLE")) c->ksample = std::max(1, atoi(ev));
    if (const char* ev = getenv("PBRE_IDLE_TOUCH")) c->idle_touch = atoi(ev);
    if (const char* ev = getenv("PBRE_ROW_MAX")) c->row_max = atoi(ev);
    if (const char* ev = getenv("PBRE_FAST3")) c->fast3 = atoi(ev);
    if (const char* ev = getenv("PBRE_FUSED")) c->fused = atoi(ev);
    if (const char* ev = getenv("PBRE_OBJV_SEQ0")) c->main.objv_seq = c->tmp.objv_seq = atoi(ev);      // (tests: start k_fused's sequence numbers next to their wrap)
    if (const char* ev = getenv("PBRE_PAIR")) c->pair = atoi(ev);
    if (const char* ev = getenv("PBRE_TAIL_PAIR")) c->tail_pair = atoi(ev);      // (0: A/B; n > 1: tests -- always the last n chunks)
    if (const char* ev = getenv("PBRE_ZERO_COPY")) c->zero_copy = atoi(ev);
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) {
        g_err = std::string("no HIP device available (") + hipGetErrorString(he) + "); libpbre has no CPU fallback";
        delete c; return PBRE_E_DEVICE;
    }
    if (c->device < 0 || c->device >= ndev) { g_err = "device_id out of range"; delete c; return PBRE_E_ARG; }
#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_err = std::string(#call) + ": " + hipGetErrorString(e_); pbre_destroy(c); return PBRE_E_DEVICE; } } while (0)
    CK(hipSetDevice(c->device));
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, c->device)); c->n_simd = std::max(1, pr.multiProcessorCount * 4); }
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(c->sp.create(0, false));
    c->side = c->sp.side;
    for (auto& ev : c->ev) CK(hipEventCreate(&ev));
    {   // fork / join of the two step kernels: both ends are on this GPU, so the events need no system-scope fence (cache write-back
        // and invalidate at every marker); PBRE_EVENT_FENCE=1 keeps it (A/B)
        const char* ef = getenv("PBRE_EVENT_FENCE");
        const unsigned fl = hipEventDisableTiming | ((ef && ef[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence);
        CK(hipEventCreateWithFlags(&c->ev_fork, fl));
        CK(hipEventCreateWithFlags(&c->ev_join, fl));
        CK(hipEventCreateWithFlags(&c->ev_join_sys, hipEventDisableTiming));
    }
    // (timing-only events around the dominant kernel: no system-scope fence at the markers)
    for (auto& pr : c->ev_k) for (auto& ev : pr) CK(hipEventCreateWithFlags(&ev, hipEventDisableSystemFence));
    CK(hipMalloc(&c->dT, sizeof(Tables)));
    CK(hipMemcpy(c->dT, &c->T, sizeof(Tables), hipMemcpyHostToDevice));
    CK(alloc_buf(c->main, c->npad));
    CK(alloc_buf(c->tmp, c->npad));
    CK(hipMalloc(&c->d_act, (size_t)c->npad * c->act_dim * sizeof(float)));
    CK(hipMalloc(&c->d_out, (size_t)c->npad * c->ow * sizeof(float)));
    CK(hipMalloc(&c->d_scratch, 64 * sizeof(float)));
    CK(hipMalloc(&c->d_bad, 2 * sizeof(int)));
    CK(hipMemset(c->d_bad, 0, 2 * sizeof(int)));
    c->P.bad_count = c->d_bad;
    CK(hipMalloc(&c->d_sweeps, (size_t)c->npad * sizeof(int)));
    CK(hipMemset(c->d_sweeps, 0, (size_t)c->npad * sizeof(int)));
    c->P.sweeps = c->d_sweeps;
    CK(hipMalloc(&c->d_ids, (size_t)(c->npad + EPB) * sizeof(unsigned long long)));
    CK(hipMalloc(&c->d_ep, (size_t)(c->npad + EPB) * sizeof(unsigned)));
    CK(hipMalloc(&c->d_idx, (size_t)c->npad * sizeof(int)));
    // every record of both buffers (incl. padding and dummy records) must hold a valid state: the un-settled reset pose
    {
        const int tot = c->npad + EPB;
        std::vector<unsigned long long> ids(tot, c->P.env_id_base); std::vector<unsigned> ep(tot, 0xFFFFFFFFu);   // episode -1
        CK(hipMemcpy(c->d_ids, ids.data(), (size_t)tot * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(c->d_ep, ep.data(), (size_t)tot * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_init, dim3((tot + 127) / 128), dim3(128), 0, c->stream, c->dT, c->P, c->main.state, c->d_ids, c->d_ep, tot);
        hipLaunchKernelGGL(k_init, dim3((tot + 127) / 128), dim3(128), 0, c->stream, c->dT, c->P, c->tmp.state, c->d_ids, c->d_ep, tot);
        CK(hipGetLastError());
        CK(classify(c, c->main, c->n, c->cfg.flags & PBRE_F_NO_OBJECT, c->stream));
        CK(hipStreamSynchronize(c->stream));
    }
#undef CK
    *out = c;
    return PBRE_OK;
}

const char* pbre_last_error(const pbre_ctx* c) { return c ? (c->wide ? wide_error(c->wide) : c->err.c_str()) : g_err.c_str(); }

int pbre_dims(const pbre_ctx* c, int32_t* od, int32_t* ad, int32_t* n) {
    if (!c) return PBRE_E_ARG;
    if (c->wide) { wide_dims(c->wide, od, ad, n, nullptr); return PBRE_OK; }
    if (od) *od = c->obs_dim;
    if (ad) *ad = c->act_dim;
    if (n) *n = c->n;
    return PBRE_OK;
}

int pbre_state_floats(const pbre_ctx* c) {
    if (!c) return PBRE_E_ARG;
    int32_t sf = STATE;
    if (c->wide) wide_dims(c->wide, nullptr, nullptr, nullptr, &sf);
    return sf;
}

int pbre_sync(pbre_ctx* c) {
    if (!c) return PBRE_E_ARG;
    if (c->wide) return wide_sync(c->wide);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    return PBRE_OK;
}

int pbre_observe(pbre_ctx* c, float* obs) {
    if (!c || !obs) return PBRE_E_ARG;
    if (c->wide) return wide_observe(c->wide, obs);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    hipLaunchKernelGGL(k_observe, dim3(c->npad / EPB), dim3(TPB), 0, c->stream, c->dT, c->P, c->main.state, c->d_out, c->d_scratch, c->n, c->ow);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy2DAsync(obs, (size_t)c->obs_dim * 4, c->d_out, (size_t)c->ow * 4, (size_t)c->obs_dim * 4, c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRE_OK;
}

int pbre_settle(pbre_ctx* c, int32_t n, int32_t flags) {
    if (!c || n < 0) return PBRE_E_ARG;
    if (c->wide) return wide_settle(c->wide, n, flags);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    const int f = flags & PBRE_F_NO_OBJECT, f0 = c->cfg.flags & PBRE_F_NO_OBJECT;
    if (f != f0) HIPCHK(classify(c, c->main, c->n, f, c->stream));           // classes depend on whether the object is present
    HIPCHK(settle_steps(c, c->main, c->n, n, f, c->stream));
    if (f != f0) HIPCHK(classify(c, c->main, c->n, f0, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRE_OK;
}

int pbre_reset(pbre_ctx* c, const uint8_t* mask, float* obs) {
    if (!c) return PBRE_E_ARG;
    if (c->wide) return wide_reset(c->wide, mask, obs);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    std::vector<int> idx;
    for (int e = 0; e < c->n; e++) if (!mask || mask[e]) idx.push_back(e);
    const int cnt = (int)idx.size();
    if (cnt > 0) {
        const int cpad = ceil16(cnt);
        std::vector<unsigned long long> ids(cpad);
        for (int i = 0; i < cpad; i++) ids[i] = c->P.env_id_base + (unsigned long long)idx[i < cnt ? i : cnt - 1];
        HIPCHK(hipMemcpyAsync(c->d_ids, ids.data(), (size_t)cpad * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_idx, idx.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, c->stream));
        // episode numbers live in the state records (the device advances them on auto-reset)
        const bool full = cnt == c->n;
        EnvBuf& work = full ? c->main : c->tmp;           // a partial reset settles a compacted copy
        hipLaunchKernelGGL(k_next_episode, dim3((cpad + 127) / 128), dim3(128), 0, c->stream, c->main.state, c->d_idx, cnt, cpad, c->d_ep, work.state);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));          // host vectors go out of scope below
        const int f0 = c->cfg.flags & PBRE_F_NO_OBJECT;
        hipLaunchKernelGGL(k_init, dim3((cpad + 127) / 128), dim3(128), 0, c->stream, c->dT, c->P, work.state, c->d_ids, c->d_ep, cpad);
        HIPCHK(hipGetLastError());
        // reset_simulation (panda_push_gym_env.py:117-148): 100 steps robot alone, then world loaded: 100 + 1 steps
        HIPCHK(classify(c, work, cnt, PBRE_F_NO_OBJECT, c->stream));
        if (c->P.use_ik) {     // pandaEnv.reset with use_IK (panda_env.py:83-91): IK targets of the home hand pose + one step
            hipLaunchKernelGGL(k_ik<true>, dim3((cnt + FTPB - 1) / FTPB), dim3(FTPB), 0, c->stream, c->dT, c->P, work.state, (const float*)nullptr, work.tgt, cnt, c->act_dim);
            HIPCHK(hipGetLastError());
            HIPCHK(settle_steps(c, work, cnt, 1, PBRE_F_NO_OBJECT, c->stream));
        }
        HIPCHK(settle_steps(c, work, cnt, 100, PBRE_F_NO_OBJECT, c->stream));
        if (!f0) HIPCHK(classify(c, work, cnt, 0, c->stream));
        HIPCHK(settle_steps(c, work, cnt, 101, f0, c->stream));
        hipLaunchKernelGGL(k_target, dim3((cpad + 127) / 128), dim3(128), 0, c->stream, c->P, work.state, c->d_ids, c->d_ep, cpad);
        HIPCHK(hipGetLastError());
        if (!full) {
            hipLaunchKernelGGL(k_scatter, dim3((cnt * STATE + 255) / 256), dim3(256), 0, c->stream, c->main.state, work.state, c->d_idx, cnt);
            HIPCHK(hipGetLastError());
            HIPCHK(classify(c, c->main, c->n, f0, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        if (full) {   // snapshot for PBRE_F_AUTO_RESET: settled robot pose and object height (identical in every env)
            float rec[STATE];
            HIPCHK(hipMemcpy(rec, c->main.state, sizeof rec, hipMemcpyDeviceToHost));
            for (int k = 0; k < NJ; k++) { c->P.rst_q[k] = rec[k]; c->T.rst_q[k] = rec[k]; }
            c->P.rst_objz = rec[11];
            HIPCHK(hipMemcpy(c->dT, &c->T, sizeof(Tables), hipMemcpyHostToDevice));
            c->have_snapshot = true; c->stale_snapshot = false;
            // end-effector pose of the settled robot (the first 6 observation entries of env 0) for the in-kernel restart
            hipLaunchKernelGGL(k_observe, dim3(c->npad / EPB), dim3(TPB), 0, c->stream, c->dT, c->P, c->main.state, c->d_out, c->d_scratch, c->n, c->ow);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream));
            float row[6]; int complex_now[NB] = {0};
            HIPCHK(hipMemcpy(row, c->d_out, sizeof row, hipMemcpyDeviceToHost));
            if (lane_per_env(c)) HIPCHK(hipMemcpy(complex_now, c->main.count + c->main.ccur * NB, sizeof complex_now, hipMemcpyDeviceToHost));
            for (int k = 0; k < 6; k++) c->P.rst_ee[k] = row[k];
            int nc = 0; for (int k = 0; k < NB; k++) nc += complex_now[k];
            c->P.rst_ok = nc == 0 ? 1 : 0;         // every env of the freshly reset batch is in the simple class
        }
    }
    c->k_steps = 0; c->launches = 0; c->launches3 = 0; c->launches_pair = 0; c->launches_fused = 0; c->launches_tail = 0;      // pbre_timing[3] averages env steps only, not the settle launches above
    if (obs) return pbre_observe(c, obs);
    return PBRE_OK;
}

int pbre_reset_snapshot(pbre_ctx* c, const uint8_t* mask, float* obs) {
    if (!c || !mask) return PBRE_E_ARG;
    if (c->wide) return wide_reset_snapshot(c->wide, mask, obs);
    if (!c->have_snapshot) { c->err = c->stale_snapshot ? stale_snapshot_msg() : "pbre_reset_snapshot: no settled snapshot yet (call pbre_reset for the whole batch first)"; return PBRE_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    if (!c->d_mask) HIPCHK(hipMalloc(&c->d_mask, (size_t)c->npad));
    HIPCHK(hipMemcpyAsync(c->d_mask, mask, (size_t)c->n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_snapshot_reset, dim3((c->n + 127) / 128), dim3(128), 0, c->stream, c->dT, c->P, c->main.state, c->d_mask, c->n);
    HIPCHK(hipGetLastError());
    HIPCHK(classify(c, c->main, c->n, c->cfg.flags & PBRE_F_NO_OBJECT, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (obs) return pbre_observe(c, obs);
    return PBRE_OK;
}

int pbre_step_device(pbre_ctx* c, const float* d_actions, float* d_out, void* stream) {
    if (!c || !d_actions || !d_out) return PBRE_E_ARG;
    if (c->wide) return wide_step_device(c->wide, d_actions, d_out, stream);
    HIPCHK(hipSetDevice(c->device));
    // PBRE_STREAM_LEGACY: the null stream (HIP's legacy default stream; the runtime torch bundles dereferences the symbolic
    // hipStreamLegacy handle, so it is passed as stream 0)
    hipStream_t s = stream == PBRE_STREAM_LEGACY ? (hipStream_t) nullptr : (stream ? (hipStream_t)stream : c->stream);
    if (stream) c->ext_dirty = true;
    if (c->stale_snapshot && (c->cfg.flags & PBRE_F_AUTO_RESET)) { c->err = stale_snapshot_msg(); return PBRE_E_ARG; }
    HIPCHK(full_step(c, d_actions, d_out, s));
    return PBRE_OK;
}

int pbre_step(pbre_ctx* c, const float* actions, float* out) {
    if (!c || !actions || !out) return PBRE_E_ARG;
    if (c->wide) return wide_step(c->wide, actions, out);
    HIPCHK(hipSetDevice(c->device));
    if (c->ext_dirty) HIPCHK(quiesce(c));
    if (c->stale_snapshot && (c->cfg.flags & PBRE_F_AUTO_RESET)) { c->err = stale_snapshot_msg(); return PBRE_E_ARG; }
    // zero-copy (PBRE_ZERO_COPY bit 0: actions, bit 1: rows): a page-locked buffer (pbre_host_alloc) is accessed by the kernels
    // themselves, over PCIe, instead of being staged through HBM with a copy
    bool za = false, zo = false;
    if (c->zero_copy) {
        hipPointerAttribute_t pa;
        za = (c->zero_copy & 1) && hipPointerGetAttributes(&pa, actions) == hipSuccess && pa.type == hipMemoryTypeHost;
        zo = (c->zero_copy & 2) && hipPointerGetAttributes(&pa, out) == hipSuccess && pa.type == hipMemoryTypeHost;
        (void)hipGetLastError();
    }
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if (!za) HIPCHK(hipMemcpyAsync(c->d_act, actions, (size_t)c->n * c->act_dim * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    c->rows_to_host = zo;
    const hipError_t fe = full_step(c, za ? actions : c->d_act, zo ? out : c->d_out, c->stream);
    c->rows_to_host = false;
    HIPCHK(fe);
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    if (!zo) HIPCHK(hipMemcpyAsync(out, c->d_out, (size_t)c->n * c->ow * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 3; i++) { float t = 0; HIPCHK(hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1])); c->ms[i] = t; }
    return PBRE_OK;
}

// ---- the pipelined host path (SURVEY 8(d)'s literal metric: action upload + kernels + row download).  pbre_step is host-synchronous: upload,
// kernels and download of ONE step follow each other (or, zero-copy, the kernels reach over PCIe themselves and are stretched by it:
// 0.44 ms instead of 0.15 at 131072 envs).  Here the three run on three streams with two buffer slots, so that in an open loop the DMA
// engines download the rows of step t (18.4 MB at 131072 envs: the PCIe floor, ~0.33 ms) while the kernels of step t + 1 run and the
// actions of step t + 2 come up.  Same kernels, same order per env: rows bit-equal to pbre_step's.
// rows out by a copy KERNEL (PBRE_ASYNC_D2H=1; A/B): coalesced 16-byte stores into the page-locked host buffer from a few waves on the
// download stream.  Measured (profiles/r06k_host_async_probe.txt): 54 GB/s alone, but beside k_fused -- which holds every wave slot -- the
// pipelined step is 0.56 ms against 0.47 with the DMA engine (hipMemcpyAsync, the default: 55 GB/s and no wave slots)
__global__ __launch_bounds__(256) void k_rows_out(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static int async_setup(pbre_ctx* c) {
    if (c->ap.ready) return PBRE_OK;
    for (int b = 0; b < 2; b++) {
        HIPCHK(hipMalloc(&c->ap.d_act[b], (size_t)c->n * c->act_dim * 4));
        HIPCHK(hipMalloc(&c->ap.d_rows[b], (size_t)c->n * c->ow * 4));
        for (hipEvent_t* e : {&c->ap.ev_step[b], &c->ap.ev_out[b]}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (const char* e = getenv("PBRE_ASYNC_BLOCKS")) c->async_blocks = std::max(1, atoi(e));
    c->ap.ready = true;
    return PBRE_OK;
}
int pbre_step_async(pbre_ctx* c, const float* actions, float* out) {
    if (!c || !actions || !out) return PBRE_E_ARG;
    if (c->wide) { c->err = "pbre_step_async: implemented for the Panda task envs (the BASELINE metric's path); use pbre_step"; return PBRE_E_UNSUPPORTED; }
    HIPCHK(hipSetDevice(c->device));
    if (c->stale_snapshot && (c->cfg.flags & PBRE_F_AUTO_RESET)) { c->err = stale_snapshot_msg(); return PBRE_E_ARG; }
    { const int rc = async_setup(c); if (rc != PBRE_OK) return rc; }
    pbre_ctx::AsyncPath& A = c->ap;
    if (A.issued - A.waited >= 2) { c->err = "pbre_step_async: two steps are in flight already -- pbre_step_wait first"; return PBRE_E_ARG; }
    if (c->ext_dirty) HIPCHK(quiesce(c));
    const int b = (int)(A.issued & 1);
    hipStream_t dl = c->sp.pick(c->stream);      // the stream that really runs beside c->stream
    // upload (in stream order: the step that last read this slot's action buffer is two steps back on the same stream), then the step --
    // behind the download that last read this slot's row buffer
    HIPCHK(hipMemcpyAsync(A.d_act[b], actions, (size_t)c->n * c->act_dim * 4, hipMemcpyHostToDevice, c->stream));
    if (A.issued >= 2) HIPCHK(hipStreamWaitEvent(c->stream, A.ev_out[b], 0));
    // PBRE_ASYNC_D2H: how the rows reach the host buffer -- 0 the DMA engine (hipMemcpyAsync), 1 a copy kernel on the download stream, 2 the
    // step kernels write them into the page-locked buffer themselves (pbre_step's zero-copy, minus its host synchronisation)
    static const int d2h_mode = [] { const char* e = getenv("PBRE_ASYNC_D2H"); return e ? atoi(e) : 0; }();
    const size_t bytes = (size_t)c->n * c->ow * 4;
    bool host_mapped = false;
    if (d2h_mode != 0) {
        hipPointerAttribute_t pa;
        host_mapped = hipPointerGetAttributes(&pa, out) == hipSuccess && pa.type == hipMemoryTypeHost;
        (void)hipGetLastError();
    }
    const bool direct = d2h_mode == 2 && host_mapped;
    c->rows_to_host = direct;
    const hipError_t fe = full_step(c, A.d_act[b], direct ? out : A.d_rows[b], c->stream);
    c->rows_to_host = false;
    HIPCHK(fe);
    HIPCHK(hipEventRecord(A.ev_step[b], c->stream));
    // download
    HIPCHK(hipStreamWaitEvent(dl, A.ev_step[b], 0));
    if (!direct) {
        const bool mapped = d2h_mode == 1 && (bytes % 16) == 0 && ((uintptr_t)out % 16) == 0 && host_mapped;
        if (mapped) {
            hipLaunchKernelGGL(k_rows_out, dim3(c->async_blocks), dim3(256), 0, dl, (const float4*)A.d_rows[b], (float4*)out, bytes / 16);
            HIPCHK(hipGetLastError());
        } else HIPCHK(hipMemcpyAsync(out, A.d_rows[b], bytes, hipMemcpyDeviceToHost, dl));
    }
    HIPCHK(hipEventRecord(A.ev_out[b], dl));
    A.issued++;
    return PBRE_OK;
}
int pbre_step_wait(pbre_ctx* c) {
    if (!c) return PBRE_E_ARG;
    if (c->wide) { c->err = "pbre_step_wait: no pbre_step_async on this engine"; return PBRE_E_UNSUPPORTED; }
    pbre_ctx::AsyncPath& A = c->ap;
    if (!A.ready || A.waited >= A.issued) { c->err = "pbre_step_wait: no step in flight"; return PBRE_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    // (poll the slot's event: hipEventSynchronize returned only once the NEWER download enqueued on the same stream was done too -- the
    // pipeline then runs one step deep)
    for (;;) {
        const hipError_t q = hipEventQuery(A.ev_out[A.waited & 1]);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { c->err = std::string("hipEventQuery: ") + hipGetErrorString(q); return PBRE_E_DEVICE; }
        for (int i = 0; i < 64; i++) __builtin_ia32_pause();
    }
    (void)hipGetLastError();
    A.waited++;
    return PBRE_OK;
}

int pbre_get_state(pbre_ctx* c, float* s) {
    if (!c || !s) return PBRE_E_ARG;
    if (c->wide) return wide_get_state(c->wide, s);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    HIPCHK(hipMemcpy(s, c->main.state, (size_t)c->n * STATE * 4, hipMemcpyDeviceToHost));
    return PBRE_OK;
}
int pbre_set_state(pbre_ctx* c, const float* s) {
    if (!c || !s) return PBRE_E_ARG;
    if (c->wide) return wide_set_state(c->wide, s);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    HIPCHK(hipMemcpy(c->main.state, s, (size_t)c->n * STATE * 4, hipMemcpyHostToDevice));
    HIPCHK(classify(c, c->main, c->n, c->cfg.flags & PBRE_F_NO_OBJECT, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRE_OK;
}

int pbre_get_state_cols(pbre_ctx* c, int32_t first, int32_t count, float* out) {
    if (!c || !out || first < 0 || count <= 0 || first + count > pbre_state_floats(c)) return PBRE_E_ARG;
    if (c->wide) return wide_get_state_cols(c->wide, first, count, out);
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    HIPCHK(hipMemcpy2D(out, (size_t)count * 4, c->main.state + first, (size_t)STATE * 4, (size_t)count * 4, c->n, hipMemcpyDeviceToHost));
    return PBRE_OK;
}
void* pbre_host_alloc(size_t bytes) {
    void* p = nullptr;
    // PBRE_HOST_NONCOHERENT=1 (A/B with PBRE_ZERO_COPY): coarse-grained host memory, device writes are cached in L2 and written back at
    // the end of the kernel in full lines instead of going out as they are issued
    const char* nc = getenv("PBRE_HOST_NONCOHERENT");
    const unsigned flags = (nc && nc[0] == '1') ? (hipHostMallocNonCoherent | hipHostMallocMapped) : hipHostMallocDefault;
    return hipHostMalloc(&p, bytes ? bytes : 1, flags) == hipSuccess ? p : nullptr;
}
void pbre_host_free(void* p) { if (p) (void)hipHostFree(p); }

int pbre_set_motors(pbre_ctx* c, int32_t n, const int32_t* dofs, const float* targets, double kp, double max_force, double max_vel, const uint8_t* mask) {
    if (!c || n < 0 || (n > 0 && (!dofs || !targets))) return PBRE_E_ARG;
    if (c->wide) return wide_set_motors(c->wide, n, dofs, targets, kp, max_force, max_vel, mask);
    c->err = "pbre_set_motors: only the robot-level engines (pbre_config.robot_level) keep a motor record";
    return PBRE_E_UNSUPPORTED;
}
int pbre_apply_action(pbre_ctx* c, const float* actions, double max_vel) {
    if (!c || !actions) return PBRE_E_ARG;
    if (c->wide) return wide_apply_action(c->wide, actions, max_vel);
    c->err = "pbre_apply_action: only the robot-level engines (pbre_config.robot_level) keep a motor record";
    return PBRE_E_UNSUPPORTED;
}
int pbre_get_motor_state(pbre_ctx* c, float* m) {
    if (!c || !m) return PBRE_E_ARG;
    if (c->wide) return wide_motor_state(c->wide, m, nullptr);
    c->err = "pbre_get_motor_state: only the robot-level engines keep a motor record";
    return PBRE_E_UNSUPPORTED;
}
int pbre_set_motor_state(pbre_ctx* c, const float* m) {
    if (!c || !m) return PBRE_E_ARG;
    if (c->wide) return wide_motor_state(c->wide, nullptr, m);
    c->err = "pbre_set_motor_state: only the robot-level engines keep a motor record";
    return PBRE_E_UNSUPPORTED;
}
int pbre_get_physics(const pbre_ctx* c, pbre_physics* phys) {
    if (!c || !phys) return PBRE_E_ARG;
    if (c->wide) return wide_get_physics(c->wide, phys);
    *phys = c->cfg.phys;
    return PBRE_OK;
}
int pbre_get_sweeps(pbre_ctx* c, int32_t* sweeps) {
    if (!c || !sweeps) return PBRE_E_ARG;
    if (c->wide) return wide_get_sweeps(c->wide, sweeps);
    if (!(c->P.res_lim > 0.f)) { c->err = "pbre_get_sweeps: pbre_physics.solver_residual_threshold is 0 (every env runs all solver_iters sweeps)"; return PBRE_E_UNSUPPORTED; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    HIPCHK(hipMemcpy(sweeps, c->d_sweeps, (size_t)c->n * sizeof(int), hipMemcpyDeviceToHost));
    return PBRE_OK;
}
int pbre_set_physics(pbre_ctx* c, const pbre_physics* phys) {
    if (!c || !phys) return PBRE_E_ARG;
    if (c->wide) return wide_set_physics(c->wide, phys);
    pbre_config cfg = c->cfg;
    cfg.phys = *phys;
    Params P2 = c->P;
    if (!apply_physics(*phys, P2)) { c->err = "bad physics parameters"; return PBRE_E_ARG; }
    if (c->fast_ok && !fast_scene_ok(P2)) { c->err = "the lane-per-env kernels need explicit joint damping"; return PBRE_E_UNSUPPORTED; }
    if (snapshot_relevant_change(c->cfg.phys, *phys)) {      // (round-2 advice) restarts would put the object at the old scene's rest height
        c->stale_snapshot = c->stale_snapshot || c->have_snapshot;
        c->have_snapshot = false; P2.rst_ok = 0;
    }
    c->cfg = cfg; c->P = P2;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    HIPCHK(classify(c, c->main, c->n, c->cfg.flags & PBRE_F_NO_OBJECT, c->stream));     // the contact margin may have changed
    HIPCHK(hipStreamSynchronize(c->stream));
    return PBRE_OK;
}

int pbre_set_object_hull(pbre_ctx* c, const double* verts, int32_t n_verts) {
    if (!c) return PBRE_E_ARG;
    if (c->wide) return wide_set_object_hull(c->wide, verts, n_verts);
    HullTable H;
    const std::string e = build_hull(verts, n_verts, H);
    if (!e.empty()) { c->err = e; return PBRE_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    if (!c->d_hull) HIPCHK(hipMalloc(&c->d_hull, sizeof H.data));
    HIPCHK(hipMemcpy(c->d_hull, H.data, sizeof H.data, hipMemcpyHostToDevice));
    c->P.hull = c->d_hull; c->P.hull_nv = H.nv; c->P.hull_nf = H.nf; c->P.hull_rb = H.rb; c->P.obj_shape = PBRE_SHAPE_HULL;
    c->cfg.phys.obj_shape = PBRE_SHAPE_HULL;
    for (int k = 0; k < 3; k++) { c->cfg.phys.obj_h[k] = H.half[k]; c->P.obj_h[k] = (float)H.half[k]; }
    c->P.rst_objz = (float)(c->cfg.h_table + H.half[2]);
    c->stale_snapshot = c->stale_snapshot || c->have_snapshot;      // a scene change: restarts from the old scene's settled snapshot are refused
    c->have_snapshot = false; c->P.rst_ok = 0;
    return PBRE_OK;
}

int pbre_set_physics_per_env(pbre_ctx* c, const uint8_t* mask, const float* obj_mass, const float* obj_mu, const float* obj_lin_damping,
                             const float* robot_lin_damping) {
    if (!c) return PBRE_E_ARG;
    if (c->wide) { c->err = "pbre_set_physics_per_env: implemented for the Panda task envs (change_physics_params, panda_push_gym_env.py:362-368)"; return PBRE_E_UNSUPPORTED; }
    for (int e = 0; e < c->n; e++) {
        if (mask && !mask[e]) continue;
        if ((obj_mass && !(obj_mass[e] > 0.f)) || (obj_mu && !(obj_mu[e] > 0.f)) || (obj_lin_damping && !(obj_lin_damping[e] >= 0.f)) ||
            (robot_lin_damping && !(robot_lin_damping[e] >= 0.f))) {
            c->err = "pbre_set_physics_per_env: mass and friction must be > 0, damping >= 0"; return PBRE_E_ARG;
        }
    }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(quiesce(c));
    // strided columns of the state records: X[12] mass, X[13] lateral friction, X[15] 1 + linear damping of the object, V[15] 1 + linear
    // damping of the robot's links (0 = batch value)
    std::vector<float> col((size_t)c->n);
    const float* src[4] = {obj_mass, obj_mu, obj_lin_damping, robot_lin_damping};
    const int slot[4] = {44, 45, 47, 31};
    for (int k = 0; k < 4; k++) {
        if (!src[k]) continue;
        HIPCHK(hipMemcpy2D(col.data(), 4, c->main.state + slot[k], (size_t)STATE * 4, 4, c->n, hipMemcpyDeviceToHost));
        for (int e = 0; e < c->n; e++) if (!mask || mask[e]) col[e] = src[k][e] + (k >= 2 ? 1.f : 0.f);
        HIPCHK(hipMemcpy2D(c->main.state + slot[k], (size_t)STATE * 4, col.data(), 4, 4, c->n, hipMemcpyHostToDevice));
    }
    return PBRE_OK;
}

int pbre_obs_limits(const pbre_ctx* c, float* lo, float* hi) {
    if (!c || !lo || !hi) return PBRE_E_ARG;
    if (c->wide) return wide_obs_limits(c->wide, lo, hi);
    obs_limits(c->cfg, c->T, lo, hi);
    return PBRE_OK;
}
int pbre_timing(const pbre_ctx* c, double* ms, int32_t n) {
    if (!c || !ms) return PBRE_E_ARG;
    if (c->wide) return wide_timing(c->wide, ms, n);
    double kd = 0.0;
    if (n > 3 && c->k_steps > 0) {      // mean over the last min(k_steps, KRING) steps
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        const long cnt = std::min<long>(c->k_steps, pbre_ctx::KRING);
        int ok = 0;
        for (long i = 0; i < cnt; i++) {
            float t = 0.f;
            hipEvent_t* ek = const_cast<pbre_ctx*>(c)->ev_k[(c->k_steps - 1 - i) % pbre_ctx::KRING];
            if (hipEventElapsedTime(&t, ek[0], ek[1]) == hipSuccess) { kd += t; ok++; }
        }
        kd = ok ? kd / ok : 0.0;
    }
    for (int i = 0; i < n; i++) ms[i] = i < 3 ? c->ms[i] : (i == 3 ? kd : 0.0);
    return PBRE_OK;
}
int pbre_kernel_info(const pbre_ctx* c, int32_t* info, int32_t n) {
    if (!c || !info) return PBRE_E_ARG;
    if (c->wide) return wide_kernel_info(c->wide, info, n);
    hipFuncAttributes fa;
    int rf = -1, rg = -1, rr = -1, rf3 = -1, rp = -1, ru = -1;
    if (hipFuncGetAttributes(&fa, (const void*)k_fused<MODE_STEP, false>) == hipSuccess) ru = fa.numRegs;
    if (hipFuncGetAttributes(&fa, (const void*)k_fused<MODE_STEP, true>) == hipSuccess) ru = std::max(ru, (int)fa.numRegs);
    if (hipFuncGetAttributes(&fa, (const void*)k_fast<MODE_STEP, 3>) == hipSuccess) rf3 = fa.numRegs;
    if (hipFuncGetAttributes(&fa, (const void*)k_fast_pair<MODE_STEP>) == hipSuccess) rp = fa.numRegs;
    if (hipFuncGetAttributes(&fa, (const void*)k_fast<MODE_STEP, 2>) == hipSuccess) rf = fa.numRegs;
    if (hipFuncGetAttributes(&fa, (const void*)k_step<MODE_STEP>) == hipSuccess) rg = fa.numRegs;
    if (hipFuncGetAttributes(&fa, (const void*)k_fast_rc<MODE_STEP>) == hipSuccess) rr = fa.numRegs;
    int complex_now = 0, complex_sum = 0, bad = 0;       // envs whose current state is "complex" (what the next step's k_fast_rc will take)
    const bool lpe = lane_per_env(c);
    if (lpe) {
        int cnt[NB] = {0};
        (void)hipSetDevice(c->device); (void)hipDeviceSynchronize();
        (void)hipMemcpy(cnt, c->main.count + c->main.ccur * NB, NB * sizeof(int), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&complex_sum, c->main.count + 3 * NB + 1, sizeof(int), hipMemcpyDeviceToHost);
        for (int k = 0; k < NB; k++) complex_now += cnt[k];
    }
    if (c->d_bad) {
        (void)hipSetDevice(c->device); (void)hipDeviceSynchronize();
        (void)hipMemcpy(&bad, c->d_bad, sizeof(int), hipMemcpyDeviceToHost);
    }
    const int v[16] = {lpe ? rf : -1, rg, lpe ? 1 : 0, lpe ? c->n - complex_now : 0, lpe ? 0 : c->n, complex_now, lpe ? rr : -1, complex_sum,
                       (int)(c->launches3 & 0x7fffffff), lpe ? rf3 : -1, (int)(c->launches_pair & 0x7fffffff), lpe ? rp : -1, bad,
                       (int)(c->launches_fused & 0x7fffffff), lpe ? ru : -1, (int)(c->launches_tail & 0x7fffffff)};
    for (int i = 0; i < n; i++) info[i] = i < 16 ? v[i] : 0;
    return PBRE_OK;
}

#ifdef PBRE_PHASE_PROBE
int pbre_debug_probe(unsigned long long* out, int reset) {      // (probe builds only; not part of include/pbre.h)
    unsigned long long z[64] = {0};
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof z) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif

}  // extern "C"
