// pbre_lane.hip -- the Shape32 engine (iCub as simulated, 20 DoF): the lane-group engine of pbre_wide.hip / pbre_wide_impl.hpp with the
// lane-per-env kernels of pbre_lane.hpp on top for the steps of the task envs.  Its own translation unit: the fully unrolled 20-link
// code needs -mllvm -pragma-unroll-threshold far above the default (build.sh), which the other kernels are not compiled with.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "pbre_wide_impl.hpp"
#include "pbre_sidepick.hpp"

// lane-per-env path of the 20-DoF iCub (pbre_lane.hpp): device definitions of its hooks
#define PBRE_ANY(x) (__any((int)(x)) != 0)
#define PBRE_REG_BARRIER() asm volatile("" ::: "memory")
#define PBRE_LAUNDER(p) asm volatile("" : "+s"(p))
#define PBRE_OPAQUE_I(x) asm volatile("" : "+v"(x))
#define PBRE_OPAQUE_F(x) asm volatile("" : "+v"(x))
#define PBRE_NOUNROLL _Pragma("nounroll")
// Per-env hand-over of the IK targets (round 5; re-done in round 6): kw_lane_ik hands an env's joint targets to the solve kernels in the
// iteration in which THAT env converges; the quads wait for their own env's targets only.
// Round 5 wrote the 20 targets and then a per-env "done" mark (relaxed agent-scope stores around a workgroup-scope fence).  Nothing ordered
// the mark behind the targets across XCDs, an s_waitcnt vmcnt(0) between them was not enough either -- a write-through store is
// acknowledged before it is visible to another XCD's loads -- and round 6's contention test (tests/test_gpu_contention.py: a second context
// keeps the GPU busy, so the quads really do spin and read the moment the mark appears) produced rows that differed from the solo run's.
// A release / acquire pair at agent scope would order them, but is an L2 write-back / invalidate per publishing lane on this chip (measured
// in round 5: kw_dyn 93 -> 207 us).  So no ordering between addresses is needed any more: every target travels as ONE 64-bit word
// (value | sequence number << 32) written and read with 64-bit relaxed agent-scope atomics -- single-copy atomic, coherent across the XCDs --
// and a quad lane polls its own five words until all of them carry this launch's number.  The plain target array is still written (plain
// stores: the kernels of later steps read it after the IK kernel has ended).
// The wait is bounded (~1 s): the IK kernel is launched first, on a stream of the highest priority, and waits for nothing; a quad that
// does reach the bound POISONS its env: it writes NaN into the env's guard element of the side buffer (element 214 of quad
// lane 0, where kw_dyn recorded whether the incoming state was finite), and the NaN / Inf guard in kw_fin returns the env-step as done = 1 /
// reward 0, restarts the env under PBRE_F_AUTO_RESET and counts it once (pbre_kernel_info[12]).
#define PBRE_IK_STORE(done, seq, j, p, v) do { const float v_ = (v); *(p) = v_; \
        if (done) __hip_atomic_store((unsigned long long*)(done) + (j), ((unsigned long long)(unsigned)(seq) << 32) | (unsigned long long)__float_as_uint(v_), \
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
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
// Cartesian control: hand-pose update + inverse kinematics -> joint targets, one thread per env
__global__ __launch_bounds__(LTPB, 2) void kw_lane_ik(const TablesT<Shape32>* __restrict__ T, const Params P, float* __restrict__ state,
                                                   const float* __restrict__ actions, float* __restrict__ tgt, int n, int act_dim,
                                                   unsigned long long* __restrict__ ik_done, int ik_seq) {
    // (a higher wave priority for this kernel -- its last waves are the head of the step's critical path beside the solve kernels -- was
    // measured: the bulk of the IK then starves kw_dyn, 95 -> 161 us, and the step gets longer; profiles/r05u_icub_timeline_ab.txt)
    const int env = blockIdx.x * LTPB + threadIdx.x;
    if (env >= n) return;
    LaneD::ik_targets(*T, P, state + (size_t)env * Shape32::STATE, actions + (size_t)env * act_dim, tgt + (size_t)env * Shape32::TGT,
                      ik_done ? (int*)(ik_done + (size_t)env * LaneD::ND) : nullptr, ik_seq);      // (the env's box: ND words)
}

// ------------------------------------------------------------------ the pipeline: kw_dyn -> kw_quad (+ kw_quad_rc) -> kw_fin
// (Lane::step as ONE kernel -- M^-1 in 40 KB of LDS per wave, a whole SIMD's register file -- ran as a lone, latency-bound wave per SIMD
// and was slower than the lane-group kernel; see pbre_lane.hpp.)  The pipeline splits the step where the data layouts want to differ:
//   kw_dyn   one thread per env: kinematics + dynamics (Lane::dynamics) -> bias torques and the joint-space inertia M, streamed to an
//            HBM side buffer (no LDS, no M^-1 in this kernel);
//   kw_quad  FOUR lanes per env (a DPP quad; 16 envs per wave): lane r owns DoF 5r..5r+4, i.e. five rows of M -- 100 registers, no
//            LDS.  Gauss-Jordan inversion across the quad, unconstrained velocities, motor / limit / robot-table contact rows, the 150
//            PGS sweeps (a motor row is one fma for its delta, a quad broadcast and the five fmacs of the velocity update), integration
//            of the joints; kw_quad_rc: the same for the envs with robot-object contacts (compacted list), the object's twist and
//            table rows inside the same sweeps;
//   kw_fin   one thread per env: the object's pose from kw_obj's twist, observation / reward / termination / auto-reset / class
//            (Lane::finish).
// Side buffer layout [quad lane r][element e][env]: per quad lane 148 elements -- 0..99 rows 5r..5r+4 of M (i * 20 + col), 100..104
// tau[5r..5r+4], 108..113 (lane 0 only) robot-table contact flags / friction / distances, 116..145 this lane's 2 x 3 x 5 contact
// Jacobian entries; envs of the complex class (robot-object contact): 148..177 this lane's robot-object Jacobian entries, 178..207 (lane
// 0) flags / friction / distances / contact frames / lever arms of the two slots, 208..213 (lane 0) the object's twist after the
// coupled solve (kw_quad_rc -> kw_fin).  kw_dyn (one thread per env) writes 256 contiguous bytes per store, a wave of kw_quad (16
// envs x 4 lanes) reads four 64-byte segments per load.
constexpr int QD = 5, DEL = 216;      // elements per quad lane (below)
static_assert(LaneD::ND == 4 * QD, "four lanes per env, five DoF each");
struct DynSink {
    float* base;        // dyn + env
    size_t cs;          // floats between consecutive elements: n_pad
    __device__ __forceinline__ void f(int r, int e, float v) { base[(size_t)(r * DEL + e) * cs] = v; }      // element e of quad lane r: a wave's store is 256 contiguous bytes
    __device__ __forceinline__ void st(int row, int col, float v) { f(row / QD, (row % QD) * 20 + col, v); }
    __device__ __forceinline__ void put(int j, int i, float v) { st(j, i, v); if (i != j) st(i, j, v); }
    __device__ __forceinline__ void zero(int, int) {}      // unrelated branch pairs: the buffer is zeroed once, nobody writes them
    __device__ __forceinline__ void tau(int j, float v) { f(j / QD, 100 + (j % QD), v); }
    // contact Jacobian rows, straight to the buffer (rows of an env without the contact are zeros; kw_quad masks them anyway)
    __device__ __forceinline__ void rt_j(int c, int d, int j, float v) { f(j / QD, 116 + (c * 3 + d) * QD + (j % QD), v); }
    __device__ __forceinline__ void rt_none(int) {}
    __device__ __forceinline__ void ro_j(int c, int d, int j, float v) { f(j / QD, 148 + (c * 3 + d) * QD + (j % QD), v); }
};
__global__ __launch_bounds__(LTPB, 1) void kw_dyn(const TablesT<Shape32>* __restrict__ T, const Params P, const float* __restrict__ state, int n, int flags,
                                                  const signed char* __restrict__ cls_cur, float* __restrict__ dyn, size_t cs) {
    const int env = blockIdx.x * LTPB + threadIdx.x;
    if (env >= n) return;
    const float* st = state + (size_t)env * Shape32::STATE;
    constexpr int LC = Shape32::LC;
    float q[LaneD::ND], qd[LaneD::ND], tau[LaneD::ND];
    PBRE_UNROLL for (int j = 0; j < LaneD::ND; j++) { q[j] = st[j]; qd[j] = st[Shape32::W + j]; }
    DynSink sink; sink.base = dyn + env; sink.cs = cs;
    {   // NaN / Inf guard (Fast::step_t): 0 while the incoming state is finite, NaN otherwise; element 214 of quad lane 0, which kw_fin adds to
        // a position of the new state (the clamps of the solve in between would turn a non-finite velocity into finite garbage)
        float fin_in = 0.f;
        PBRE_UNROLL for (int j = 0; j < LaneD::ND; j++) { fin_in = fmaf(q[j], 0.f, fin_in); fin_in = fmaf(qd[j], 0.f, fin_in); }
        if (!(flags & 1)) { PBRE_UNROLL for (int k = 0; k < 7; k++) fin_in = fmaf(st[LC + k], 0.f, fin_in); PBRE_UNROLL for (int k = 0; k < 6; k++) fin_in = fmaf(st[Shape32::W + LC + k], 0.f, fin_in); }
        sink.f(0, 214, fin_in);
    }
    if (st[2 * Shape32::W + 14] != 0.f) return;          // left the apply_action loop: no simulation step
    LaneD::RtC rt;
    LaneD::RoC ro;
    const bool rc = cls_cur[env] != 0 && !(flags & 1);   // complex class: a robot sphere within the contact margin of the object
    const bool any_rc = __any((int)rc) != 0;
    LaneD::V3 op; op.x = st[LC]; op.y = st[LC + 1]; op.z = st[LC + 2];
    LaneD::Q4 oq; oq.x = st[LC + 3]; oq.y = st[LC + 4]; oq.z = st[LC + 5]; oq.w = st[LC + 6];
    LaneD::dynamics(*T, P, q, qd, sink, tau, rt, any_rc ? &ro : nullptr, op, oq);
    PBRE_UNROLL for (int j = 0; j < LaneD::ND; j++) sink.tau(j, tau[j]);
    // robot-table contact slots: flags / friction / distances always, the Jacobian rows of the envs that have a contact
    sink.f(0, 108, rt.act[0] ? 1.f : 0.f); sink.f(0, 109, rt.act[1] ? 1.f : 0.f); sink.f(0, 110, rt.mu[0]); sink.f(0, 111, rt.mu[1]);
    sink.f(0, 112, rt.dist[0]); sink.f(0, 113, rt.dist[1]);
    if (any_rc && rc) {          // robot-object contact slots of a complex env
        PBRE_UNROLL for (int c = 0; c < LaneD::NRO; c++) {
            sink.f(0, 178 + c, ro.act[c] ? 1.f : 0.f); sink.f(0, 180 + c, ro.mu[c]); sink.f(0, 182 + c, ro.dist[c]);
            PBRE_UNROLL for (int d = 0; d < 3; d++) {
                sink.f(0, 184 + (c * 3 + d) * 3, ro.dir[c][d].x); sink.f(0, 185 + (c * 3 + d) * 3, ro.dir[c][d].y); sink.f(0, 186 + (c * 3 + d) * 3, ro.dir[c][d].z);
            }
            sink.f(0, 202 + c * 3, ro.rB[c].x); sink.f(0, 203 + c * 3, ro.rB[c].y); sink.f(0, 204 + c * 3, ro.rB[c].z);
        }
    }
}

// value of quad lane o in every lane of the quad (o static after unrolling)
template <int O> __device__ __forceinline__ float qb_t(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), O * 0x55, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ float qb_x(float x) {      // quad_perm CTRL (0xB1: lanes 1,0,3,2; 0x4E: lanes 2,3,0,1)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float qb(float x, int o) { return o == 0 ? qb_t<0>(x) : (o == 1 ? qb_t<1>(x) : (o == 2 ? qb_t<2>(x) : qb_t<3>(x))); }

// The solve of one env by the four lanes of a quad.  RC: the env has robot-object contacts -- the object's twist and its rows against the
// table (ObjStep, evaluated by all four lanes alike) are part of the same sweeps, coupled to the joints through the robot-object rows.
template <bool RC>
__device__ __forceinline__ void quad_step(const TablesT<Shape32>* __restrict__ T, const Params& P, float* __restrict__ state,
                                          const float* __restrict__ actions, int act_dim, int MODE, const float* __restrict__ tgt,
                                          float* __restrict__ dyn, size_t cs, int env, int r, const unsigned long long* __restrict__ ik_done = nullptr, int ik_seq = 0) {
    constexpr int ND = LaneD::ND, W = Shape32::W, XO = 2 * Shape32::W;
    float* st = state + (size_t)env * Shape32::STATE;
    if (st[XO + 14] != 0.f) return;
    const float dt = P.dt, inv_dt = P.inv_dt, vmax = P.vmax;
    float own[4];                                          // 1 on the lane that owns block o
    PBRE_UNROLL for (int o = 0; o < 4; o++) own[o] = r == o ? 1.f : 0.f;

    // ---- rows 5r..5r+4 of M and the bias torques of this lane's DoF
    float A[QD][ND], tl[QD];
    {
        const float* dl = dyn + (size_t)r * DEL * cs + env;      // this lane's elements: dl[e * cs]
        PBRE_UNROLL for (int e = 0; e < QD * ND; e++) A[e / ND][e % ND] = dl[(size_t)e * cs];
        PBRE_UNROLL for (int i = 0; i < QD; i++) tl[i] = dl[(size_t)(100 + i) * cs];
    }
    const int d0 = QD * r;                                 // this lane's first DoF
    float q[QD], qd[QD];
    PBRE_UNROLL for (int i = 0; i < QD; i++) { q[i] = st[d0 + i]; qd[i] = st[W + d0 + i]; }
    if (P.jd_dt != 0.f) {                                  // implicit joint damping: M + dt C
        PBRE_UNROLL for (int i = 0; i < QD; i++) {
            const float a = P.jd_dt * T->jdamp[d0 + i];
            PBRE_UNROLL for (int o = 0; o < 4; o++) A[i][QD * o + i] = fmaf(a, own[o], A[i][QD * o + i]);
        }
    }
    // ---- M^-1 by in-place Gauss-Jordan (SPD, no pivoting) across the quad: the pivot row is broadcast from its owner; the owner's own
    //      copy of it is brought to pivot_row / pivot by the same fma as every other row (with f - 1 in place of f)
    PBRE_UNROLL for (int c = 0; c < ND; c++) {
        const int o = c / QD, i0 = c % QD;
        const float inv = 1.f / qb(A[i0][c], o);
        float fp[QD], off[QD];
        PBRE_UNROLL for (int i = 0; i < QD; i++) { const float f = A[i][c]; fp[i] = i == i0 ? f - own[o] : f; off[i] = -f * inv; }
        PBRE_UNROLL for (int k = 0; k < ND; k++) {
            if (k == c) continue;
            const float rc = qb(A[i0][k], o) * inv;            // pivot-row entry / pivot (read before the owner's row is updated below)
            PBRE_UNROLL for (int i = 0; i < QD; i++) A[i][k] = fmaf(-fp[i], rc, A[i][k]);
        }
        PBRE_UNROLL for (int i = 0; i < QD; i++) A[i][c] = i == i0 ? (r == o ? inv : off[i]) : off[i];
    }
    // ---- unconstrained velocities v* = qd + dt M^-1 tau, motor rows against the running velocity, limit rows (see Lane::step)
    float w[QD], w0[QD], m_dinv[QD], m_rhs[QD], sabs[QD], l_dir[QD], l_rhs[QD], l_app[QD];
    float ik_t[QD] = {0.f, 0.f, 0.f, 0.f, 0.f};      // Cartesian control with the per-env hand-over: this lane's targets out of the env's box
    unsigned long long lim_b[QD];
    {
        float acc[QD];
        PBRE_UNROLL for (int i = 0; i < QD; i++) acc[i] = 0.f;
        PBRE_UNROLL for (int c = 0; c < ND; c++) {
            const float tc = qb(tl[c % QD], c / QD);
            PBRE_UNROLL for (int i = 0; i < QD; i++) acc[i] = fmaf(A[i][c], tc, acc[i]);
        }
        // Cartesian control: this env's IK targets are read next -- wait for ITS mark (the IK kernel runs beside this one)
        if ((MODE & LaneD::M_TGT) && ik_done) {
            const unsigned long long* box = ik_done + (size_t)env * ND + d0;      // this quad lane's five (value, sequence) words
            // Poll ONE word per ENV -- the last one the IK lane writes, the same address for the four lanes of the quad -- and read this lane's
            // own five only once that one carries the launch's number (and go on polling if one of them does not yet: nothing orders the
            // words).  Per-lane polling (five words, then one) cost 10 x the memory transactions of round 5's per-env mark and slowed the IK
            // kernel everybody waits for: iCub reach 0.38 -> 0.62 ms per step at 32768 envs (profiles/r06_icub_handover.txt).
            const unsigned long long* last = ik_done + (size_t)env * ND + (ND - 1);
            int spins = 0;
            for (;;) {
                const unsigned long long xl = __hip_atomic_load(last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool all = (int)(xl >> 32) == ik_seq;
                if (all) {
                    PBRE_UNROLL for (int i = 0; i < QD; i++) {
                        const unsigned long long x = __hip_atomic_load(box + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        all = all && (int)(x >> 32) == ik_seq;
                        ik_t[i] = __uint_as_float((unsigned)x);
                    }
                }
                if (all) break;
                __builtin_amdgcn_s_sleep(16);
                if (++spins > (1 << 21)) { dyn[(size_t)214 * cs + env] = __builtin_nanf(""); break; }
            }
        }
        PBRE_UNROLL for (int i = 0; i < QD; i++) {
            const int d = d0 + i;
            const float wj = fminf(fmaxf(fmaf(dt, acc[i], qd[i]), -vmax), vmax);
            w[i] = wj; w0[i] = wj;
            float qdes = T->home[d], kp = T->kp_hold[d], kd = T->kd_hold[d];
            const float lo = T->lower[d], up = T->upper[d];
            if (MODE & LaneD::M_TGT) qdes = ik_done ? ik_t[i] : tgt[(size_t)env * Shape32::TGT + d];
            if (MODE & LaneD::M_ACTION) {
                kp = T->kp_act[d]; kd = T->kd_act[d];
                const int ai = T->act_idx[d];
                if (ai >= 0) qdes = fminf(fmaxf(fmaf(actions[(size_t)env * act_dim + ai], P.act_scale, q[i]), lo), up);
            }
            float diag = 0.f;
            PBRE_UNROLL for (int o = 0; o < 4; o++) diag = fmaf(own[o], A[i][QD * o + i], diag);
            m_dinv[i] = 1.f / diag;
            m_rhs[i] = (kp * (qdes - q[i]) * inv_dt + (1.f - kd) * wj) * m_dinv[i];
            sabs[i] = 0.f;
            const float pl = q[i] - lo, pu = up - q[i];
            const bool lo_v = pl <= 0.f, up_v = !lo_v && pu <= 0.f;
            l_dir[i] = lo_v ? 1.f : (up_v ? -1.f : 0.f);
            const float pen = lo_v ? pl : pu;
            l_rhs[i] = (lo_v || up_v) ? (-pen * P.erp * inv_dt) * m_dinv[i] : 0.f;
            l_app[i] = 0.f;
            lim_b[i] = __ballot(lo_v || up_v);             // which quad lanes of the wave have their DoF 5r+i at a limit
        }
    }
    bool lim_on[ND], has_limit = false;                    // row j = 5 o + i runs if any env of the wave has joint j at a limit
    PBRE_UNROLL for (int j = 0; j < ND; j++) { lim_on[j] = (lim_b[j % QD] & (0x1111111111111111ull << (j / QD))) != 0ull; has_limit = has_limit || lim_on[j]; }

    // ---- robot-table contact rows (the slots kw_dyn found): this lane's five entries of J and of B = M^-1 J^T per row; the row's
    //      scalars (1 / (J B), applied impulse, rhs) are computed by every lane of the quad alike
    constexpr int NRT = LaneD::NRT;
    float rJ[NRT][3][QD], rB[NRT][3][QD], r_dinv[NRT][3], r_app[NRT][3], r_rhs[NRT], r_mu[NRT];
    bool rt_on[NRT];
    {
        const float* d0p = dyn + env;                         // the env's quad-lane-0 elements
        const float a0 = d0p[(size_t)108 * cs], a1 = d0p[(size_t)109 * cs];
        rt_on[0] = __any((int)(a0 != 0.f)) != 0; rt_on[1] = __any((int)(a1 != 0.f)) != 0;      // wave-uniform; rows of a quad without the contact are exact no-ops
        PBRE_UNROLL for (int c = 0; c < NRT; c++) {
            r_rhs[c] = 0.f; r_mu[c] = 0.f;
            PBRE_UNROLL for (int d = 0; d < 3; d++) { r_dinv[c][d] = 0.f; r_app[c][d] = 0.f; PBRE_UNROLL for (int i = 0; i < QD; i++) { rJ[c][d][i] = 0.f; rB[c][d][i] = 0.f; } }
        }
        if (rt_on[0]) {
            float jl[30];
            const float* dl = dyn + (size_t)r * DEL * cs + env;
            PBRE_UNROLL for (int e = 0; e < 30; e++) jl[e] = dl[(size_t)(116 + e) * cs];
            PBRE_UNROLL for (int c = 0; c < NRT; c++) {
                if (!rt_on[c]) continue;
                const bool act = (c == 0 ? a0 : a1) != 0.f;
                r_mu[c] = act ? d0p[(size_t)(110 + c) * cs] : 0.f;
                PBRE_UNROLL for (int d = 0; d < 3; d++) {
                    PBRE_UNROLL for (int i = 0; i < QD; i++) rJ[c][d][i] = act ? jl[(c * 3 + d) * QD + i] : 0.f;      // (stale rows of an earlier step where there is no contact)
                    float b[QD];
                    PBRE_UNROLL for (int i = 0; i < QD; i++) b[i] = 0.f;
                    PBRE_UNROLL for (int col = 0; col < ND; col++) {
                        const float jc = qb(rJ[c][d][col % QD], col / QD);
                        PBRE_UNROLL for (int i = 0; i < QD; i++) b[i] = fmaf(A[i][col], jc, b[i]);
                    }
                    float den = 0.f;
                    PBRE_UNROLL for (int i = 0; i < QD; i++) { rB[c][d][i] = b[i]; den = fmaf(rJ[c][d][i], b[i], den); }
                    den += qb_x<0xB1>(den); den += qb_x<0x4E>(den);        // sum over the quad
                    r_dinv[c][d] = act ? 1.f / den : 0.f;
                }
                const float pen = d0p[(size_t)(112 + c) * cs] + P.slop;
                r_rhs[c] = act ? (pen > 0.f ? -pen * inv_dt : -pen * P.erp * inv_dt) * r_dinv[c][0] : 0.f;
            }
        }
    }
    auto rrow = [&](int c, int d) {
        float jv = 0.f;
        PBRE_UNROLL for (int i = 0; i < QD; i++) jv = fmaf(rJ[c][d][i], w[i], jv);
        jv += qb_x<0xB1>(jv); jv += qb_x<0x4E>(jv);
        float sn;
        if (d == 0) sn = __builtin_amdgcn_fmed3f(r_app[c][0] - fmaf(jv, r_dinv[c][0], -r_rhs[c]), 0.f, 1e10f);
        else {
            const float hi = r_mu[c] * r_app[c][0];
            sn = __builtin_amdgcn_fmed3f(r_app[c][d] - jv * r_dinv[c][d], -hi, hi);
            sn = hi > 0.f ? sn : r_app[c][d];
        }
        const float dd = sn - r_app[c][d]; r_app[c][d] = sn;
        PBRE_UNROLL for (int k = 0; k < QD; k++) w[k] = fmaf(dd, rB[c][d][k], w[k]);
    };
    const bool has_rt = rt_on[0];
    // ---- RC: the object (pbre_objstep.hpp: unconstrained twist, rows against the table) and the robot-object rows that couple it to the
    //      joints.  Row along dir at the box point op + rB: J = [J_robot ; -dir ; -(rB x dir)]; an impulse dd changes the joint
    //      velocities by dd B, the object's twist by (-dd / m dir, -dd I_w^-1 (rB x dir)).
    constexpr int NRO = LaneD::NRO;
    ObjStep ob;
    float oJ[RC ? NRO : 1][3][QD], oB[RC ? NRO : 1][3][QD], o_dir[RC ? NRO : 1][3][3], o_rxd[RC ? NRO : 1][3][3], o_g[RC ? NRO : 1][3][3];
    float o_dinv[RC ? NRO : 1][3], o_app[RC ? NRO : 1][3], o_rhs[RC ? NRO : 1], o_mu[RC ? NRO : 1];
    float pose[7], tw0[6];
    bool ro_on[RC ? NRO : 1];          // wave-uniform: some env of the wave uses the slot (rows of an env without it are exact no-ops)
    PBRE_UNROLL for (int c = 0; c < (RC ? NRO : 1); c++) ro_on[c] = false;
    const float inv_m = 1.f / P.obj_m;
    if (RC) {
        PBRE_UNROLL for (int k = 0; k < 7; k++) pose[k] = st[Shape32::LC + k];
        PBRE_UNROLL for (int k = 0; k < 6; k++) tw0[k] = st[W + Shape32::LC + k];
        ob.setup(P, pose, tw0, P.obj_m, P.obj_mu, P.kl);
        const float* d0p = dyn + env;
        const float* dl = dyn + (size_t)r * DEL * cs + env;
        PBRE_UNROLL for (int c = 0; c < NRO; c++) {
            const bool act = d0p[(size_t)(178 + c) * cs] != 0.f;
            ro_on[c] = __any((int)act) != 0;
            o_mu[c] = act ? d0p[(size_t)(180 + c) * cs] : 0.f;
            const float rbx = d0p[(size_t)(202 + c * 3) * cs], rby = d0p[(size_t)(203 + c * 3) * cs], rbz = d0p[(size_t)(204 + c * 3) * cs];
            PBRE_UNROLL for (int d = 0; d < 3; d++) {
                const float dx = act ? d0p[(size_t)(184 + (c * 3 + d) * 3) * cs] : 0.f, dy = act ? d0p[(size_t)(185 + (c * 3 + d) * 3) * cs] : 0.f,
                            dz = act ? d0p[(size_t)(186 + (c * 3 + d) * 3) * cs] : 0.f;
                o_dir[c][d][0] = dx; o_dir[c][d][1] = dy; o_dir[c][d][2] = dz;
                const float cx = rby * dz - rbz * dy, cy = rbz * dx - rbx * dz, cz = rbx * dy - rby * dx;      // rB x dir
                o_rxd[c][d][0] = cx; o_rxd[c][d][1] = cy; o_rxd[c][d][2] = cz;
                // I_w^-1 (rB x dir) = (m I_w^-1)(rB x dir) / m
                o_g[c][d][0] = (ob.Ii[0] * cx + ob.Ii[3] * cy + ob.Ii[4] * cz) * inv_m;
                o_g[c][d][1] = (ob.Ii[3] * cx + ob.Ii[1] * cy + ob.Ii[5] * cz) * inv_m;
                o_g[c][d][2] = (ob.Ii[4] * cx + ob.Ii[5] * cy + ob.Ii[2] * cz) * inv_m;
                PBRE_UNROLL for (int i = 0; i < QD; i++) oJ[c][d][i] = act ? dl[(size_t)(148 + (c * 3 + d) * QD + i) * cs] : 0.f;
                float b[QD];
                PBRE_UNROLL for (int i = 0; i < QD; i++) b[i] = 0.f;
                PBRE_UNROLL for (int col = 0; col < ND; col++) {
                    const float jc = qb(oJ[c][d][col % QD], col / QD);
                    PBRE_UNROLL for (int i = 0; i < QD; i++) b[i] = fmaf(A[i][col], jc, b[i]);
                }
                float den = 0.f;
                PBRE_UNROLL for (int i = 0; i < QD; i++) { oB[c][d][i] = b[i]; den = fmaf(oJ[c][d][i], b[i], den); }
                den += qb_x<0xB1>(den); den += qb_x<0x4E>(den);
                den += (dx * dx + dy * dy + dz * dz) * inv_m + (cx * o_g[c][d][0] + cy * o_g[c][d][1] + cz * o_g[c][d][2]);
                o_dinv[c][d] = act ? 1.f / den : 0.f;
                o_app[c][d] = 0.f;
            }
            const float pen = d0p[(size_t)(182 + c) * cs] + P.slop;
            o_rhs[c] = act ? (pen > 0.f ? -pen * inv_dt : -pen * P.erp * inv_dt) * o_dinv[c][0] : 0.f;
        }
    }
    auto orow = [&](int c, int d) {    // robot-object row (RC)
        float jv = fmaf(oJ[c][d][2], w[2], fmaf(oJ[c][d][1], w[1], oJ[c][d][0] * w[0])) + fmaf(oJ[c][d][4], w[4], oJ[c][d][3] * w[3]);      // (two short chains)
        jv += qb_x<0xB1>(jv); jv += qb_x<0x4E>(jv);
        jv -= fmaf(o_dir[c][d][2], ob.vz, fmaf(o_dir[c][d][1], ob.vy, o_dir[c][d][0] * ob.vx)) + fmaf(o_rxd[c][d][2], ob.wz, fmaf(o_rxd[c][d][1], ob.wy, o_rxd[c][d][0] * ob.wx));
        float sn;
        if (d == 0) sn = __builtin_amdgcn_fmed3f(o_app[c][0] - fmaf(jv, o_dinv[c][0], -o_rhs[c]), 0.f, 1e10f);
        else {
            const float hi = o_mu[c] * o_app[c][0];
            sn = __builtin_amdgcn_fmed3f(o_app[c][d] - jv * o_dinv[c][d], -hi, hi);
            sn = hi > 0.f ? sn : o_app[c][d];
        }
        const float dd = sn - o_app[c][d]; o_app[c][d] = sn;
        PBRE_UNROLL for (int k = 0; k < QD; k++) w[k] = fmaf(dd, oB[c][d][k], w[k]);
        const float dm = -dd * inv_m;
        ob.vx = fmaf(dm, o_dir[c][d][0], ob.vx); ob.vy = fmaf(dm, o_dir[c][d][1], ob.vy); ob.vz = fmaf(dm, o_dir[c][d][2], ob.vz);
        ob.wx = fmaf(-dd, o_g[c][d][0], ob.wx); ob.wy = fmaf(-dd, o_g[c][d][1], ob.wy); ob.wz = fmaf(-dd, o_g[c][d][2], ob.wz);
    };
    auto contacts = [&]() {            // Bullet: all normals (object-table, robot-object, robot-table), then all frictions
        if (RC) { ob.sweep_normals(); PBRE_UNROLL for (int c = 0; c < NRO; c++) if (ro_on[c]) orow(c, 0); }
        PBRE_UNROLL for (int c = 0; c < NRT; c++) if (rt_on[c]) rrow(c, 0);
        if (RC) { ob.sweep_frictions(); PBRE_UNROLL for (int c = 0; c < NRO; c++) if (ro_on[c]) { orow(c, 1); orow(c, 2); } }
        PBRE_UNROLL for (int c = 0; c < NRT; c++) if (rt_on[c]) { rrow(c, 1); rrow(c, 2); }
    };

    const float mlim = P.motor_imp, llim = P.limit_imp;
    auto axpy = [&](int j, float d) { PBRE_UNROLL for (int k = 0; k < QD; k++) w[k] = fmaf(d, A[k][j], w[k]); };
    float peak = 0.f;                  // clamp-free rows: the largest |applied impulse| a motor of this lane had at the end of any sweep
    auto motor_free = [&](int j) {     // clamp-free row; sabs: the applied impulse of the motor this lane owns
        const int o = j / QD, i0 = j % QD;
        const float d = qb(fmaf(-m_dinv[i0], w[i0], m_rhs[i0]), o);
        sabs[i0] = fmaf(d, own[o], sabs[i0]);
        axpy(j, d);
    };
    auto track = [&]() {               // (a motor's impulse changes once per sweep, in its own row: every value it takes is seen)
        PBRE_UNROLL for (int i = 0; i < QD; i++) peak = fmaxf(peak, fabsf(sabs[i]));
    };
    auto motor = [&](int j) {          // clamping row, delta form (sabs holds the applied impulse here)
        const int o = j / QD, i0 = j % QD;
        const float nt = fmaf(-m_dinv[i0], w[i0], m_rhs[i0]);
        const float d = qb(__builtin_amdgcn_fmed3f(nt, -mlim - sabs[i0], mlim - sabs[i0]), o);
        sabs[i0] = fmaf(d, own[o], sabs[i0]);
        axpy(j, d);
    };
    auto limit = [&](int j) {
        const int o = j / QD, i0 = j % QD;
        const float t = fmaf(m_dinv[i0] * l_dir[i0], w[i0], -l_rhs[i0]);
        const float s = __builtin_amdgcn_fmed3f(l_app[i0] - t, 0.f, llim);
        const float d = qb((s - l_app[i0]) * l_dir[i0], o);
        l_app[i0] = r == o ? s : l_app[i0];
        axpy(j, d);
    };
    auto solve = [&](auto&& mrow) {
        for (int it = 0; it < P.iters; it += 2) {
            PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) mrow(j);
            track();
            if (has_limit) { PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) if (lim_on[j]) limit(j); }
            if (RC || has_rt) contacts();
            if (it + 1 >= P.iters) break;
            if (has_limit) { PBRE_UNROLL for (int j = 0; j < ND; j++) if (lim_on[j]) limit(j); }
            PBRE_UNROLL for (int j = 0; j < ND; j++) mrow(j);
            track();
            if (RC || has_rt) contacts();
        }
    };
    // The clamp-free attempt is only worth making in a wave without contact rows: a hand pressed on the table or on the object is
    // exactly where a position motor runs into its impulse bound (the contact stops what the motor drives), the attempt then fails
    // and the solve runs twice -- in the stationary random-action mix that was most waves.  Either way an env's result is the same
    // (a clamping row whose clamp does not bind returns the free row's delta bit for bit).  PBRE_QUAD_FREE_ALWAYS=1: round 2's rule (A/B).
#ifndef PBRE_QUAD_FREE_ALWAYS
#define PBRE_QUAD_FREE_ALWAYS 0
#endif
    bool again = true;
    if (PBRE_QUAD_FREE_ALWAYS || !(RC || has_rt)) {
        solve(motor_free);
        const bool over = !(peak <= mlim);      // (a NaN fails the test as well)
        again = __any((int)over) != 0;
        if (again) {
            PBRE_UNROLL for (int i = 0; i < QD; i++) { w[i] = w0[i]; sabs[i] = 0.f; l_app[i] = 0.f; }
            PBRE_UNROLL for (int c = 0; c < NRT; c++) PBRE_UNROLL for (int d = 0; d < 3; d++) r_app[c][d] = 0.f;
            if (RC) {
                ob.setup(P, pose, tw0, P.obj_m, P.obj_mu, P.kl);
                PBRE_UNROLL for (int c = 0; c < NRO; c++) PBRE_UNROLL for (int d = 0; d < 3; d++) o_app[c][d] = 0.f;
            }
        }
    }
    if (again) solve(motor);
    // ---- integrate the joints (semi-implicit Euler)
    PBRE_UNROLL for (int i = 0; i < QD; i++) {
        const float v = fminf(fmaxf(w[i], -vmax), vmax);
        st[W + d0 + i] = v; st[d0 + i] = fmaf(dt, v, q[i]);
    }
    if (RC && r == 0) {      // the object's twist after the coupled solve, for kw_fin
        float o[6];
        ob.result(P, o);
        PBRE_UNROLL for (int k = 0; k < 6; k++) dyn[(size_t)(208 + k) * cs + env] = o[k];
    }
}

#ifndef PBRE_QUAD_WAVES
#define PBRE_QUAD_WAVES 2            // waves per SIMD kw_quad is register-limited to (A/B on MI355X, 65536 envs: 3 -> 100, 2 -> 115 M env-steps/s; no spills at 2)
#endif
// Simple envs: every env of the batch in natural order, the quads of complex envs idle.
__global__ __launch_bounds__(LTPB, PBRE_QUAD_WAVES) void kw_quad(const TablesT<Shape32>* __restrict__ T, const Params P, float* __restrict__ state,
                                                   const float* __restrict__ actions, int n, int act_dim, int MODE, const float* __restrict__ tgt,
                                                   const signed char* __restrict__ cls_cur, float* __restrict__ dyn, size_t cs,
                                                   const unsigned long long* __restrict__ ik_done, int ik_seq) {
    const int gl = blockIdx.x * LTPB + threadIdx.x;
    const int env = gl >> 2;
    if (env >= n || cls_cur[env] != 0) return;             // (whole quads)
    quad_step<false>(T, P, state, actions, act_dim, MODE, tgt, dyn, cs, env, gl & 3, ik_done, ik_seq);
}
// Complex envs (robot-object contact) over the compacted list, 16 per wave; a whole SIMD's register file per wave.  Persistent blocks
// (the host does not know the list's length).
__global__ __launch_bounds__(LTPB, 1) void kw_quad_rc(const TablesT<Shape32>* __restrict__ T, const Params P, float* __restrict__ state,
                                                      const float* __restrict__ actions, int act_dim, int MODE, const float* __restrict__ tgt,
                                                      const int* __restrict__ cur_list, const int* __restrict__ cur_count, float* __restrict__ dyn, size_t cs,
                                                      const unsigned long long* __restrict__ ik_done, int ik_seq, int epw) {
    // epw envs per wave (<= 16; PBRE_QUAD_RC_EPW).  Which limit rows and contact slots a wave sweeps is the UNION over its envs (rows of an env
    // without them are exact no-ops), and these few lone waves are the step's critical path in both control modes -- but fewer envs per wave do
    // not shorten them (measured, profiles/r05w_quad_rc_epw.txt: 16 / 4 / 1 envs per wave: 378 / 397 / 391 us under joint control): the
    // kernel's duration is ONE env's coupled chain on four lanes.  What an env computes does not depend on epw.
    const int total = *cur_count;
    if ((int)(threadIdx.x >> 2) >= epw) return;           // (whole quads)
    for (int base = blockIdx.x * epw; base < total; base += gridDim.x * epw) {
        const int i = base + (int)(threadIdx.x >> 2);
        if (i >= total) break;                             // (whole quads)
        quad_step<true>(T, P, state, actions, act_dim, MODE, tgt, dyn, cs, cur_list[i], (int)(threadIdx.x & 3), ik_done, ik_seq);
    }
}

// The end of an env's step: object pose from the twist kw_obj left in the side record (simple envs) / kw_quad_rc left in the dyn buffer
// (complex envs), then Lane::finish.
__global__ __launch_bounds__(LTPB) void kw_fin(const TablesT<Shape32>* __restrict__ T, const Params P, float* __restrict__ state, float* __restrict__ out,
                                               int n, int ow, int flags, int MODE, const float* __restrict__ objv, const float* __restrict__ dyn, size_t cs,
                                               const signed char* __restrict__ cls_cur, signed char* __restrict__ cls, int* __restrict__ next_list,
                                               int* __restrict__ next_count, int* __restrict__ zero_count) {
    constexpr int ND = LaneD::ND, W = Shape32::W, LC = Shape32::LC, XO = 2 * Shape32::W;
    const int env = blockIdx.x * LTPB + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *zero_count = 0;
    if (env >= n) return;
    float* st = state + (size_t)env * Shape32::STATE;
    float q[ND], qd[ND];
    PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = st[j]; qd[j] = st[W + j]; }
    LaneD::V3 op; op.x = st[LC]; op.y = st[LC + 1]; op.z = st[LC + 2];
    LaneD::Q4 oq; oq.x = st[LC + 3]; oq.y = st[LC + 4]; oq.z = st[LC + 5]; oq.w = st[LC + 6];
    if (!(flags & 1) && st[XO + 14] == 0.f) {
        const float dt = P.dt;
        float o[6];
        if (cls_cur[env] != 0) { PBRE_UNROLL for (int k = 0; k < 6; k++) o[k] = dyn[(size_t)(208 + k) * cs + env]; }
        else { PBRE_UNROLL for (int k = 0; k < 6; k++) o[k] = objv[(size_t)env * W + LC + k]; }
        LaneD::V3 ov, ow_; ov.x = o[0]; ov.y = o[1]; ov.z = o[2]; ow_.x = o[3]; ow_.y = o[4]; ow_.z = o[5];
        op.x = fmaf(dt, ov.x, op.x); op.y = fmaf(dt, ov.y, op.y); op.z = fmaf(dt, ov.z, op.z);
        float ang = sqrtf(fmaf(ow_.x, ow_.x, fmaf(ow_.y, ow_.y, ow_.z * ow_.z)));
        if (ang * dt > 0.78539816339744831f) ang = 0.78539816339744831f * P.inv_dt;
        const float sc_ = ang < 0.001f ? 0.5f * dt - dt * dt * dt * 0.020833333333f * ang * ang : sinf(0.5f * ang * dt) / ang;
        LaneD::Q4 dq; dq.x = ow_.x * sc_; dq.y = ow_.y * sc_; dq.z = ow_.z * sc_; dq.w = cosf(ang * dt * 0.5f);
        const LaneD::Q4 nq = LaneD::FX::qmul(dq, oq);
        const float in = 1.f / sqrtf(nq.x*nq.x + nq.y*nq.y + nq.z*nq.z + nq.w*nq.w);
        oq.x = nq.x * in; oq.y = nq.y * in; oq.z = nq.z * in; oq.w = nq.w * in;
        st[LC] = op.x; st[LC + 1] = op.y; st[LC + 2] = op.z; st[LC + 3] = oq.x; st[LC + 4] = oq.y; st[LC + 5] = oq.z; st[LC + 6] = oq.w;
        PBRE_UNROLL for (int k = 0; k < 6; k++) st[W + LC + k] = o[k];
    }
    {   // NaN / Inf guard: what kw_dyn found in the incoming state
        const float fin_in = dyn[(size_t)214 * cs + env];
        if (!(fin_in == 0.f)) { q[0] += fin_in; st[0] = q[0]; }
    }
    const int c = LaneD::finish(*T, P, st, q, qd, op, oq, (MODE & LaneD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags, P.env_id_base + (unsigned long long)env);
    wpublish(env, c, cls, next_list, next_count);
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
    float* dyn = nullptr;             // side buffer of the quad pipeline
    hipStream_t side = nullptr;       // kw_obj, the IK kernel and kw_quad_rc run beside kw_dyn / kw_quad; picked per caller stream (pick_side)
    SidePick sp;                      // candidates + calibration (pbre_sidepick.hpp)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_dyn = nullptr;
    // Cartesian control, round 5: the IK kernel on a stream of its own (highest priority) and per-env "targets complete" marks (ik_done[env] ==
    // ik_seq), so that kw_quad / kw_quad_rc start right behind kw_dyn and only the quads of envs whose IK is still iterating wait
    hipStream_t ik_stream = nullptr;
    unsigned long long* ik_done = nullptr;      // [n][ND] the hand-over boxes: (target | sequence number << 32) per DoF
    int ik_seq = 0;
    int ik_overlap = 1;               // PBRE_IK_OVERLAP=0: the kernel-level dependency of rounds 2-4 (A/B)
    int rc_epw = 16;                  // envs per wave of kw_quad_rc (PBRE_QUAD_RC_EPW: A/B, measured neutral)
    size_t dyn_cs = 0;
    int n_simd = 1024;
    ~WideLane() override {
        for (void* p : {(void*)cls, (void*)list, (void*)count, (void*)dyn, (void*)ik_done}) if (p) (void)hipFree(p);
        if (ik_stream) (void)hipStreamDestroy(ik_stream);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_dyn) (void)hipEventDestroy(ev_dyn);
        if (ev_ik) (void)hipEventDestroy(ev_ik);
        sp.destroy();
    }
    // (round objects -- pbre_physics.obj_shape -- included: the object's own rows are ObjStep's, the robot-object test is Fast::sphere_obj)
    // (not with Bullet's residual exit, pbre_physics.solver_residual_threshold > 0: the pipeline splits an env's rows over kernels, the
    // test is a maximum over all of them -- such a batch is stepped by the lane-group kernel, Core::step<RT>)
    // (nor with a convex-hull object: the pipeline's narrow phase is compiled for the primitives)
    bool lane_ok() const override { return enabled && topo_ok && cls != nullptr && objv != nullptr && !(P.res_lim > 0.f) && P.obj_shape != PBRE_SHAPE_HULL; }
    void lane_invalidate() override { cls_valid = false; }
    hipError_t lane_alloc() override {
        const char* knob = getenv("PBRE_ICUB_LANE");
        // Default: the pipeline from 16384 envs on.  Below that its chain of lone-wave kernels (kw_dyn -> kw_quad -> kw_fin) on a
        // mostly idle GPU (whose clocks follow the load) is slower than the one lane-group kernel: stationary random-action mix, ms per
        // step pipeline / lane-group: 8192 envs 1.41 / 1.32 (IK), 0.80 / 0.66 (joint); 16384 envs 1.51 / 1.92, 0.86 / 0.98; 32768 envs
        // 1.3 / 3.2, 0.73 / 1.6.  PBRE_ICUB_LANE=1 / 0 forces the pipeline / the lane-group kernel (tests, A/B).
        enabled = knob ? knob[0] != '0' : n >= 16384;
        topo_ok = lane_topo_matches<TopoICub, Shape32>(T);
        if (!enabled || !topo_ok) return hipSuccess;
        hipError_t e;
        if ((e = hipMalloc(&cls, 2 * (size_t)n)) != hipSuccess) return e;
        if ((e = hipMalloc(&list, 2 * (size_t)n * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMalloc(&count, 3 * sizeof(int))) != hipSuccess) return e;
        if (!(getenv("PBRE_ICUB_SIDE") && getenv("PBRE_ICUB_SIDE")[0] == '0')) {      // PBRE_ICUB_SIDE=0: everything in stream order (A/B)
            // highest priority: the complex envs' few waves need a whole SIMD's registers each -- they have to be placed before
            // kw_quad's waves fill every SIMD, or they would run after it
            int plo = 0, phi = 0;
            (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
            if ((e = sp.create(phi, true)) != hipSuccess) return e;
            side = sp.side;
            const char* ef = getenv("PBRE_EVENT_FENCE");      // (see pbre_capi.hip: device-only dependencies need no system-scope fence)
            const unsigned efl = hipEventDisableTiming | ((ef && ef[0] == '1') ? 0u : (unsigned)hipEventDisableSystemFence);
            if ((e = hipEventCreateWithFlags(&ev_fork, efl)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev_join, efl)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev_dyn, efl)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&ev_ik, efl)) != hipSuccess) return e;
            if (const char* ev = getenv("PBRE_IK_OVERLAP")) ik_overlap = atoi(ev);
            if (const char* ev = getenv("PBRE_QUAD_RC_EPW")) rc_epw = std::min(16, std::max(1, atoi(ev)));
            if (ik_overlap) {
                if ((e = hipStreamCreateWithPriority(&ik_stream, hipStreamNonBlocking, phi)) != hipSuccess) return e;
                if ((e = hipMalloc(&ik_done, (size_t)n * LaneD::ND * sizeof(unsigned long long))) != hipSuccess) return e;
                if ((e = hipMemset(ik_done, 0, (size_t)n * LaneD::ND * sizeof(unsigned long long))) != hipSuccess) return e;
            }
        }
        {
            // element stride: a multiple of 16 floats (kw_quad's 64-byte segments stay aligned), but never a power of two -- with 32768 envs
            // consecutive elements were 128 KB apart and every store / load of a wave fell on the same few HBM channels; which channels
            // depended on where the allocation landed: the same command ran at 0.35 or at 0.6 ms per step from one process to the next
            const size_t npad = ((size_t)n + 15) / 16 * 16 + 1040;
            dyn_cs = npad;
            if ((e = hipMalloc(&dyn, (size_t)4 * DEL * dyn_cs * sizeof(float))) != hipSuccess) return e;
            if ((e = hipMemset(dyn, 0, (size_t)4 * DEL * dyn_cs * sizeof(float))) != hipSuccess) return e;
        }
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, device) == hipSuccess) n_simd = pr.multiProcessorCount * 4;
        return hipSuccess;
    }
    long n_lane_steps = 0;
    bool ik_pending = false;          // Cartesian control: the IK kernel of this step is launched by lane_t (beside kw_dyn, which does not need the targets)
    hipEvent_t ev_ik = nullptr;
    void launch_lane_ik(const float* act, hipStream_t s) override {
        if (side) { ik_pending = true; return; }
        hipLaunchKernelGGL(kw_lane_ik, dim3((n + LTPB - 1) / LTPB), dim3(LTPB), 0, s, dT, P, state, act, tgt, n, act_dim, (unsigned long long*)nullptr, 0);
    }
    void lane_t(int MODE, const float* act, float* out, int flags, hipStream_t s, hipEvent_t* ek) {
        signed char* c_cur = cls + (size_t)cur * n; signed char* c_nxt = cls + (size_t)(cur ^ 1) * n;
        int* l_cur = list + (size_t)cur * n; int* l_nxt = list + (size_t)(cur ^ 1) * n;
        int* k_cur = count + ccur; int* k_nxt = count + (ccur + 1) % 3; int* k_zero = count + (ccur + 2) % 3;
        const int be = (n + LTPB - 1) / LTPB;
        if (side) side = sp.pick(s);
        // PBRE_ICUB_TRACE=<k>: HIP events around every kernel of the k-th and the following two steps, durations and start offsets on stderr
        static const int trace_at = getenv("PBRE_ICUB_TRACE") ? atoi(getenv("PBRE_ICUB_TRACE")) : -1;
        const bool tr = trace_at >= 0 && n_lane_steps >= trace_at && n_lane_steps < trace_at + 3;
        n_lane_steps++;
        hipEvent_t te[14] = {};
        if (tr) for (auto& e : te) (void)hipEventCreate(&e);
        auto mark = [&](int k, hipStream_t st) { if (tr) (void)hipEventRecord(te[k], st); };
        {
            // side stream: the object solve of the simple envs (45 us of latency), the IK kernel (Cartesian control; kw_dyn does not need
            // the targets), then -- once kw_dyn is through -- the complex envs' coupled solve (a few lone waves) beside kw_quad; kw_fin
            // needs all of it
            hipStream_t s2 = side ? side : s;
            mark(0, s);
            if (side) { (void)hipEventRecord(ev_fork, s); (void)hipStreamWaitEvent(side, ev_fork, 0); }
            mark(1, s2);
            // (Cartesian control: the IK kernel is the head of the step's critical path -- IK -> kw_quad_rc -> kw_fin -- so it goes first
            // on the side stream and the object solve moves to the caller's stream behind kw_quad, which is done before kw_quad_rc on the
            // side stream anyway (measured: beside the IK kernel -- right behind kw_dyn -- it stretched that kernel from 0.41 to 0.66 ms);
            // joint control: the object solve stays on the side stream, beside kw_dyn)
            const bool ik_side = ik_pending;
            // Round 5 (profiles/r05s_icub_timeline.txt: IK 1..399 us, kw_quad / kw_quad_rc 409..818, kw_fin ..887): the solve kernels waited for
            // the WHOLE IK kernel, whose duration is that of the 0.09 % of the envs that iterate to the cap.  With per-env "targets complete"
            // marks (kw_lane_ik publishes an env's targets in the iteration in which it converges; quad_step waits for its own env's mark
            // right before it reads the targets) the IK kernel runs on a stream of its own beside kw_dyn AND the solve kernels.  Only while the
            // IK kernel's, kw_dyn's and kw_obj's waves are all resident at once (3 be waves on 2 n_simd slots): the IK waves must be running
            // before quads start to wait for them; larger batches keep the kernel-level dependency.
            const bool ik_ovl = ik_side && side != nullptr && ik_stream != nullptr && ik_done != nullptr && 3 * be <= 2 * n_simd;
            const bool obj_main = ik_side && side != nullptr;      // (also with the per-env hand-over: kw_obj beside the IK kernel AND kw_dyn stretched all three -- dyn 92 -> 221 us)
            if (!(flags & 1) && !obj_main) hipLaunchKernelGGL((kw_obj<Shape32>), dim3(be), dim3(64), 0, s2, P, state, objv, n);
            mark(2, s2);
            if (ik_ovl) {
                if (ik_seq == 0x7fffffff) { (void)hipMemsetAsync(ik_done, 0, (size_t)n * LaneD::ND * sizeof(unsigned long long), s); ik_seq = 0; (void)hipEventRecord(ev_fork, s); }
                ik_seq++;
                (void)hipStreamWaitEvent(ik_stream, ev_fork, 0);
                hipLaunchKernelGGL(kw_lane_ik, dim3(be), dim3(LTPB), 0, ik_stream, dT, P, state, act, tgt, n, act_dim, ik_done, ik_seq);
                (void)hipEventRecord(ev_ik, ik_stream);
                ik_pending = false;
            } else if (ik_side) {      // (kw_quad_rc must be the next thing in this stream when the targets are ready, see lane_alloc)
                hipLaunchKernelGGL(kw_lane_ik, dim3(be), dim3(LTPB), 0, side, dT, P, state, act, tgt, n, act_dim, (unsigned long long*)nullptr, 0);
                (void)hipEventRecord(ev_ik, side);
                ik_pending = false;
            }
            const unsigned long long* ikd = ik_ovl ? ik_done : nullptr;
            mark(3, s2);
            if (ek) (void)hipEventRecord(ek[0], s);
            mark(4, s);
            hipLaunchKernelGGL(kw_dyn, dim3(be), dim3(LTPB), 0, s, dT, P, state, n, flags, c_cur, dyn, dyn_cs);
            mark(5, s);
            if (side) { (void)hipEventRecord(ev_dyn, s); (void)hipStreamWaitEvent(side, ev_dyn, 0); }
            mark(6, s2);
            hipLaunchKernelGGL(kw_quad_rc, dim3(std::min((n + rc_epw - 1) / rc_epw, 256)), dim3(LTPB), 0, s2, dT, P, state, act, act_dim, MODE, tgt, l_cur, k_cur, dyn, dyn_cs, ikd, ik_seq, rc_epw);
            mark(7, s2);
            if (side) (void)hipEventRecord(ev_join, side);
            if (ik_side && !ik_ovl) (void)hipStreamWaitEvent(s, ev_ik, 0);
            mark(8, s);
            hipLaunchKernelGGL(kw_quad, dim3((4 * n + LTPB - 1) / LTPB), dim3(LTPB), 0, s, dT, P, state, act, n, act_dim, MODE, tgt, c_cur, dyn, dyn_cs, ikd, ik_seq);
            mark(9, s);
            if (ek) (void)hipEventRecord(ek[1], s);
            if (!(flags & 1) && obj_main) hipLaunchKernelGGL((kw_obj<Shape32>), dim3(be), dim3(64), 0, s, P, state, objv, n);
            if (side) (void)hipStreamWaitEvent(s, ev_join, 0);
            if (ik_ovl) (void)hipStreamWaitEvent(s, ev_ik, 0);      // (the step is complete when every kernel of it is: the IK kernel's last waves too)
            mark(10, s);
            hipLaunchKernelGGL(kw_fin, dim3(be), dim3(LTPB), 0, s, dT, P, state, out, n, ow, flags, MODE, objv, dyn, dyn_cs, c_cur, c_nxt, l_nxt, k_nxt, k_zero);
            mark(11, s);
        }
        if (tr) {
            (void)hipEventSynchronize(te[11]);
            auto el = [&](int a, int b) { float ms = 0.f; (void)hipEventElapsedTime(&ms, te[a], te[b]); return ms * 1e3f; };
            fprintf(stderr, "[pbre] iCub step %ld (us): obj %.0f (+%.0f)  ik %.0f (+%.0f)  dyn %.0f (+%.0f)  quad_rc %.0f (+%.0f)  quad %.0f (+%.0f)  fin %.0f (+%.0f)  total %.0f\n",
                    n_lane_steps - 1, el(1, 2), el(0, 1), el(2, 3), el(0, 2), el(4, 5), el(0, 4), el(6, 7), el(0, 6), el(8, 9), el(0, 8), el(10, 11), el(0, 10), el(0, 11));
            for (auto& e : te) (void)hipEventDestroy(e);
        }
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
            case K_SETTLE: lane_t(0, act, out, flags, s, ek); break;
            case K_SETTLE_TGT: lane_t(LaneD::M_TGT, act, out, flags, s, ek); break;
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
        *vg = hipFuncGetAttributes(&fa, (const void*)kw_quad) == hipSuccess ? fa.numRegs : -1;
        *complex_now = 0;
        if (!cls_valid && lane_ok()) {        // (e.g. right after a reset done by the lane-group kernel: classify the batch as the next lane step would)
            cur = 0; ccur = 0;
            (void)hipMemsetAsync(count, 0, 3 * sizeof(int), stream);
            hipLaunchKernelGGL(kw_lane_classify, dim3((n + LTPB - 1) / LTPB), dim3(LTPB), 0, stream, dT, P, state, n, cfg.flags & PBRE_F_NO_OBJECT, cls, list, count);
            cls_valid = hipGetLastError() == hipSuccess;
        }
        if (cls_valid) { (void)hipDeviceSynchronize(); (void)hipMemcpy(complex_now, count + ccur, sizeof(int), hipMemcpyDeviceToHost); }
        return 1;
    }
};

WideEngine* make_lane_engine() { return new WideLane(); }

}  // namespace pbre
