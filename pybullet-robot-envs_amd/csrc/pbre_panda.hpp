// pbre_panda.hpp -- the Panda engine's step kernels (templates), its context structs and the step launcher launch_step_t<MODE, RT>.
// Round 6: split out of pbre_capi.hip so that the twelve (MODE, RT) instantiations of the launcher -- each of which instantiates every
// step kernel -- compile as separate translation units in parallel (pbre_step_inst.hip; build.sh): the single TU took 9-12 minutes.
// pbre_capi.hip (the C-ABI and the small non-template kernels) declares them `extern template`; -DPBRE_UNITY puts the instantiations back
// into pbre_capi.hip (tools/build_variant.sh: A/B and phase-probe builds of one object file).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define PBRE_HD __device__ __forceinline__
#define PBRE_UNROLL _Pragma("unroll")
#define PBRE_ANY(x) (__any((int)(x)) != 0)
#define PBRE_REG_BARRIER() asm volatile("" ::: "memory")
#define PBRE_LAUNDER(p) asm volatile("" : "+s"(p))
#define PBRE_PAIR_G_FLAGS (55 * 64 * 4 + 64)       /* byte offset of the 64 sequence words in a tail pair's global record: behind a whole PairX */
#define PBRE_PAIR_G_BYTES (PBRE_PAIR_G_FLAGS + 64 * 4)
// (Fast::finish, robot wave of a pair: the object wave's seven values in px->o.  px->g == nullptr: the object wave is a sibling wave of the block --
// block barrier.  Else it is a wave of another block (k_fused's tail pairs) that writes the seven values into the pair's GLOBAL record -- a PairX
// both waves use (px == px->g there: the k_fused grid's 64-thread blocks keep no LDS, measured: 13.8 KB of static LDS per block cost every k_fast
// wave of the grid 9 %) -- and then, per lane, the launch's sequence number behind it (release): the lane waits for ITS word (acquire), bounded
// like PBRE_OBJV_SYNC -- the object wave waits for nothing and its block is dispatched before the robot wave's; a wait that runs out hands NaNs over: the NaN / Inf guard returns the env-step
// as done = 1, counts it once and restarts the env.)
#define PBRE_PAIR_SYNC(px, ln) do { if ((px)->g == nullptr) __syncthreads(); else { \
        const int* f_ = (const int*)((const char*)((px)->g) + PBRE_PAIR_G_FLAGS) + (ln); const int want_ = (px)->seq; int spins_ = 0; bool ok_ = true; \
        while (__hip_atomic_load(f_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want_) { \
            __builtin_amdgcn_s_sleep(8); \
            if (++spins_ > (1 << 22)) { ok_ = false; break; } } \
        PBRE_UNROLL for (int k_ = 0; k_ < 7; k_++) (px)->o[k_][ln] = ok_ ? (px)->g[k_ * 64 + (ln)] : __builtin_nanf(""); } } while (0)
#define PBRE_COUNT_BAD(p) atomicAdd((p), 1)
// (Fast::sweep<3>: an int summed over the 16 lanes of a row -- the DPP butterfly of DevLanes::sum on integers)
static __device__ __forceinline__ int pbre_row_sum_i(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);     // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);     // row_mirror
    return x;
}
#define PBRE_ROW_SUM_I(x) pbre_row_sum_i(x)
#ifndef PBRE_ROW_FINISH16      // 1: Fast::finish of a row wave's env on all 16 lanes of its group (the collision spheres' tests of the new state's class
#define PBRE_ROW_FINISH16 1    // one per lane, in one pass); 0: on lane 0 alone, as until round 6 (A/B)
#endif
// (Core::step, where `objv` and `P` are in scope: the side record is complete behind the block barrier -- its producer is a sibling wave of the
// block, k_row_list -- or, P.objv_seq != 0, once its first word carries this launch's sequence number: the producer is a wave of another block,
// k_fused's 64-thread grid.  P.objv_seq is uniform, so the barrier is not in divergent code.)
// The wait is bounded (~1 s): the producer waits for nothing and is dispatched first, so the bound is never reached -- but a wait that could
// not end would hang the device, so a row that does reach it POISONS its env-step (`poison` becomes NaN, Core::step makes the new velocities NaN):
// the NaN / Inf guard returns it as done = 1 / reward 0, restarts the env under PBRE_F_AUTO_RESET and counts it ONCE (pbre_kernel_info[12];
// every waiting lane used to count, 16 per env, and the env went on with a stale record).
#define PBRE_OBJV_SYNC(poison) do { if (P.objv_seq == 0) __syncthreads(); else { const int* f_ = (const int*)objv; int spins_ = 0; \
        while (__hip_atomic_load(f_, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != P.objv_seq) { \
            __builtin_amdgcn_s_sleep(8); \
            if (++spins_ > (1 << 22)) { (poison) = __builtin_nanf(""); break; } } } } while (0)
#ifndef PBRE_CONST_AS        // (-DPBRE_CONST_AS= builds the A/B variant with the model constants re-read through a plain pointer)
#define PBRE_CONST_AS __attribute__((address_space(4)))
#endif
#ifdef PBRE_PHASE_PROBE      // tools/phase_probe.py: cycles per phase of lane 0 of block 0's waves, summed over the launches since the last reset of the counters
__device__ unsigned long long g_probe[64];
#define PBRE_PROBE_DECL unsigned long long pb_t_ = __builtin_readcyclecounter();
#define PBRE_PROBE(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_probe[k], t_ - pb_t_); pb_t_ = t_; } while (0)
// (path k of a row wave: its count in slot 32 + k, the ticks of its sweeps in slot 48 + k)
#define PBRE_PROBE_PATH(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { atomicAdd(&g_probe[32 + (k)], 1ull); atomicAdd(&g_probe[48 + (k)], t_ - pb_t_); } } while (0)
#endif
#ifdef PBRE_WAVE_TRACE       // tools/wave_trace.py (variant build of ONE step unit): every row wave records the ticks from the start of Core::step to the end
// of its sweeps and what its sweeps were made of -- bits 0..15: solver paths taken (11 = started over with clamping motor rows, 12..15 see
// PBRE_PROBE_PATH in pbre_core.hpp), 16..19 robot-object slots, 20..23 object-table slots, 24..25 robot-table slots, 30 a joint-limit row -- to find the step's longest wave
__device__ unsigned long long g_wtrace[16384][3];      // ticks, bits, address of the wave's (first group's) state record
__device__ unsigned int g_wtrace_n;
// (diagnosis of what a k_fast wave costs the row wave it shares a SIMD with, DESIGN 5.3: pbre_debug_wave_diag(mode) makes every k_fast wave of k_fused<., false>
// hold its slot for ~90 us WITHOUT stepping its envs -- 1: asleep (no vector instruction, no code streamed), 2: a dependent v_fma chain in a 16-instruction loop (the vector
// unit as busy as a latency-bound wave keeps it, no instruction-cache footprint), 3: the same chain as 128 KB of straight-line code, 4: eight independent chains, 5: state-record streaming, 9: 3 and 5 in turn; 0: the step as it is.  Trace builds only: the rows of the simple envs are garbage then.)
__device__ int g_wave_diag;
#define PBRE_PROBE_DECL unsigned wt_bits_ = 0u; unsigned long long wt_t0_ = __builtin_readcyclecounter(); (void)wt_bits_; (void)wt_t0_;
#define PBRE_PROBE_PATH(k) (wt_bits_ |= 1u << (k))
#define PBRE_TRACE_ROWS(ob, hl) (wt_bits_ |= ((((unsigned)(ob) >> 4) & 15u) << 16) | (((unsigned)(ob) & 15u) << 20) | ((((unsigned)(ob) >> 8) & 3u) << 24) | ((hl) ? 1u << 30 : 0u))
#define PBRE_PROBE(k) do { if ((k) == 9 && (threadIdx.x & 63) == 0) { const unsigned i_ = atomicAdd(&g_wtrace_n, 1u); \
        if (i_ < 16384u) { g_wtrace[i_][0] = __builtin_readcyclecounter() - wt_t0_; g_wtrace[i_][1] = wt_bits_; g_wtrace[i_][2] = (unsigned long long)st; } } } while (0)
#endif
#include "pbre_host.hpp"
#include "lanes_device.hpp"
#include "pbre_core.hpp"
#include "pbre_fast.hpp"
#include "pbre_wide.hpp"
#include "pbre_sidepick.hpp"

using namespace pbre;
using CoreD = Core<DevLanes>;
using FastD = Fast<TopoPanda>;

constexpr int EPB = 16;              // envs per block of the row kernel
constexpr int TPB = EPB * W;         // 256 threads
constexpr int FTPB = 64;             // lane-per-env kernels: one wave per block
#ifndef PBRE_FAST_WAVES
#define PBRE_FAST_WAVES 2            // default waves per SIMD k_fast is register-limited to (round-1 A/B on MI355X: 1 -> 404, 2 -> 495, 3 -> 325 M env-steps/s;
                                     // since round 3 launch_step also instantiates <MODE, 3> and picks per step)
#endif
#ifndef PBRE_RC_PRIO
#define PBRE_RC_PRIO 3               // wave priority (s_setprio) of the complex-env kernels, 0: leave it (A/B)
#endif
constexpr int MODE_STEP = CoreD::M_ACTION | CoreD::M_OBS | CoreD::M_TASK;
constexpr int MODE_STEP_IK = CoreD::M_TGT | CoreD::M_OBS | CoreD::M_TASK;      // use_IK = 1: targets from k_ik
constexpr int MODE_SETTLE_IK = CoreD::M_TGT;
// action_repeat > 1: the non-final iterations of the apply_action loop simulate, test termination and count, without outputs
constexpr int MODE_INNER = CoreD::M_ACTION | CoreD::M_TASK | CoreD::M_INNER;
constexpr int MODE_INNER_IK = CoreD::M_TGT | CoreD::M_TASK | CoreD::M_INNER;
static_assert((int)CoreD::M_INNER == (int)FastD::M_INNER && (int)CoreD::M_TGT == (int)FastD::M_TGT, "mode bits shared by the row and lane kernels");

// ------------------------------------------------------------------ kernels
// General row kernel.  n = real env count; state has ceil16(n) + 16 records (the last 16 are valid dummy records for the
// padding rows of a partially filled block); actions/out rows of padding rows are redirected to env 0 / a scratch row.
// RT (all step kernels): pbre_physics.solver_residual_threshold > 0 -- the variants with Bullet's exit test of the sweep loop
template <int MODE, bool RT = false>
__global__ __launch_bounds__(TPB) void k_step(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                              const float* __restrict__ actions, float* __restrict__ out,
                                              float* __restrict__ scratch_row, int n, int dummy_base, int act_dim, int ow, int flags,
                                              const float* __restrict__ tgt) {
    const int row = threadIdx.x >> 4;
    const int i = blockIdx.x * EPB + row;
    const bool real = i < n;
    const int env = real ? i : dummy_base + row;
    float* st = state + (size_t)env * STATE;
    const float* a = nullptr;
    float* o = nullptr;
    if (MODE & CoreD::M_ACTION) a = actions + (size_t)(real ? env : 0) * act_dim;
    if (MODE & CoreD::M_OBS) o = real ? out + (size_t)env * ow : scratch_row;
    // (padding rows: pristine dummy record in, scratch record out -- see k_row_list)
    CoreD::step<RT>(*T, P, st, a, o, MODE, flags, (MODE & CoreD::M_TGT) ? tgt + (size_t)env * NJ : nullptr, P.env_id_base + (unsigned long long)env, nullptr,
                    real ? nullptr : st + (size_t)EPB * STATE, (RT && real && P.sweeps) ? P.sweeps + env : nullptr);
}

// Complex envs are kept in NB = NCLASS - 1 bucket lists (one per class, see Fast::classify) so that the waves of
// k_fast_rc are homogeneous.  lists: [NB][cap] ints, counts: [NB] ints.
constexpr int NB = FastD::NCLASS - 1;
// (c: what Fast::step / finish returned -- the class, with BAD_BIT when the NaN / Inf guard fired: counted here, pbre_kernel_info[12])
__device__ __forceinline__ void publish_class(int env, int c, signed char* __restrict__ cls, int* __restrict__ next_list, int* __restrict__ next_count, int cap,
                                              int* __restrict__ bad_count = nullptr) {
    if (c & FastD::BAD_BIT) { if (bad_count) atomicAdd(bad_count, 1); c &= FastD::BAD_BIT - 1; }
    cls[env] = (signed char)c;
    if (c) next_list[(size_t)(c - 1) * cap + atomicAdd(next_count + (c - 1), 1)] = env;
}

// Simple envs: every env of the batch in natural order, lanes of complex envs idle.  Block = one wave.
// (Staging the wave's 64 output rows through LDS and streaming them out as one contiguous block with coalesced 256-byte stores was
// measured too -- profiles/r02_pmc_hbm.json: WRITE_SIZE 62.1 MB against 63.3 MB with each lane writing its own 140-byte row, and the
// same step time -- the L2 already merges the lanes' 4-byte stores into full lines; what WRITE_SIZE carries beyond the records and
// rows is the register spill traffic.  The direct per-lane row writes stayed.)
// WPS: the waves per SIMD the register allocation is limited to.  2 (256 VGPRs) is the fast one; 3 (168 VGPRs, the setup phase
// spills ~0.8 KB per lane) leaves room for the complex envs' waves beside a batch that fills every wave slot of the 2-wave variant
// (131072 envs = 2048 waves = 1024 SIMDs x 2): there the displaced k_fast waves of the 2-wave variant run in a second round
// (0.249 ms per stationary step against 0.220 ms with this variant; right after reset(), without complex envs, 0.157 against 0.187).
// launch_step picks per step.
// (fast_wave: one wave's work -- the 64 envs of `chunk`, lane ln; k_fast's body and, round 5, the simple envs' part of k_fused)
// CT: read the model constants through the constant address space (see Fast::step; measured first in k_fused -- 131072 envs fresh 0.101 ->
// 0.089 ms -- then made the default of every lane-per-env step kernel; false: A/B)
template <int MODE, bool RT, bool CT = true>
__device__ __forceinline__ void fast_wave(const Tables* __restrict__ T, const Params& P, float* __restrict__ state,
                                          const float* __restrict__ actions, float* __restrict__ out, int n, int act_dim, int ow, int flags,
                                          const signed char* __restrict__ cls_cur, signed char* __restrict__ cls, int* __restrict__ next_list,
                                          int* __restrict__ next_count, int cap, const float* __restrict__ tgt, int* __restrict__ zero_count, int chunk, int ln) {
    const int env = chunk * FTPB + ln;
    if (chunk == 0 && ln < NB) zero_count[ln] = 0;   // the counter the step after this one appends to (idle now)
#ifdef PBRE_WAVE_TRACE
    if (g_wave_diag >= 6 && g_wave_diag <= 8) {      // 6, 7, 8: the step as it is, but every k_fast wave starts ~3 / 6 / 12 us late (does the row waves' setup -- dependent loads -- get through before the herd?)
        const unsigned long long t0 = __builtin_readcyclecounter(), wait = 6300ull << (g_wave_diag - 6);
        while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    } else
    if (const int diag = g_wave_diag) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        float x = (float)ln;
        while (__builtin_readcyclecounter() - t0 < 190000ull) {      // ~90 us at 2.1 GHz
            if (diag == 1) __builtin_amdgcn_s_sleep(32);
            else if (diag == 2) { PBRE_UNROLL for (int k = 0; k < 16; k++) x = __builtin_fmaf(x, 1.0000001f, 1e-9f); }
            else if (diag == 4) {      // 4: EIGHT independent v_fma chains in a 64-instruction loop: the vector unit saturated (a k_fast wave's matrix squarings)
                float y0 = x, y1 = x + 1.f, y2 = x + 2.f, y3 = x + 3.f, y4 = x + 4.f, y5 = x + 5.f, y6 = x + 6.f, y7 = x + 7.f;
                PBRE_UNROLL for (int k = 0; k < 8; k++) {
                    y0 = __builtin_fmaf(y0, 1.0000001f, 1e-9f); y1 = __builtin_fmaf(y1, 1.0000001f, 1e-9f); y2 = __builtin_fmaf(y2, 1.0000001f, 1e-9f); y3 = __builtin_fmaf(y3, 1.0000001f, 1e-9f);
                    y4 = __builtin_fmaf(y4, 1.0000001f, 1e-9f); y5 = __builtin_fmaf(y5, 1.0000001f, 1e-9f); y6 = __builtin_fmaf(y6, 1.0000001f, 1e-9f); y7 = __builtin_fmaf(y7, 1.0000001f, 1e-9f);
                }
                x = ((y0 + y1) + (y2 + y3)) + ((y4 + y5) + (y6 + y7));
            } else if (diag == 5) {    // 5: memory traffic: a simple env's lane reads its state record and writes it back (what k_fast moves per env, over and over)
                if (env < n && cls_cur[env] == 0) {
                    float* st_ = state + (size_t)env * STATE;
                    PBRE_UNROLL for (int k = 0; k < STATE; k++) { const float v_ = __builtin_nontemporal_load(st_ + k); x += v_; __builtin_nontemporal_store(v_, st_ + k); }
                }
            } else {
                // 3: the same dependent chain as 16384 instructions of straight-line code (128 KB: twice the instruction cache), streamed again and again
                // 9: 3 and 5 in turn -- code streaming AND memory traffic (an evicted row loop is re-fetched through a busy L2?)
                if (diag == 9 && env < n && cls_cur[env] == 0) {
                    float* st_ = state + (size_t)env * STATE;
                    PBRE_UNROLL for (int k = 0; k < STATE; k++) { const float v_ = __builtin_nontemporal_load(st_ + k); x += v_ * 0.f; __builtin_nontemporal_store(v_, st_ + k); }
                }
#define PBRE_DIAG_R4(s) s s s s
#define PBRE_DIAG_R16(s) PBRE_DIAG_R4(PBRE_DIAG_R4(s))
#define PBRE_DIAG_R256(s) PBRE_DIAG_R16(PBRE_DIAG_R16(s))
                PBRE_DIAG_R256(PBRE_DIAG_R16(PBRE_DIAG_R4(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(1.0000001f), "v"(1e-9f));)))
            }
        }
        if (x == 12345.678f) zero_count[0] = 1;      // (keeps the chain alive)
        return;
    }
#endif
    if (env >= n || cls_cur[env] != 0) return;      // classes of the state this step starts from (the kernels of the step write the next array)
    int c;
    if constexpr (CT)
        c = FastD::step<RT>(*(const CTables*)T, P, state + (size_t)env * STATE, (MODE & FastD::M_ACTION) ? actions + (size_t)env * act_dim : nullptr,
                            (MODE & FastD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags, P.env_id_base + (unsigned long long)env,
                            (MODE & FastD::M_TGT) ? tgt + (size_t)env * NJ : nullptr, (RT && P.sweeps) ? P.sweeps + env : nullptr);
    else
        c = FastD::step<RT>(*T, P, state + (size_t)env * STATE, (MODE & FastD::M_ACTION) ? actions + (size_t)env * act_dim : nullptr,
                            (MODE & FastD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags, P.env_id_base + (unsigned long long)env,
                            (MODE & FastD::M_TGT) ? tgt + (size_t)env * NJ : nullptr, (RT && P.sweeps) ? P.sweeps + env : nullptr);
    publish_class(env, c, cls, next_list, next_count, cap, P.bad_count);
}
template <int MODE, int WPS = PBRE_FAST_WAVES, bool RT = false>
__global__ __launch_bounds__(FTPB, WPS) void k_fast(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                               const float* __restrict__ actions, float* __restrict__ out, int n, int act_dim, int ow, int flags,
                                               const signed char* __restrict__ cls_cur, signed char* __restrict__ cls, int* __restrict__ next_list,
                                               int* __restrict__ next_count, int cap, const float* __restrict__ tgt, int* __restrict__ zero_count) {
    fast_wave<MODE, RT>(T, P, state, actions, out, n, act_dim, ow, flags, cls_cur, cls, next_list, next_count, cap, tgt, zero_count, (int)blockIdx.x, (int)threadIdx.x);
}

// Simple envs of a batch that leaves most SIMDs without a wave (a per-GPU shard of a strongly scaled batch, BASELINE configs 2 and 3):
// the same step as k_fast, spread over two waves per 64 envs.  In the simple class the robot's rows and the object's rows share no
// unknown, so wave 0 of a block does the robot's half (kinematics, dynamics, M^-1, the motor rows' closed form, integration, kinematics
// of the new state) while wave 1 does the object's (contact candidates, the 150 sweeps over its <= 12 rows, integration) on another SIMD
// of the CU; one block barrier, behind which wave 0 finds the object's new pose in LDS and writes observation, reward, done and the
// class.  A lone wave issues one instruction per ~5.4 cycles whatever its dependencies (profiles/r01_ubench_pkfma.txt), so a batch of
// lone waves steps in the LONGER half + the observation instead of the sum of both.  Same operations on the same operands as k_fast:
// bit-identical results (tests/test_gpu_parity.py), so which kernel a shard size selects is invisible in the data (sharding invariance).
constexpr int PTPB = 2 * FTPB;
// (pair_wave: one wave's work -- role 0 the robots, role 1 the objects of the 64 envs of `chunk`, exchange record px; k_fast_pair's body and,
// round 5, the simple envs' part of k_fused<.., true>.  One block barrier per wave that has a simulating lane.)
template <int MODE, bool CT = true>
__device__ __forceinline__ void pair_wave(const Tables* __restrict__ T, const Params& P, float* __restrict__ state,
                                          const float* __restrict__ actions, float* __restrict__ out, int n, int act_dim, int ow, int flags,
                                          const signed char* __restrict__ cls_cur, signed char* __restrict__ cls, int* __restrict__ next_list,
                                          int* __restrict__ next_count, int cap, const float* __restrict__ tgt, int* __restrict__ zero_count,
                                          PairX& px_lds, int chunk, int ln, int role, float* __restrict__ pg = nullptr, int seq = 0) {
    // pg != nullptr (k_fused's tail pairs): the two waves are two one-wave BLOCKS and the pair's record is the chunk's GLOBAL one (a PairX + 64
    // sequence words, PBRE_PAIR_SYNC): the object wave only ever touches px.o, the robot wave px.sc / px.g / px.seq
    PairX& px = pg ? *(PairX*)pg : px_lds;
    if (role == 0) { px.g = pg; px.seq = seq; }                // (every lane the same values; read back by this wave alone)
    const int env = chunk * FTPB + ln;
    if (chunk == 0 && role == 0 && ln < NB) zero_count[ln] = 0;
    const bool live = env < n && cls_cur[env] == 0;
    if (!PBRE_ANY(live)) return;              // the same decision in both waves of the block (same envs): no barrier is left waiting
    if (!live) return;
    float* st = state + (size_t)env * STATE;
    const bool sim = st[46] == 0.f;           // (action_repeat > 1 only: the env already left this env.step()'s apply_action loop)
    if (role == 0) {
        const float* a = (MODE & FastD::M_ACTION) ? actions + (size_t)env * act_dim : nullptr;
        float* o = (MODE & FastD::M_OBS) ? out + (size_t)env * ow : nullptr;
        const unsigned long long id = P.env_id_base + (unsigned long long)env;
        int c;
        if constexpr (CT) {
            if (sim) c = FastD::step_t<false, 1>(*(const CTables*)T, P, st, a, o, MODE, flags, id, (MODE & FastD::M_TGT) ? tgt + (size_t)env * NJ : nullptr, &px, ln);
            else c = FastD::skipped(*(const CTables*)T, P, st, o, MODE, flags, id);
        } else {
            if (sim) c = FastD::step_t<false, 1>(*T, P, st, a, o, MODE, flags, id, (MODE & FastD::M_TGT) ? tgt + (size_t)env * NJ : nullptr, &px, ln);
            else c = FastD::skipped(*T, P, st, o, MODE, flags, id);
        }
        publish_class(env, c, cls, next_list, next_count, cap, P.bad_count);
    } else {
        if constexpr (CT) { if (sim) (void)FastD::step_t<false, 2>(*(const CTables*)T, P, st, nullptr, nullptr, MODE, flags, 0ull, nullptr, &px, ln); }
        else if (sim) (void)FastD::step_t<false, 2>(*T, P, st, nullptr, nullptr, MODE, flags, 0ull, nullptr, &px, ln);
        if (pg) { if (sim) __hip_atomic_store((int*)((char*)pg + PBRE_PAIR_G_FLAGS) + ln, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }     // its stores (state record, pg) first
        else __syncthreads();                 // its stores (state record, LDS) are complete before the robot wave goes on
    }
}
#ifndef PBRE_PAIR_WPS
#define PBRE_PAIR_WPS 2            // waves per SIMD k_fast_pair is register-limited to (A/B: tools/build_variant.sh)
#endif
template <int MODE>
__global__ __launch_bounds__(PTPB, PBRE_PAIR_WPS) void k_fast_pair(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                               const float* __restrict__ actions, float* __restrict__ out, int n, int act_dim, int ow, int flags,
                                               const signed char* __restrict__ cls_cur, signed char* __restrict__ cls, int* __restrict__ next_list,
                                               int* __restrict__ next_count, int cap, const float* __restrict__ tgt, int* __restrict__ zero_count) {
    __shared__ PairX px;
    // role: wave-uniform, 0 robot, 1 object
    pair_wave<MODE>(T, P, state, actions, out, n, act_dim, ow, flags, cls_cur, cls, next_list, next_count, cap, tgt, zero_count, px, (int)blockIdx.x,
                    (int)(threadIdx.x & (FTPB - 1)), __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
}

// Complex envs (robot contacts and/or limit rows), compacted per class.  Persistent blocks (the host does not know the
// list lengths): work item w = (bucket, 64-env chunk); block b takes items b, b + gridDim.x, ...  The grid is one block
// per SIMD (the kernel needs a whole SIMD's register file), blocks without work exit at once.
// Scheduling hints for the host (pinned memory, written by one thread of the complex-env kernel of every step): [0] complex envs
// stepped in this step, [1] 16 while there were any, counting down by one per step once there are none -- "no complex env for
// the last 16 steps" is what lets the host drop the second stream (launch_step).  `recent` is the device copy of [1].
static __device__ __forceinline__ void report_hint(int total, int* __restrict__ recent, int* __restrict__ host_total) {
    int r = *recent;
    r = total > 0 ? 16 : (r > 0 ? r - 1 : 0);
    *recent = r;
    recent[1] += total;                 // running sum of complex env-steps (pbre_kernel_info[7]; wraps at 2^31)
    host_total[0] = total; host_total[1] = r;
}

template <int MODE, bool RT = false>
__global__ __launch_bounds__(FTPB) void k_fast_rc(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                                  const float* __restrict__ actions, float* __restrict__ out, int act_dim, int ow, int flags,
                                                  const int* __restrict__ cur_list, const int* __restrict__ cur_count,
                                                  signed char* __restrict__ cls, int* __restrict__ next_list, int* __restrict__ next_count, int cap,
                                                  const float* __restrict__ tgt, int* __restrict__ host_total, int* __restrict__ recent) {
    if (PBRE_RC_PRIO > 0) __builtin_amdgcn_s_setprio(PBRE_RC_PRIO);      // see k_row_list
    int chunks[NB], total = 0, envs = 0;
    PBRE_UNROLL for (int b = 0; b < NB; b++) { chunks[b] = (cur_count[b] + FTPB - 1) / FTPB; total += chunks[b]; envs += cur_count[b]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) report_hint(envs, recent, host_total);
    for (int w = blockIdx.x; w < total; w += gridDim.x) {
        int b = 0, k = w;
        PBRE_UNROLL for (int j = 0; j < NB - 1; j++) if (b == j && k >= chunks[j]) { k -= chunks[j]; b = j + 1; }
        const int i = k * FTPB + threadIdx.x;
        if (i < cur_count[b]) {
            const int env = cur_list[(size_t)b * cap + i];
            const int c = FastD::step_rc<RT>(*(const CTables*)T, P, state + (size_t)env * STATE, (MODE & FastD::M_ACTION) ? actions + (size_t)env * act_dim : nullptr,
                                             (MODE & FastD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags, P.env_id_base + (unsigned long long)env,
                                             (MODE & FastD::M_TGT) ? tgt + (size_t)env * NJ : nullptr, (RT && P.sweeps) ? P.sweeps + env : nullptr);
            publish_class(env, c, cls, next_list, next_count, cap, P.bad_count);
        }
    }
}

// Complex envs, few of them: one env per 16-lane row (4 per wave) over the compacted list.  The physics of the step is the
// general row kernel's (Core::step: all row types), the observation / reward / termination / auto-reset / class of the new
// state is the lane-per-env kernels' Fast::finish run by lane 0 of the row, so both complex-env kernels are interchangeable.
// A row spreads an env over 16 lanes, so a wave's latency is ~1/3 of a k_fast_rc wave's: with few complex envs the step is
// no longer gated by that latency.  Grid-stride over the list (the host only has a hint of its length).
// The object of a complex env WITHOUT robot-object contact (95 % of them) shares no unknown with the robot's rows, so its half of the
// step is not the row waves' business: a fifth wave of the block steps the objects of the block's 16 envs, one lane per env (ObjStep,
// pbre_objstep.hpp: the same rows, 150 sweeps, as a lane-per-env chain of ~45 us), and leaves the new twists in LDS.  A row wave none
// of whose four envs has such a contact neither builds nor sweeps the object-table rows -- its sweep is the robot's rows alone, about
// half the instructions -- and picks the twist up at the end (Core::step's `objv`, the mechanism of the iCub's kw_obj); an env WITH a
// robot-object contact solves the coupled system as before (zipped sweeps), and so do its wave-mates, whose own object result is then
// dropped in favour of the side record's: what an env computes never depends on the envs it shares a wave with.
// The block is four waves -- three row waves (12 envs) and the object wave -- so that each has a SIMD of the CU to itself: as a fifth
// wave the object wave shared its SIMD with a row wave and both ran at little more than half their lone speed (measured, phase probe).
constexpr int REPB = 12;             // envs per block of k_row_list
constexpr int RTPB = REPB * W + FTPB;
// (RT: the object wave idles -- Bullet's exit test is a maximum over ALL rows of an env, so the object's rows stay in the row wave's solve)
// (row_list_block: the work of block `bid` of `nblk` -- k_row_list's body and, round 5, the complex envs' part of k_fused.
// G = false: a 256-thread block, vt = threadIdx.x, the object wave hands its twists to the row waves in LDS (objv) across a block barrier.
// G = true: the same four waves as four 64-thread blocks (vt = the thread index the wave would have had; the object wave's block first), the
// twists travel through the per-env global records objv_g[env][W], whose first word the object wave sets to P.objv_seq last (release); no
// barrier anywhere.  The object wave waits for nothing, and its block is dispatched before its row waves' blocks: the row waves' wait ends.)
template <int MODE, bool RT, bool G = false>
__device__ __forceinline__ void row_list_block(const Tables* __restrict__ T, const Params& P, float* __restrict__ state,
                                               const float* __restrict__ actions, float* __restrict__ out, int act_dim, int ow, int flags,
                                               const int* __restrict__ cur_list, const int* __restrict__ cur_count,
                                               signed char* __restrict__ cls, int* __restrict__ next_list, int* __restrict__ next_count, int cap,
                                               const float* __restrict__ tgt, int* __restrict__ host_total, int dummy_base, int* __restrict__ recent,
                                               float (*objv)[W], int bid, int nblk, int vt, float* __restrict__ objv_g = nullptr) {
    static_assert(NB <= 2 || MODE < 0, "the row kernel walks one complex list (PBRE_NCLASS=2) or two (3: uncoupled / coupled)");
    // These few waves are the tail of the step: each shares its SIMD with a k_fast wave, and a row wave is latency-bound (it leaves
    // most issue slots to its neighbour anyway), so it gets the higher wave priority and runs at its lone-wave speed.
    if (PBRE_RC_PRIO > 0) __builtin_amdgcn_s_setprio(PBRE_RC_PRIO);
    // Work items of REPB rows: bucket 0 (complex envs without robot-object contact) fills every row, four envs per wave; bucket 1 (the
    // coupled ones, PBRE_NCLASS=3) gets ONE env per wave -- rows 1..3 of the wave idle on the pristine dummy record -- so that the long
    // coupled sweep runs over that env's own row slots only and slows nobody else down.
    const int total0 = cur_count[0], total1 = NB > 1 ? cur_count[NB > 1 ? 1 : 0] : 0;
    const int items0 = (total0 + REPB - 1) / REPB, items1 = (total1 + REPB / 4 - 1) / (REPB / 4);
    if (bid == 0 && vt == (G ? REPB * W : 0)) report_hint(total0 + total1, recent, host_total);
    const bool obj_on = !(flags & 1) && !RT;      // (here: "the object wave solves the objects of the block's envs")
    const bool obj_wave = __builtin_amdgcn_readfirstlane((int)(vt >= REPB * W)) != 0;
    const int row = obj_wave ? vt - REPB * W : (vt >> 4);
    constexpr int PHYS = MODE & (CoreD::M_ACTION | CoreD::M_TGT);
    // (the coupled envs' items come first: theirs are the longest waves of the step)
    for (int item = bid; item < items0 + items1; item += nblk) {
        const bool coupled = item < items1;
        const int* __restrict__ lst = coupled ? cur_list + cap : cur_list;
        const int total = coupled ? total1 : total0;
        const int i = coupled ? item * (REPB / 4) + (row >> 2) : (item - items1) * REPB + row;
        const bool real = i < total && row < REPB && !(coupled && (row & 3) != 0);
        // Idle rows run in lockstep with the real ones and the wave pays for the rows of its heaviest group, so what they step must be the
        // cheapest state there is -- and stay it: they READ a pristine dummy record (the un-settled reset pose pbre_create wrote: arm at
        // home, object in the air, no contact, no joint at a limit) and WRITE their result to a scratch record EPB further on.  (Until
        // round 4 they stepped the dummy record in place, with env 0's actions, launch after launch and without ever being reset: a
        // random walk into joint limits and table contacts that made every partially filled wave carry the longest chain of the step.)
        const int env = real ? lst[i] : dummy_base + (row < REPB ? row : 0);
        float* st = state + (size_t)env * STATE;
        float* st_idle = real ? nullptr : st + (size_t)EPB * STATE;
        if (obj_wave) {
            if (obj_on) {
                if (row < REPB) {
                    float pose[7], tw[6], o[6];
                    PBRE_UNROLL for (int k = 0; k < 7; k++) pose[k] = st[CoreD::LC + k];
                    PBRE_UNROLL for (int k = 0; k < 6; k++) tw[k] = st[W + CoreD::LC + k];
                    // per-env object parameters (pbre_set_physics_per_env): X[12] mass, X[13] lateral friction, X[15] 1 + linear damping
                    const float o_m = st[44] > 0.f ? st[44] : P.obj_m, o_mu = st[45] > 0.f ? st[45] : P.obj_mu, o_kl = st[47] > 0.f ? st[47] - 1.f : P.kl;
                    ObjStep::run_p(P, pose, tw, o, o_m, o_mu, o_kl);
                    if constexpr (G) {
                        float* r = objv_g + (size_t)env * W;
                        PBRE_UNROLL for (int k = 0; k < 6; k++) r[CoreD::LC + k] = o[k];
                        __hip_atomic_store((int*)r, P.objv_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        PBRE_UNROLL for (int k = 0; k < 6; k++) objv[row][CoreD::LC + k] = o[k];
                    }
                }
                if constexpr (!G) __syncthreads();                               // pairs with PBRE_OBJV_SYNC in the row waves' Core::step
            }
        } else {
            CoreD::step<RT>(*T, P, st, (MODE & CoreD::M_ACTION) ? actions + (size_t)(real ? env : 0) * act_dim : nullptr, nullptr, PHYS, flags,
                            (MODE & CoreD::M_TGT) ? tgt + (size_t)env * NJ : nullptr, 0ull,
                            obj_on ? (G ? objv_g + (size_t)env * W : &objv[row < REPB ? row : 0][0]) : nullptr, st_idle, (RT && real && P.sweeps) ? P.sweeps + env : nullptr);
            // the row's stores are read back by its lane 0 below: same wave, so a WORKGROUP-scope fence (wait for the stores; the CU's L1 is
            // write-through).  Until round 5 this was __atomic_thread_fence(SEQ_CST) = system scope: a write-back AND invalidate of the XCD's
            // whole L2 -- full of the state records the simple envs' waves are writing -- on every row wave's critical path
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            PBRE_PROBE_DECL
            // (PBRE_ROW_FINISH16: every lane of the group runs the same scalar code on the same inputs -- and stores the same values to the same
            // addresses --, the sphere tests are dealt out one per lane (Fast::sweep<3>); lane 0 publishes the class)
            if (real && (PBRE_ROW_FINISH16 || (vt & 15) == 0)) {
                float q[NJ], qd[NJ];
                PBRE_UNROLL for (int j = 0; j < NJ; j++) { q[j] = st[j]; qd[j] = st[16 + j]; }
                FastD::V3 op; op.x = st[9]; op.y = st[10]; op.z = st[11];
                FastD::Q4 oq; oq.x = st[12]; oq.y = st[13]; oq.z = st[14]; oq.w = st[15];
                // (the tables through the constant address space: scalar loads although the row's stores precede them -- Fast::finish)
                const int c = FastD::finish<PBRE_ROW_FINISH16 ? 3 : 0>(*(const CTables*)T, P, st, q, qd, op, oq, (MODE & FastD::M_OBS) ? out + (size_t)env * ow : nullptr, MODE, flags,
                                            P.env_id_base + (unsigned long long)env, false, nullptr, vt & 15);
                if ((vt & 15) == 0) publish_class(env, c, cls, next_list, next_count, cap, P.bad_count);
            }
            PBRE_PROBE(11);     // Fast::finish on lane 0 of each row
        }
        if constexpr (!G) if (obj_on && item + nblk < items0 + items1) __syncthreads();   // the side records are rewritten by the next trip
    }
}
template <int MODE, bool RT = false>
__global__ __launch_bounds__(RTPB, 2) void k_row_list(const Tables* __restrict__ T, const Params P, float* __restrict__ state,
                                                  const float* __restrict__ actions, float* __restrict__ out, int act_dim, int ow, int flags,
                                                  const int* __restrict__ cur_list, const int* __restrict__ cur_count,
                                                  signed char* __restrict__ cls, int* __restrict__ next_list, int* __restrict__ next_count, int cap,
                                                  const float* __restrict__ tgt, int* __restrict__ host_total, int dummy_base, int* __restrict__ recent) {
    __shared__ float objv[REPB][W];
    row_list_block<MODE, RT>(T, P, state, actions, out, act_dim, ow, flags, cur_list, cur_count, cls, next_list, next_count, cap, tgt, host_total, dummy_base, recent,
                             objv, (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}

// Round 5: ONE kernel per step.  The two kernels of a step ran on two streams with a fork and a join event between them, and the step-kernel
// timeline (profiles/r05k_step_kernels.txt) showed what that costs at 131072 envs: the second kernel starts 5.6 us after the first and the next
// step's first kernel 12.8 us after this step's last one ends -- 13 us of a 179-us step in which nothing runs.  k_fused is both kernels in one
// grid of 256-thread blocks: the first `rblocks` blocks are k_row_list's (dispatched first, so the step's longest waves -- the coupled envs' --
// start first, as the rc_first order arranged before), every later block is four k_fast waves (PAIR: two robot / object wave pairs of
// k_fast_pair, whose one barrier then spans both pairs).  Same device functions, same arithmetic: which launch form a step took is invisible
// in the data (tests: PBRE_FUSED=0 against the default, bit for bit).  256 VGPRs like both of its parts; the 168-VGPR build of k_fast
// (PBRE_FAST3) has no fused counterpart -- the row waves need 248 -- and is not needed: at 131072 envs k_fast<7, 2> beside the row waves
// measured 151 us against 157 us for <7, 3> (same file), and it does not spill.
constexpr int FUSED_WAVES = RTPB / FTPB;
static_assert(RTPB % FTPB == 0 && FUSED_WAVES % 2 == 0, "k_fused: whole waves, whole pairs");
static_assert(sizeof(PairX) <= PBRE_PAIR_G_FLAGS, "a tail pair's global record: the sequence words behind the PairX");
// The arguments travel as ONE struct and each role reads them through its own (laundered) pointer into the kernarg segment.  As plain kernel
// arguments all of them -- Params is ~600 bytes -- are loaded in the entry block and stay live in SGPRs through whichever role the block takes:
// the first build of this kernel had 5.7 k more v_readlane_b32 (SGPRs spilled to VGPR lanes, re-read inside the sweep loops) than its two
// parts together, and its row waves' sweeps ran 11-32 % slower than k_row_list's (phase probe, profiles/r05m_phase_probe.txt).
struct FusedArgs {
    const Tables* T; Params P; float* state; const float* actions; float* out; int n, act_dim, ow, flags;
    const signed char* cls_cur; signed char* cls; const int* cur_list; const int* cur_count; int* next_list; int* next_count; int cap;
    const float* tgt; int* zero_count; int* host_total; int dummy_base; int* recent; int rblocks;
    float* objv_g;                            // (k_fused<.., false>) [cap + 32][W] the object waves' twists, first word = the launch's sequence number (P.objv_seq)
    float* pair_g;                            // (k_fused<.., false>) [chunks] the tail pairs' global records (a PairX + 64 sequence words, PBRE_PAIR_SYNC), sequence number P.objv_seq
    int ntail;                                // the last `ntail` 64-env chunks are stepped by a robot wave + an object wave (two blocks each) instead of one k_fast wave
};
// A tail pair's wave (k_fused<.., false, false, TAIL = true>): role 0 the robots, 1 the objects of the 64 envs of chunk `tchunk`.
// (Measured, profiles/r06q / r06r: the two roles inlined beside the row roles and the k_fast role change the kernel's ONE register allocation -- the k_fast role came out
// with 50 % more SGPR spills re-read in its loops, v_readlane 972 -> ~1500, and the FRESH step at 131072 envs went 0.0905 -> 0.097 ms -- so the kernel WITH tail pairs is an
// instantiation of its own, launched only for steps that will have displaced chunks; a step without complex envs runs the kernel of before.  As a non-inlined FUNCTION
// the pair's wave cost the kernel 656 B of scratch per lane for the call ABI and crashed on the device: dropped.)
template <int MODE>
__device__ __forceinline__ void tail_pair_role(const FusedArgs PBRE_CONST_AS* a, const Tables* __restrict__ T, int tchunk, int role, int ln) {
    float* pg = (float*)((char*)a->pair_g + (size_t)tchunk * PBRE_PAIR_G_BYTES);
    const int seq = a->P.objv_seq;
    PBRE_LAUNDER(a);
    pair_wave<MODE, true>(T, *(const Params*)&a->P, a->state, a->actions, a->out, a->n, a->act_dim, a->ow, a->flags, a->cls_cur, a->cls, a->next_list, a->next_count,
                          a->cap, a->tgt, a->zero_count, *(PairX*)pg, tchunk, ln, role, pg, seq);
}
// RT (round 6): Bullet's residual exit -- the 64-thread grid only (the pair mapping splits an env over two waves, the exit test is a maximum
// over all of its rows); the row blocks' object waves idle (Core::step<RT> sweeps the object's rows itself).
// TAIL: the instantiation with tail pairs (64-thread grid, no residual exit), see tail_pair_role
template <int MODE, bool PAIR, bool RT = false, bool TAIL = false>
__global__ __launch_bounds__(PAIR ? RTPB : FTPB, 2) void k_fused(const FusedArgs args_in_kernarg_segment, const Tables* __restrict__ T) {
    static_assert(!(PAIR && RT), "the residual exit steps an env on one lane");
    static_assert(!TAIL || (!PAIR && !RT), "tail pairs: the 64-thread grid of the default step");
    // (T -- the model constants -- is a kernel argument of its own: as a __restrict__ argument it cannot alias the pointers the roles load from the
    // struct, so the reads through it stay scalar loads behind the roles' stores; read from the struct, 50 of k_fast's s_load_dwordx16 / x8 table
    // reads had become per-lane global loads and the fast role spilled 488 bytes per lane)
    const FusedArgs PBRE_CONST_AS* a = (const FusedArgs PBRE_CONST_AS*)__builtin_amdgcn_kernarg_segment_ptr();      // (explicit arguments start at offset 0)
    if constexpr (PAIR) {
        // 256-thread blocks: a row block as in k_row_list / two robot-object wave pairs of k_fast_pair (whose one barrier then spans both pairs)
        if ((int)blockIdx.x < a->rblocks) {
            __shared__ float objv[REPB][W];
            PBRE_LAUNDER(a);
            row_list_block<MODE, false>(T, *(const Params*)&a->P, a->state, a->actions, a->out, a->act_dim, a->ow, a->flags, a->cur_list, a->cur_count, a->cls,
                                        a->next_list, a->next_count, a->cap, a->tgt, a->host_total, a->dummy_base, a->recent, objv, (int)blockIdx.x, a->rblocks,
                                        (int)threadIdx.x);
            return;
        }
        const int fb = (int)blockIdx.x - a->rblocks, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), ln = (int)(threadIdx.x & (FTPB - 1));
        PBRE_LAUNDER(a);
        __shared__ PairX px[FUSED_WAVES / 2];
        pair_wave<MODE, true>(T, *(const Params*)&a->P, a->state, a->actions, a->out, a->n, a->act_dim, a->ow, a->flags, a->cls_cur, a->cls, a->next_list, a->next_count, a->cap,
                        a->tgt, a->zero_count, px[wv >> 1], fb * (FUSED_WAVES / 2) + (wv >> 1), ln, wv & 1);
    } else {
        // 64-thread blocks, like k_fast's: beside a machine-filling batch (131072 envs = 2048 waves = every wave slot) the row waves displace some
        // hundred k_fast waves into a second round, and those must be free to start wherever ONE wave slot frees up -- as whole 4-wave blocks
        // they waited for four free slots on one CU, and the fast part of the grid ended at 173 us against 151 us (profiles/r05l_*).  So a row
        // block is four one-wave blocks here: 4 s the object wave of slot s, 4 s + 1 .. 3 its row waves (row_list_block<.., G = true>).
        const int rb = FUSED_WAVES * a->rblocks;
        if ((int)blockIdx.x < rb) {
            const int role = (int)blockIdx.x & (FUSED_WAVES - 1);
            PBRE_LAUNDER(a);
            row_list_block<MODE, RT, true>(T, *(const Params*)&a->P, a->state, a->actions, a->out, a->act_dim, a->ow, a->flags, a->cur_list, a->cur_count, a->cls,
                                              a->next_list, a->next_count, a->cap, a->tgt, a->host_total, a->dummy_base, a->recent, nullptr,
                                              (int)blockIdx.x / FUSED_WAVES, a->rblocks, (role == 0 ? REPB * W : (role - 1) * FTPB) + (int)threadIdx.x, a->objv_g);
            return;
        }
        const int chunk = (int)blockIdx.x - rb;
        if constexpr (TAIL) {
            // Tail pairs (round 6, last session).  A batch that fills every wave slot (131072 envs = 2048 k_fast waves on 2048 slots) runs as many
            // k_fast waves in a SECOND round as the row / object waves hold slots, and those start when the first k_fast waves end: the step was
            // first round + one lone k_fast wave (~90 + ~60 us) whatever the chain did.  The chunks that will be displaced -- the last ones of the
            // grid, `ntail` of them by the host's hint -- are split like k_fast_pair's: an object wave (its block first) and a robot wave, which
            // in the second round find free SIMDs and take the LONGER half instead of the sum.  Same operations on the same operands as k_fast
            // (step_t<false, 1 / 2>): which mapping a chunk took is invisible in the data.
            const int nfast = (a->n + FTPB - 1) / FTPB - a->ntail;
            if (chunk >= nfast) {
                const int t = chunk - nfast;
                tail_pair_role<MODE>(a, T, nfast + (t >> 1), (t & 1) ^ 1, (int)threadIdx.x);      // even t: the object wave (dispatched first)
                return;
            }
        }
        PBRE_LAUNDER(a);
        fast_wave<MODE, RT, true>(T, *(const Params*)&a->P, a->state, a->actions, a->out, a->n, a->act_dim, a->ow, a->flags, a->cls_cur, a->cls, a->next_list, a->next_count,
                               a->cap, a->tgt, a->zero_count, chunk, (int)threadIdx.x);
    }
}

// ------------------------------------------------------------------ context
struct EnvBuf {                       // a batch of state records with its class bookkeeping
    float* state = nullptr;           // cap + 16 records
    signed char* cls = nullptr;       // [2][cap] class per env: cls + cur*cap describes the current state, a step writes the other half (the
                                      // two kernels of a step run concurrently, so k_fast must not see classes the other one just produced)
    float* tgt = nullptr;             // [cap + 16][NJ] joint targets of the IK mode
    int* list[2] = {nullptr, nullptr};  // each [NB][cap]
    int* count = nullptr;             // [3][NB]: counters rotate over three buffers so that the one the step after next
                                      // appends to can be zeroed by a kernel of the current step (no memset on the hot path)
    int cur = 0, ccur = 0;            // list[cur] / count + ccur*NB: complex envs of the current state, per class
    float* objv_g = nullptr;          // [cap + 32][W] k_fused's side records: the object waves' twists for the row waves of other blocks
    int objv_seq = 0;                 // sequence number of the last k_fused launch (the records' "complete" mark)
    float* pair_g = nullptr;          // [(cap + 63) / 64] records of PBRE_PAIR_G_BYTES: k_fused's tail pairs (a PairX + per-lane sequence words)
    int* h_total = nullptr;           // pinned host int the device writes the complex-env count of the step it runs into
    int cap = 0;
};

struct pbre_ctx {
    WideEngine* wide = nullptr;        // robots with more than 9 DoF (iCub): every entry point forwards to the 64-lane engine
    pbre_config cfg;
    Tables T; Params P;
    int n = 0, npad = 0, obs_dim = 0, act_dim = 0, ow = 0, device = 0;
    Tables* dT = nullptr;
    EnvBuf main, tmp;
    float *d_act = nullptr, *d_out = nullptr, *d_scratch = nullptr;
    int* d_bad = nullptr;              // NaN / Inf guard: env-steps that met a non-finite state (Params::bad_count)
    float* d_hull = nullptr;           // PBRE_SHAPE_HULL: the object's vertex / face table (Params::hull; pbre_set_object_hull)
    int* d_sweeps = nullptr;           // [npad] sweeps every env's solver ran in the last step (Params::sweeps; pbre_physics.solver_residual_threshold > 0)
    unsigned long long* d_ids = nullptr; unsigned* d_ep = nullptr; int* d_idx = nullptr;
    bool fast_ok = false;
    int n_simd = 1024;
    int rc_first_min = 1;              // complex envs (reported by the device) from which their kernel is scheduled ahead of k_fast
    int idle_touch = 16;               // in that mode, an (empty) fork / join through the side stream every idle_touch-th step: a side stream
                                       // left idle for hundreds of steps makes the first steps after the switch back ~8 % slower (0: never)
    int idle_single = 1;               // with no complex envs reported, both kernels go to the caller's stream in order (no fork / join events); 0: A/B
    int row_max = 4096;                // up to this many complex envs they are stepped by the row kernel (1 wave per 4 envs)
    int pair = 2;                      // k_fast_pair (robot wave + object wave per 64 envs): 0 never, 1 whenever it applies, 2 while its waves fit two per SIMD (PBRE_PAIR)
    long launches_pair = 0;
    int tail_pair = 0;                 // k_fused's 64-thread grid: the chunks the row waves displace into a second round as robot / object wave pairs (PBRE_TAIL_PAIR: 0 never,
                                       // 1 by the hint, n > 1: always the last n chunks -- tests).  Measured (profiles/r06s_final_structure_ab.txt, r06t_bench_tail_pairs_by_hint.json):
                                       // by the hint the stationary mix at 131072 envs gains 1.3 % (0.1494 -> 0.1476 ms) and a batch with FEW complex envs -- synchronised
                                       // episode clocks, ~30 per step -- loses 34 % (0.091 -> 0.122: two dozen pairs at the end of the grid start when the first round ends): off
    long launches_tail = 0;
    int fast3 = 2;                     // k_fast variant limited to 3 waves per SIMD: 0 never, 1 whenever complex envs are reported, 2 when they would displace k_fast waves (PBRE_FAST3)
                                       // -- only where the step is NOT one fused launch (PBRE_FUSED=0, residual exit, action_repeat's inner steps).  (Round 5 also measured the
                                       // spill-free build in two half-grid launches back to back, one wave per SIMD each: 0.21 ms against 0.177, profiles/r05_fast3_halves_ab.txt.)
    int fused = 1;                     // the step as ONE launch (k_fused: row-list blocks + fast / pair blocks in one grid); 0: the two kernels on two streams (PBRE_FUSED)
    unsigned long long launches_fused = 0;
    hipStream_t stream = nullptr, side = nullptr;      // side: the candidate of `sp` that overlaps with the caller's stream
    SidePick sp;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // pipelined host path (pbre_step_async / pbre_step_wait, round 6): two (actions, rows) device buffer pairs and the events that order upload
    // -> step -> download of each slot; created on first use.  The upload rides on the ctx's stream ahead of the step, the download on the side
    // stream the overlap probe picked (pbre_sidepick.hpp: HIP multiplexes streams onto a few hardware queues, and streams that share one do
    // not overlap -- three more streams of their own measured 0.42 ms per step in a small process and 0.95 ms in bench.py's)
    struct AsyncPath {
        float* d_act[2] = {nullptr, nullptr};
        float* d_rows[2] = {nullptr, nullptr};
        hipEvent_t ev_step[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
        long issued = 0, waited = 0;
        bool ready = false;
    } ap;
    int async_blocks = 128;            // PBRE_ASYNC_BLOCKS: 256-thread blocks of the row-download kernel (k_rows_out)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join_sys = nullptr;   // ev_join_sys: with the system-scope fence (see pbre_step)
    bool rows_to_host = false;         // the step in flight writes its output rows straight into page-locked host memory
    static constexpr int KRING = 64;           // HIP event pairs around the dominant kernel of the last KRING sampled steps,
    static constexpr int KSAMPLE = 8;          // (every KSAMPLE-th launch is sampled) recorded on the stream that kernel runs on
    int ksample = KSAMPLE;                     // PBRE_KSAMPLE: A/B of the sampling interval
    hipEvent_t ev_k[KRING][2] = {};
    long k_steps = 0, launches = 0, launches3 = 0;      // launches3: steps whose k_fast was the 3-waves-per-SIMD variant
    double ms[3] = {0, 0, 0};
    int zero_copy = 3;                 // PBRE_ZERO_COPY: pbre_step lets the kernels access page-locked host buffers directly (bit 0 actions, bit 1 rows; 0: staged copies)
    bool have_snapshot = false;        // a full pbre_reset has recorded the settled snapshot (rst_q, rst_objz)
    bool stale_snapshot = false;       // ... and a later pbre_set_physics changed the scene it was recorded in
    unsigned char* d_mask = nullptr;
    bool ext_dirty = false;            // a pbre_step_device was enqueued on a caller-supplied stream since the last quiesce()
    std::string err;
};
static inline int ceil16(int n) { return (n + EPB - 1) / EPB * EPB; }
// (a convex-hull object is stepped by the general row kernel: the lane-per-env kernels are compiled for the primitives)
static inline bool lane_per_env(const pbre_ctx* c) { return c->fast_ok && !(c->cfg.flags & PBRE_F_FORCE_GENERAL) && c->P.obj_shape != PBRE_SHAPE_HULL; }

// one batched step of the first n envs of b on stream s
template <int MODE, bool RT>
hipError_t launch_step_t(pbre_ctx* c, EnvBuf& b, int n, const float* act, float* out, int flags, hipStream_t s) {
    // (the per-env sweep counts -- pbre_get_sweeps, residual exit -- are indexed by the env's place in the MAIN buffer; settle and
    // partial-reset steps of the compacted tmp buffer must not write theirs over other envs' counts)
    Params Pk = c->P;
    if (&b != &c->main) Pk.sweeps = nullptr;
    if (!lane_per_env(c)) {
        hipEvent_t* ek = c->ev_k[c->k_steps % pbre_ctx::KRING];
        (void)hipEventRecord(ek[0], s);
        hipLaunchKernelGGL((k_step<MODE, RT>), dim3(ceil16(n) / EPB), dim3(TPB), 0, s, c->dT, Pk, b.state, act, out, c->d_scratch, n, ceil16(n),
                           c->act_dim, c->ow, flags, b.tgt);
        (void)hipEventRecord(ek[1], s);
        c->k_steps++;
        return hipGetLastError();
    }
    flags |= c->cfg.flags & (PBRE_F_SEQ_MOTORS | PBRE_F_SEQ_OBJECT);
    const int cur = b.cur, nxt = cur ^ 1;
    const int cc = b.ccur, cn = (cc + 1) % 3, cz = (cc + 2) % 3;      // counters: current, next (zero on entry), the one after
    const int blocks = (n + FTPB - 1) / FTPB;
    hipError_t e;
    // The two kernels of a step run concurrently on two streams (fork/join events) and both append to list[nxt].
    // Which one gets the caller's stream is a scheduling choice made from the complex-env count the device reported for
    // an earlier step (a hint; either order is correct):
    //  * complex envs present: k_fast_rc (few waves, each needs a whole SIMD's register file, long latency) is enqueued first on
    //    the caller's stream so that its waves claim their SIMDs before k_fast floods the chip from the side stream;
    //  * none (e.g. the first steps after a reset): the (empty) complex-env kernel and k_fast are enqueued in order on the caller's
    //    stream, with no fork / join events at all: the event packets and the concurrently dispatched empty kernel cost 6 % of the
    //    step (613 M -> 650 M env-steps/s at 131072 envs).  Taken only when the device has reported no complex env for 16 steps in a
    //    row (a count that flickers between 0 and a few would otherwise serialise the two kernels every other step); a stale
    //    hint only serialises them for that step.
    const int hint = b.h_total[0];
    // complex envs: row kernel while they are few (latency), lane-per-env k_fast_rc when many (throughput)
    bool rows = NB <= 2 && hint <= c->row_max;
    if (c->cfg.flags & PBRE_F_COMPLEX_ROWS) rows = NB <= 2;
    if (c->cfg.flags & PBRE_F_COMPLEX_LANES) rows = false;
    if (!c->P.obj_iso || c->P.obj_shape != 0) rows = NB <= 2;      // k_fast_rc's object rows assume a cube: other boxes and the round objects' complex envs go to the row kernel
    // (the host knows the complex envs' total, not how many of them are coupled -- one env per wave: about a tenth, generously)
    // At least 64 blocks whatever the hint says: the hint is the count of a step the DEVICE has finished, and a host that runs ahead
    // of it (the 201 launches of a reset are enqueued in ~1 ms) sizes every launch by a count that may be a hundred steps old --
    // 16384 envs that had all become complex meanwhile were walked by 8 blocks, 11 ms per launch.  Blocks without work exit at once.
    const int rblocks = std::max(64, std::min(c->n_simd / 4, (hint + REPB - 1) / REPB + (NB > 1 ? std::min(hint, 8 + hint / 4) / (REPB / 4) : 0) + 8));
    // small batches: the pair kernel (two waves per 64 envs) while all of its waves are resident at once, two per SIMD at most
    bool pair = false;
    if (!RT && c->pair != 0 && !(flags & PBRE_F_NO_OBJECT) && !(MODE & FastD::M_INNER))
        pair = c->pair == 1 || 2 * blocks <= 2 * c->n_simd;
    // (env.step() under joint control, and the settle steps of reset(): 201 launches per reset)
    if constexpr (NB <= 2 && (MODE == MODE_STEP || MODE == 0)) {
        // the whole step as one launch on the caller's stream (k_fused): no fork / join through the side stream
        if (c->fused && rows) {
            const bool timed = (c->launches++ % c->ksample) == c->ksample - 1;
            hipEvent_t* ek = c->ev_k[c->k_steps % pbre_ctx::KRING];
            if (timed) (void)hipEventRecord(ek[0], s);
            c->launches_fused++;
            FusedArgs fa = {c->dT, Pk, b.state, act, out, n, c->act_dim, c->ow, flags, b.cls + (size_t)cur * b.cap, b.cls + (size_t)nxt * b.cap, b.list[cur],
                            b.count + cc * NB, b.list[nxt], b.count + cn * NB, b.cap, b.tgt, b.count + cz * NB, b.h_total, b.cap, b.count + 3 * NB, rblocks, b.objv_g, b.pair_g, 0};
            if (pair) {
                c->launches_pair++;
                if constexpr (!RT)      // (`pair` is never set with the residual exit)
                hipLaunchKernelGGL((k_fused<MODE, true>), dim3(rblocks + (blocks + FUSED_WAVES / 2 - 1) / (FUSED_WAVES / 2)), dim3(RTPB), 0, s, fa, (const Tables*)c->dT);
            } else {
                // (never 0: that value selects the block barrier, PBRE_OBJV_SYNC.  After 2^31 - 1 launches -- four days of stepping -- the numbers
                // start over, behind a clear of the records: a record last written 2^31 launches ago must not look complete)
                if (b.objv_seq == 0x7fffffff) {
                    if ((e = hipMemsetAsync(b.objv_g, 0, (size_t)(b.cap + 32) * W * sizeof(float), s)) != hipSuccess) return e;
                    if ((e = hipMemsetAsync(b.pair_g, 0, (size_t)((b.cap + FTPB - 1) / FTPB) * PBRE_PAIR_G_BYTES, s)) != hipSuccess) return e;
                    b.objv_seq = 0;
                }
                b.objv_seq++;
                fa.P.objv_seq = b.objv_seq;
                // tail pairs: as many of the last chunks as the row blocks' waves (the hint's complex envs: the coupled ones -- about a quarter,
                // generously -- one per wave, the others four, an object wave per 12) will keep k_fast waves out of the first round
                int ntail = 0;
                if (!RT && c->tail_pair != 0 && c->cfg.action_repeat <= 1 && !(flags & PBRE_F_NO_OBJECT) && b.pair_g) {
                    const int coupled = NB > 1 ? std::min(hint, 8 + hint / 4) : 0;
                    const int rw = coupled + (hint - coupled + 3) / 4 + (hint + REPB - 1) / REPB;
                    ntail = c->tail_pair > 1 ? c->tail_pair : (hint > 0 ? blocks + rw - 2 * c->n_simd : 0);
                    ntail = std::max(0, std::min(ntail, blocks - 1));
                }
                fa.ntail = ntail;
                if (ntail) c->launches_tail++;
                bool launched = false;
                if constexpr (!RT) if (ntail) {
                    hipLaunchKernelGGL((k_fused<MODE, false, false, true>), dim3(FUSED_WAVES * rblocks + blocks + ntail), dim3(FTPB), 0, s, fa, (const Tables*)c->dT);
                    launched = true;
                }
                if (!launched) hipLaunchKernelGGL((k_fused<MODE, false, RT>), dim3(FUSED_WAVES * rblocks + blocks), dim3(FTPB), 0, s, fa, (const Tables*)c->dT);
            }
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if (timed) { (void)hipEventRecord(ek[1], s); c->k_steps++; }
            b.ccur = cn;
            b.cur = nxt;
            return hipSuccess;
        }
    }
    c->side = c->sp.pick(s);
    const bool rc_first = hint >= c->rc_first_min;
    const bool single = c->idle_single && hint == 0 && b.h_total[1] == 0;      // both kernels in order on the caller's stream
    hipStream_t s_rc = (rc_first || single) ? s : c->side, s_fast = (rc_first && !single) ? c->side : s;
    const bool touch_side = single && c->idle_touch > 0 && (c->launches % c->idle_touch) == 0;   // keep the side stream's queue mapped
    if (!single || touch_side) {
        if ((e = hipEventRecord(c->ev_fork, s)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(c->side, c->ev_fork, 0)) != hipSuccess) return e;
    }
    if constexpr (NB <= 2) {
        if (rows) {
            hipLaunchKernelGGL((k_row_list<MODE, RT>), dim3(rblocks), dim3(RTPB), 0, s_rc, c->dT, Pk, b.state, act, out, c->act_dim, c->ow, flags,
                               b.list[cur], b.count + cc * NB, b.cls + (size_t)nxt * b.cap, b.list[nxt], b.count + cn * NB, b.cap, b.tgt, b.h_total, b.cap, b.count + 3 * NB);
        }
    }
    if (!rows)
        hipLaunchKernelGGL((k_fast_rc<MODE, RT>), dim3(std::min(c->n_simd, blocks + NB)), dim3(FTPB), 0, s_rc, c->dT, Pk, b.state, act, out, c->act_dim, c->ow, flags,
                           b.list[cur], b.count + cc * NB, b.cls + (size_t)nxt * b.cap, b.list[nxt], b.count + cn * NB, b.cap, b.tgt, b.h_total, b.count + 3 * NB);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // HIP event pair around the dominant kernel on the stream it runs on, for pbre_timing[3]; sampled (every KSAMPLE-th step):
    // an event record is a barrier packet the next dispatch waits for, a pair per step costs ~10% of this kernel
    // (the last launch of every group of KSAMPLE: the first launches after a reset -- cold instruction cache, first touch of the
    // state -- are warm-up, not samples)
    const bool timed = (c->launches++ % c->ksample) == c->ksample - 1;
    hipEvent_t* ek = c->ev_k[c->k_steps % pbre_ctx::KRING];
    if (timed) (void)hipEventRecord(ek[0], s_fast);
    // the 3-waves-per-SIMD variant when the complex envs' waves would push k_fast waves of the 2-wave variant into an extra round
    bool fast3 = false;
    if (!single && c->fast3 != 0 && !RT) {      // (RT: the sweep loop of the residual-exit variant needs ~170 live registers -- at 168 it spills inside the loop)
        const int slots2 = 2 * c->n_simd, rw = (rows ? (hint + 3) / 4 + (hint + REPB - 1) / REPB : (hint + FTPB - 1) / FTPB * 2) + 8;
        fast3 = c->fast3 == 1 || ((blocks + rw + slots2 - 1) / slots2 > (blocks + slots2 - 1) / slots2 && blocks + rw <= 3 * c->n_simd);
    }
    if (pair) {
        c->launches_pair++;
        if constexpr (!(MODE & FastD::M_INNER) && !RT)      // (the inner iterations of action_repeat > 1 stay on k_fast: not instantiated; RT: one lane per env)
        hipLaunchKernelGGL((k_fast_pair<MODE>), dim3(blocks), dim3(PTPB), 0, s_fast, c->dT, Pk, b.state, act, out, n, c->act_dim, c->ow, flags,
                           b.cls + (size_t)cur * b.cap, b.cls + (size_t)nxt * b.cap, b.list[nxt], b.count + cn * NB, b.cap, b.tgt, b.count + cz * NB);
    } else {
    if (fast3) c->launches3++;
    if constexpr (!RT) if (fast3)
        hipLaunchKernelGGL((k_fast<MODE, 3, RT>), dim3(blocks), dim3(FTPB), 0, s_fast, c->dT, Pk, b.state, act, out, n, c->act_dim, c->ow, flags,
                           b.cls + (size_t)cur * b.cap, b.cls + (size_t)nxt * b.cap, b.list[nxt], b.count + cn * NB, b.cap, b.tgt, b.count + cz * NB);
    if (!fast3)
        hipLaunchKernelGGL((k_fast<MODE, 2, RT>), dim3(blocks), dim3(FTPB), 0, s_fast, c->dT, Pk, b.state, act, out, n, c->act_dim, c->ow, flags,
                           b.cls + (size_t)cur * b.cap, b.cls + (size_t)nxt * b.cap, b.list[nxt], b.count + cn * NB, b.cap, b.tgt, b.count + cz * NB);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (timed) { (void)hipEventRecord(ek[1], s_fast); c->k_steps++; }
    if (!single || touch_side) {
        // (round-2 advice) the side stream's kernel may have written rows into host memory: join through the event that keeps its
        // system-scope release, so the host sees them once the caller's stream is synchronised, whatever the buffer's coherence mode
        hipEvent_t ej = c->rows_to_host ? c->ev_join_sys : c->ev_join;
        if ((e = hipEventRecord(ej, c->side)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(s, ej, 0)) != hipSuccess) return e;
    }
    b.ccur = cn;
    b.cur = nxt;
    return hipSuccess;
}


constexpr int PBRE_STEP_MODES[6] = {0, MODE_SETTLE_IK, MODE_STEP, MODE_STEP_IK, MODE_INNER, MODE_INNER_IK};      // what launch_step<MODE> is called with
#define PBRE_STEP_INST(KW, M, RT) KW template hipError_t launch_step_t<PBRE_STEP_MODES[M], RT>(pbre_ctx*, EnvBuf&, int, const float*, float*, int, hipStream_t);
#define PBRE_STEP_INST_ALL(KW) PBRE_STEP_INST(KW, 0, false) PBRE_STEP_INST(KW, 1, false) PBRE_STEP_INST(KW, 2, false) PBRE_STEP_INST(KW, 3, false) \
    PBRE_STEP_INST(KW, 4, false) PBRE_STEP_INST(KW, 5, false) PBRE_STEP_INST(KW, 0, true) PBRE_STEP_INST(KW, 1, true) PBRE_STEP_INST(KW, 2, true) \
    PBRE_STEP_INST(KW, 3, true) PBRE_STEP_INST(KW, 4, true) PBRE_STEP_INST(KW, 5, true)
