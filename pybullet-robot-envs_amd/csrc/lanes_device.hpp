// lanes_device.hpp -- gfx950 (CDNA4) lane backend for pbre_core.hpp.
//
// One env = one 16-lane DPP row; a wave64 carries 4 envs.  Row all-reduces are 4 DPP
// steps (quad_perm xor1, quad_perm xor2, row_half_mirror, row_mirror) that the compiler
// folds into v_add_f32_dpp; broadcasts/gathers inside a row are ds_bpermute_b32 (LDS
// crossbar, no LDS memory).  Nothing here touches another row, so a wave never needs a
// barrier and a workgroup never needs LDS.
#pragma once
#include <hip/hip_runtime.h>
#ifndef PBRE_HD
#define PBRE_HD __device__ __forceinline__
#endif
#include "pbre_math.hpp"

namespace pbre {

struct DevLanes {
    using F = float; using I = int; using B = bool;
    using Robot = DevLanes;                                       // robot-lane view (pbre_core.hpp): the backend itself
    static __device__ __forceinline__ F lo(F x) { return x; }
    static __device__ __forceinline__ F wide(F x) { return x; }
    static __device__ __forceinline__ F fma_lo(F s, F m, F acc) { return __builtin_fmaf(s, m, acc); }
    static __device__ __forceinline__ F uni(F x) { return x; }     // a group-uniform value
    // lane j of every group <- src.  Written as one compare + select pair in asm: as plain C++ the compiler hoists the 2 x n_dof
    // loop-invariant lane masks of the solver's row sweeps into SGPR pairs, runs out of SGPRs, and reloads every mask from a spill
    // VGPR with two v_readlane per row.
    static __device__ __forceinline__ F setlane_l(F x, int j, F src, int lane_) {
        asm("v_cmp_eq_u32_e32 vcc, %2, %1\n\tv_cndmask_b32_e32 %0, %0, %3, vcc" : "+v"(x) : "v"(lane_), "n"(j), "v"(src) : "vcc");
        return x;
    }
    static __device__ __forceinline__ F setlane(F x, int j, F src) { return setlane_l(x, j, src, (int)(threadIdx.x & 15u)); }
    // The tail of a one-DoF row (motor / limit row of joint J) as ONE block of three instructions:
    //     rec lane J <- recval          (v_cmp_eq_u32 + v_cndmask_b32, as setlane)
    //     acc += (lane J of bval) * m   (v_fmac_f32 with a DPP row_newbcast source operand: no v_mov_b32_dpp, no register for the broadcast)
    // Round 6 measured that a lone wave issues one instruction per ~4.5 cycles whether or not it depends on its predecessor
    // (profiles/r06_ubench_valu.txt): every instruction of the row waves' sweeps is on the step's critical path, the ones "off the chain"
    // too.  The compiler does not fold update_dpp into the fma (VOP3 at that point), hence the asm.  The DPP operand needs two wait
    // states behind the VALU write of bval: the compare and the select are those two (one block: the order is fixed).
    template <int J>
    static __device__ __forceinline__ F row_tail(F& rec, F recval, F bval, F m, F acc) {
        asm("v_cmp_eq_u32_e32 vcc, %3, %2\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_fmac_f32_dpp %1, %5, %6 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
            : "+v"(rec), "+v"(acc) : "v"((int)(threadIdx.x & 15u)), "n"(J), "v"(recval), "v"(bval), "v"(m) : "vcc");
        return acc;
    }
    static __device__ __forceinline__ unsigned long long lanebits(B b) {      // bit j: b holds on lane j of some group of the wave
        unsigned long long m = __ballot((int)b);
        m |= m >> 32; m |= m >> 16;
        return m & 0xFFFFull;
    }
    template <int N> struct RowStore {                            // contact rows of the solver: registers
        F v[N];
        __device__ __forceinline__ void init() {}
        __device__ __forceinline__ F get(int i) const { return v[i]; }
        __device__ __forceinline__ void put(int i, F x) { v[i] = x; }
    };
    static __device__ __forceinline__ F c(float x) { return x; }
    static __device__ __forceinline__ I ci(int x) { return x; }
    static __device__ __forceinline__ I lane() { return (int)(threadIdx.x & 15u); }
    static __device__ __forceinline__ F load(const float* p) { return p[threadIdx.x & 15u]; }
    static __device__ __forceinline__ I loadI(const int* p) { return p[threadIdx.x & 15u]; }
    static __device__ __forceinline__ F loadm(const float* p, B m) { return m ? p[threadIdx.x & 15u] : 0.f; }
    static __device__ __forceinline__ F loadu(const float* p) { return *p; }                       // group-uniform address
    static __device__ __forceinline__ float first(F x) { return x; }                               // a group-uniform value as a scalar
    static __device__ __forceinline__ bool lane0() { return (threadIdx.x & 15u) == 0; }
    static __device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }    // a lane reads what another lane of the group stored
    static __device__ __forceinline__ F loadx(const float* p, I idx, B m) { return m ? p[idx] : 0.f; }
    static __device__ __forceinline__ void storex(float* p, I idx, F x, B m) { if (m) p[idx] = x; }
    static __device__ __forceinline__ void store(float* p, F x) { p[threadIdx.x & 15u] = x; }
    static __device__ __forceinline__ void storem(float* p, F x, B m) { if (m) p[threadIdx.x & 15u] = x; }
    static __device__ __forceinline__ F abs(F x) { return __builtin_fabsf(x); }
    static __device__ __forceinline__ F sqrt(F x) { return sqrtf(x); }
    static __device__ __forceinline__ F sin(F x) { return sinf(x); }
    static __device__ __forceinline__ F cos(F x) { return cosf(x); }
    static __device__ __forceinline__ F asin(F x) { return asinf(x); }
    static __device__ __forceinline__ void sincos(F x, F& s, F& c) { pbre::sincos_f(x, s, c); }      // shared range reduction (pbre_math.hpp)
    static __device__ __forceinline__ F atan2(F a, F b) { return atan2f(a, b); }
    static __device__ __forceinline__ F fma(F a, F b, F c_) { return __builtin_fmaf(a, b, c_); }
    static __device__ __forceinline__ F min(F a, F b) { return __builtin_fminf(a, b); }
    static __device__ __forceinline__ F max(F a, F b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ F med3(F x, F lo, F hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
    static __device__ __forceinline__ B lt(F a, F b) { return a < b; }
    static __device__ __forceinline__ B le(F a, F b) { return a <= b; }
    static __device__ __forceinline__ B gt(F a, F b) { return a > b; }
    static __device__ __forceinline__ B ge(F a, F b) { return a >= b; }
    static __device__ __forceinline__ B eq(F a, F b) { return a == b; }
    static __device__ __forceinline__ B ne(F a, F b) { return a != b; }
    static __device__ __forceinline__ B eqi(I a, I b) { return a == b; }
    static __device__ __forceinline__ B nei(I a, I b) { return a != b; }
    static __device__ __forceinline__ B lti(I a, I b) { return a < b; }
    static __device__ __forceinline__ B gei(I a, I b) { return a >= b; }
    static __device__ __forceinline__ B band(B a, B b) { return a & b; }
    static __device__ __forceinline__ B bor(B a, B b) { return a | b; }
    static __device__ __forceinline__ B bnot(B a) { return !a; }
    static __device__ __forceinline__ B bfalse() { return false; }
    static __device__ __forceinline__ bool any(B a) { return __any((int)a) != 0; }   // wave-uniform
    static __device__ __forceinline__ F sel(B m, F a, F b) { return m ? a : b; }
    static __device__ __forceinline__ I seli(B m, I a, I b) { return m ? a : b; }
    static __device__ __forceinline__ B bit(I m, int k) { return (m >> k) & 1; }
    static __device__ __forceinline__ B biti(I m, I k) { return (m >> (k & 31)) & 1; }
    static __device__ __forceinline__ I maxi(I a, int b) { return a > b ? a : b; }
    static __device__ __forceinline__ F itof(I a) { return (float)a; }
    static __device__ __forceinline__ I ftoi(F a) { return (int)a; }

    // ---- cross-lane, confined to the 16-lane row
    static __device__ __forceinline__ int row_base() { return (int)(threadIdx.x & 48u); }
    static __device__ __forceinline__ F gather(F a, I idx) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute((row_base() | (idx & 15)) << 2, __float_as_int(a)));
    }
    static __device__ __forceinline__ I gatherI(I a, I idx) {
        return __builtin_amdgcn_ds_bpermute((row_base() | (idx & 15)) << 2, a);
    }
    // Broadcast of lane k of every row: one DPP move with row_newbcast:k (gfx90a+) instead of a ds_bpermute_b32 round trip through the
    // LDS crossbar.  k is a constant wherever the solver calls this (unrolled row loops), so the switch folds to one instruction.
    template <int K>
    static __device__ __forceinline__ F row_bc(F x) {
        // (old = the source itself: row_newbcast writes every lane, so no register has to be zeroed first)
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), 0x150 + K, 0xF, 0xF, false));
    }
    static __device__ __forceinline__ F bcast16(F a, int k) {
        switch (k & 15) {
            case 0: return row_bc<0>(a);   case 1: return row_bc<1>(a);   case 2: return row_bc<2>(a);   case 3: return row_bc<3>(a);
            case 4: return row_bc<4>(a);   case 5: return row_bc<5>(a);   case 6: return row_bc<6>(a);   case 7: return row_bc<7>(a);
            case 8: return row_bc<8>(a);   case 9: return row_bc<9>(a);   case 10: return row_bc<10>(a); case 11: return row_bc<11>(a);
            case 12: return row_bc<12>(a); case 13: return row_bc<13>(a); case 14: return row_bc<14>(a); default: return row_bc<15>(a);
        }
    }
    static __device__ __forceinline__ F bcast(F a, int k) { return bcast16(a, k); }
    static __device__ __forceinline__ F bcast_row(F a, int k) { return bcast16(a, k); }

    template <int CTRL>
    static __device__ __forceinline__ F dpp(F x) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
    }
    static __device__ __forceinline__ F sum(F x) {
        x += dpp<0xB1>(x);     // quad_perm [1,0,3,2]
        x += dpp<0x4E>(x);     // quad_perm [2,3,0,1]
        x += dpp<0x141>(x);    // row_half_mirror
        x += dpp<0x140>(x);    // row_mirror
        return x;
    }
    // one stage of sum() (k = 0..3): the solver's two-chain sweeps issue the stages of two rows in turn (pbre_core.hpp)
    static __device__ __forceinline__ F sum_step(F x, int k) {
        switch (k) {
            case 0: return x + dpp<0xB1>(x);
            case 1: return x + dpp<0x4E>(x);
            case 2: return x + dpp<0x141>(x);
            default: return x + dpp<0x140>(x);
        }
    }
    static __device__ __forceinline__ F vmin(F x) {
        x = __builtin_fminf(x, dpp<0xB1>(x));
        x = __builtin_fminf(x, dpp<0x4E>(x));
        x = __builtin_fminf(x, dpp<0x141>(x));
        x = __builtin_fminf(x, dpp<0x140>(x));
        return x;
    }
    static __device__ __forceinline__ F sum_obj(F x) { return sum(x); }    // all-reduce of a value that is zero on the robot lanes
};

// One env = one half-wave of 32 lanes (<= 20 DoF: the iCub without its legs), 2 envs per wave64.  Broadcasts / gathers are
// ds_bpermute_b32 inside the half-wave; an all-reduce is the 16-lane DPP butterfly plus one ds_swizzle_b32 (xor 16).
struct DevLanes32 : DevLanes {
    using Robot = DevLanes32;
    // (measured slower here: the 16-lane backend's compare + select pair in asm, -1.5 %; the mask written to VCC by two s_mov and one
    // select, as in DevLanes64::setlane, -6 % although it removes the v_readlane pairs that restore the compiler's spilled SGPR masks
    // -- the half-wave solver loop is bound by its per-wave serial chain, not by VALU issue)
    static __device__ __forceinline__ F setlane(F x, int j, F src) { return (int)(threadIdx.x & 31u) == j ? src : x; }
    template <int J>
    static __device__ __forceinline__ F row_tail(F& rec, F recval, F bval, F m, F acc) { rec = setlane(rec, J, recval); return __builtin_fmaf(bcast(bval, J), m, acc); }
    static __device__ __forceinline__ unsigned long long lanebits(B b) { unsigned long long m = __ballot((int)b); return (m | (m >> 32)) & 0xFFFFFFFFull; }
    static __device__ __forceinline__ I lane() { return (int)(threadIdx.x & 31u); }
    static __device__ __forceinline__ bool lane0() { return (threadIdx.x & 31u) == 0; }
    static __device__ __forceinline__ F load(const float* p) { return p[threadIdx.x & 31u]; }
    static __device__ __forceinline__ I loadI(const int* p) { return p[threadIdx.x & 31u]; }
    static __device__ __forceinline__ F loadm(const float* p, B m) { return m ? p[threadIdx.x & 31u] : 0.f; }
    static __device__ __forceinline__ void store(float* p, F x) { p[threadIdx.x & 31u] = x; }
    static __device__ __forceinline__ void storem(float* p, F x, B m) { if (m) p[threadIdx.x & 31u] = x; }
    static __device__ __forceinline__ int half_base() { return (int)(threadIdx.x & 32u); }
    static __device__ __forceinline__ F gather(F a, I idx) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute((half_base() | (idx & 31)) << 2, __float_as_int(a)));
    }
    static __device__ __forceinline__ I gatherI(I a, I idx) { return __builtin_amdgcn_ds_bpermute((half_base() | (idx & 31)) << 2, a); }
    static __device__ __forceinline__ F bcast(F a, int k) { return gather(a, k); }
    // Broadcast of lane k of each half without the LDS round trip: two v_readlane and a select.  Four vector instructions more than
    // ds_bpermute_b32, a much shorter wait: for the one place where the row-to-row dependency chain is all that is left (the clamp-free
    // motor rows of Core::step, vector pipe half idle); everywhere else the solver loops are issue-bound and bcast() is the cheaper one.
    // (Feeding the two scalars straight into one v_fmac per half under a half EXEC mask -- no v_mov / v_cndmask on the chain --
    // measured 8 % slower with IK control: three s_mov exec per row.)
    static __device__ __forceinline__ F bcast_row(F a, int k) {
        if (k < 16) {
            // source in the lower row of each half: row_newbcast:k inside the rows, then row_bcast:15 carries lane 15 of rows 0 and 2
            // (which now hold the value) into rows 1 and 3 -- two DPP moves, no scalar round trip
            const int r = __float_as_int(bcast16(a, k));
            return __int_as_float(__builtin_amdgcn_update_dpp(r, r, 0x142, 0xA, 0xF, false));
        }
        const int ai = __float_as_int(a);
        const int lo = __builtin_amdgcn_readlane(ai, k), hi = __builtin_amdgcn_readlane(ai, k + 32);
        return __int_as_float((threadIdx.x & 32u) ? hi : lo);
    }
    // x is zero on lanes 0..15 of either half (robot joints 0..15; the object lanes 20..26 are in the upper row): the half's sum is
    // its upper row's sum, r1 + r0 with r0 = 0 exactly.  Row butterfly, then lane 16 / 48 to every lane by v_readlane + a select
    // by half -- no trip through the LDS crossbar (ds_swizzle) on the solver's serial chain; same value as sum() bit for bit.
    static __device__ __forceinline__ F sum_obj(F x) {
        x += dpp<0xB1>(x);
        x += dpp<0x4E>(x);
        x += dpp<0x141>(x);
        x += dpp<0x140>(x);
        const float lo_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
        const float hi_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
        return (threadIdx.x & 32u) ? hi_ : lo_;
    }
    static __device__ __forceinline__ F swap16(F x) {     // lane i <-> lane i ^ 16 inside each 32-lane half (bit-mode swizzle: and 0x1F, or 0, xor 0x10)
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));
    }
    static __device__ __forceinline__ F sum(F x) {
        x += dpp<0xB1>(x);
        x += dpp<0x4E>(x);
        x += dpp<0x141>(x);
        x += dpp<0x140>(x);              // every lane: sum of its 16-lane row
        return x + swap16(x);            // + the other row of the half
    }
    static __device__ __forceinline__ F vmin(F x) {
        x = __builtin_fminf(x, dpp<0xB1>(x));
        x = __builtin_fminf(x, dpp<0x4E>(x));
        x = __builtin_fminf(x, dpp<0x141>(x));
        x = __builtin_fminf(x, dpp<0x140>(x));
        return __builtin_fminf(x, swap16(x));
    }
};

// One env = one whole wave64 (<= 32 DoF).  Broadcasts of a compile-time lane are v_readlane_b32 (the value becomes
// an SGPR operand), gathers are ds_bpermute_b32 over the wave, all-reduces are the 16-lane DPP butterfly followed by
// row_bcast15 / row_bcast31 and a v_readlane of lane 63 (summation order (r3 + r2) + (r1 + r0), mirrored by the host
// emulation so that CPU tests and device agree bit for bit).
struct DevLanes64 : DevLanes {
    using Robot = DevLanes64;
    static __device__ __forceinline__ unsigned long long lanebits(B b) { return __ballot((int)b); }
    static __device__ __forceinline__ F setlane(F x, int j, F src) {      // constant one-lane mask written to VCC by the scalar unit + one select
        asm("s_mov_b32 vcc_lo, %2\n\ts_mov_b32 vcc_hi, %3\n\tv_cndmask_b32_e32 %0, %0, %1, vcc"       // (j is a constant once the row loops are unrolled)
            : "+v"(x) : "v"(src), "n"(j < 32 ? 1u << (j & 31) : 0u), "n"(j >= 32 ? 1u << (j & 31) : 0u) : "vcc");
        return x;
    }
    template <int J>
    static __device__ __forceinline__ F row_tail(F& rec, F recval, F bval, F m, F acc) { rec = setlane(rec, J, recval); return __builtin_fmaf(bcast(bval, J), m, acc); }
    static __device__ __forceinline__ I lane() { return (int)(threadIdx.x & 63u); }
    static __device__ __forceinline__ bool lane0() { return (threadIdx.x & 63u) == 0; }
    static __device__ __forceinline__ F load(const float* p) { return p[threadIdx.x & 63u]; }
    static __device__ __forceinline__ I loadI(const int* p) { return p[threadIdx.x & 63u]; }
    static __device__ __forceinline__ F loadm(const float* p, B m) { return m ? p[threadIdx.x & 63u] : 0.f; }
    static __device__ __forceinline__ void store(float* p, F x) { p[threadIdx.x & 63u] = x; }
    static __device__ __forceinline__ void storem(float* p, F x, B m) { if (m) p[threadIdx.x & 63u] = x; }
    static __device__ __forceinline__ F gather(F a, I idx) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute((idx & 63) << 2, __float_as_int(a)));
    }
    static __device__ __forceinline__ I gatherI(I a, I idx) { return __builtin_amdgcn_ds_bpermute((idx & 63) << 2, a); }
    static __device__ __forceinline__ F bcast(F a, int k) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), __builtin_amdgcn_readfirstlane(k)));
    }
    static __device__ __forceinline__ F bcast_row(F a, int k) { return bcast(a, k); }
    template <int CTRL, int ROWS>
    static __device__ __forceinline__ F dppr(F x) {      // rows outside ROWS receive 0
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROWS, 0xF, false));
    }
    static __device__ __forceinline__ F sum(F x) {
        x += dpp<0xB1>(x);
        x += dpp<0x4E>(x);
        x += dpp<0x141>(x);
        x += dpp<0x140>(x);              // every lane: sum of its 16-lane row
        x += dppr<0x142, 0xA>(x);        // row_bcast15 into rows 1, 3: r1 + r0, r3 + r2
        x += dppr<0x143, 0xC>(x);        // row_bcast31 into rows 2, 3: lane 63 = (r3 + r2) + (r1 + r0)
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    }
    static __device__ __forceinline__ F vmin(F x) {
        x = __builtin_fminf(x, dpp<0xB1>(x));
        x = __builtin_fminf(x, dpp<0x4E>(x));
        x = __builtin_fminf(x, dpp<0x141>(x));
        x = __builtin_fminf(x, dpp<0x140>(x));
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 16));
        const float c_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 48));
        return __builtin_fminf(__builtin_fminf(a, b), __builtin_fminf(c_, d));
    }
};

// One env = one wave64 carrying 128 virtual lanes (<= 60 DoF + object + constant lane: the iCub with hands): physical
// lane t holds virtual lanes t (.a) and t + 64 (.b).  Arithmetic is done on both halves; an all-reduce adds the halves and
// runs the 64-lane butterfly once; a broadcast of a compile-time lane is one v_readlane of the half that holds it; a gather
// is two ds_bpermute per source half.  Robot-lane-only data (rows of M^-1, motor / limit rows) live on the low half alone
// (Robot = DevLanes64), which keeps the 60 x 60 inverse in 60 VGPRs.
struct F2 { float a, b; };
struct I2 { int a, b; __device__ __forceinline__ I2() {} __device__ __forceinline__ I2(int x) : a(x), b(x) {} __device__ __forceinline__ I2(int x, int y) : a(x), b(y) {} };
struct B2 { bool a, b; };
static __device__ __forceinline__ F2 operator+(F2 x, F2 y) { return F2{x.a + y.a, x.b + y.b}; }
static __device__ __forceinline__ F2 operator-(F2 x, F2 y) { return F2{x.a - y.a, x.b - y.b}; }
static __device__ __forceinline__ F2 operator*(F2 x, F2 y) { return F2{x.a * y.a, x.b * y.b}; }
static __device__ __forceinline__ F2 operator/(F2 x, F2 y) { return F2{x.a / y.a, x.b / y.b}; }

struct DevLanes128 {
    using F = F2; using I = I2; using B = B2;
    using Robot = DevLanes64;
    using D = DevLanes64;
    static __device__ __forceinline__ float lo(F x) { return x.a; }
    static __device__ __forceinline__ F wide(float r) { return F{r, 0.f}; }
    static __device__ __forceinline__ F fma_lo(float s, float m, F acc) { return F{__builtin_fmaf(s, m, acc.a), acc.b}; }
    static __device__ __forceinline__ F uni(F x) { return F{x.a, x.a}; }   // group-uniform: both halves are the same value -- one register
    static __device__ __forceinline__ int t() { return (int)(threadIdx.x & 63u); }
    // Contact rows of the solver (72 rows x 128 virtual lanes) do not fit the register file next to the 60 rows of M^-1: they
    // live in LDS, one private region per wave (no barriers: a wave only ever touches its own region, and its LDS accesses
    // are ordered).  The low halves are lane-contiguous (conflict-free); of a row's high half only virtual lanes 64..66 are
    // ever non-zero (object angular y, z and the constant lane), so it is stored as 4 floats: lanes 0..2 own one each, every
    // other lane reads the zero in slot 3 (an LDS broadcast).  19.6 KB per wave: two 4-wave blocks fit a CU's 160 KB.
    static constexpr int WPB = 4;                                 // waves per block of the kernels that use this backend
    typedef __attribute__((address_space(3))) float lds_float;
    template <int N> struct RowStore {
        lds_float *lo, *hi;
        __device__ __forceinline__ void init() {
            __shared__ float lds[WPB * N * 68];
            lds_float* w = (lds_float*)lds + (threadIdx.x >> 6) * (N * 68);
            const int t_ = (int)(threadIdx.x & 63u);
            lo = w + t_;
            hi = w + N * 64 + (t_ < 3 ? t_ : 3);
        }
        // the rows are loop-invariant inside the solver loop; the empty asm makes every read's address opaque so that the
        // compiler reads LDS where the row is used instead of hoisting 144 row registers out of the loop (and spilling them)
        __device__ __forceinline__ F get(int i) const {
            lds_float *a = lo, *b = hi;
            asm volatile("" : "+v"(a), "+v"(b));
            return F{a[i * 64], b[i * 4]};
        }
        __device__ __forceinline__ void put(int i, F x) {
            lo[i * 64] = x.a;
            const int t_ = (int)(threadIdx.x & 63u);
            if (t_ < 4) hi[i * 4] = t_ < 3 ? x.b : 0.f;
        }
    };
    static __device__ __forceinline__ F c(float x) { return F{x, x}; }
    static __device__ __forceinline__ I ci(int x) { return I(x); }
    static __device__ __forceinline__ I lane() { return I(t(), t() + 64); }
    static __device__ __forceinline__ F load(const float* p) { return F{p[t()], p[t() + 64]}; }
    static __device__ __forceinline__ I loadI(const int* p) { return I(p[t()], p[t() + 64]); }
    static __device__ __forceinline__ F loadm(const float* p, B m) { return F{m.a ? p[t()] : 0.f, m.b ? p[t() + 64] : 0.f}; }
    static __device__ __forceinline__ F loadu(const float* p) { const float v = *p; return F{v, v}; }
    static __device__ __forceinline__ float first(F x) { return x.a; }
    static __device__ __forceinline__ bool lane0() { return t() == 0; }
    static __device__ __forceinline__ void fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
    static __device__ __forceinline__ F loadx(const float* p, I idx, B m) { return F{m.a ? p[idx.a] : 0.f, m.b ? p[idx.b] : 0.f}; }
    static __device__ __forceinline__ void storex(float* p, I idx, F x, B m) { if (m.a) p[idx.a] = x.a; if (m.b) p[idx.b] = x.b; }
    static __device__ __forceinline__ void store(float* p, F x) { p[t()] = x.a; p[t() + 64] = x.b; }
    static __device__ __forceinline__ void storem(float* p, F x, B m) { if (m.a) p[t()] = x.a; if (m.b) p[t() + 64] = x.b; }
#define PBRE_U1(name, fn) static __device__ __forceinline__ F name(F x) { return F{fn(x.a), fn(x.b)}; }
    PBRE_U1(abs, __builtin_fabsf) PBRE_U1(sqrt, sqrtf) PBRE_U1(sin, sinf) PBRE_U1(cos, cosf) PBRE_U1(asin, asinf)
#undef PBRE_U1
    static __device__ __forceinline__ void sincos(F x, F& s, F& c) { pbre::sincos_f(x.a, s.a, c.a); pbre::sincos_f(x.b, s.b, c.b); }
    static __device__ __forceinline__ F atan2(F x, F y) { return F{atan2f(x.a, y.a), atan2f(x.b, y.b)}; }
    static __device__ __forceinline__ F fma(F x, F y, F z) { return F{__builtin_fmaf(x.a, y.a, z.a), __builtin_fmaf(x.b, y.b, z.b)}; }
    static __device__ __forceinline__ F min(F x, F y) { return F{__builtin_fminf(x.a, y.a), __builtin_fminf(x.b, y.b)}; }
    static __device__ __forceinline__ F max(F x, F y) { return F{__builtin_fmaxf(x.a, y.a), __builtin_fmaxf(x.b, y.b)}; }
    static __device__ __forceinline__ F med3(F x, F l, F h) { return F{__builtin_amdgcn_fmed3f(x.a, l.a, h.a), __builtin_amdgcn_fmed3f(x.b, l.b, h.b)}; }
#define PBRE_CMP(name, op) static __device__ __forceinline__ B name(F x, F y) { return B{x.a op y.a, x.b op y.b}; }
    PBRE_CMP(lt, <) PBRE_CMP(le, <=) PBRE_CMP(gt, >) PBRE_CMP(ge, >=) PBRE_CMP(eq, ==) PBRE_CMP(ne, !=)
#undef PBRE_CMP
#define PBRE_CMPI(name, op) static __device__ __forceinline__ B name(I x, I y) { return B{x.a op y.a, x.b op y.b}; }
    PBRE_CMPI(eqi, ==) PBRE_CMPI(nei, !=) PBRE_CMPI(lti, <) PBRE_CMPI(gei, >=)
#undef PBRE_CMPI
    static __device__ __forceinline__ B band(B x, B y) { return B{(bool)(x.a & y.a), (bool)(x.b & y.b)}; }
    static __device__ __forceinline__ B bor(B x, B y) { return B{(bool)(x.a | y.a), (bool)(x.b | y.b)}; }
    static __device__ __forceinline__ B bnot(B x) { return B{!x.a, !x.b}; }
    static __device__ __forceinline__ B bfalse() { return B{false, false}; }
    static __device__ __forceinline__ bool any(B x) { return __any((int)(x.a | x.b)) != 0; }
    static __device__ __forceinline__ F sel(B m, F x, F y) { return F{m.a ? x.a : y.a, m.b ? x.b : y.b}; }
    static __device__ __forceinline__ I seli(B m, I x, I y) { return I(m.a ? x.a : y.a, m.b ? x.b : y.b); }
    static __device__ __forceinline__ B bit(I m, int k) { return B{(bool)((m.a >> k) & 1), (bool)((m.b >> k) & 1)}; }
    static __device__ __forceinline__ B biti(I m, I k) { return B{(bool)((m.a >> (k.a & 31)) & 1), (bool)((m.b >> (k.b & 31)) & 1)}; }
    static __device__ __forceinline__ I maxi(I x, int y) { return I(x.a > y ? x.a : y, x.b > y ? x.b : y); }
    static __device__ __forceinline__ F itof(I x) { return F{(float)x.a, (float)x.b}; }
    static __device__ __forceinline__ I ftoi(F x) { return I((int)x.a, (int)x.b); }
    // ---- cross-lane over the 128 virtual lanes
    static __device__ __forceinline__ int g1(int lo_, int hi_, int idx) {
        const int v0 = __builtin_amdgcn_ds_bpermute((idx & 63) << 2, lo_), v1 = __builtin_amdgcn_ds_bpermute((idx & 63) << 2, hi_);
        return (idx & 64) ? v1 : v0;
    }
    static __device__ __forceinline__ F gather(F x, I idx) {
        return F{__int_as_float(g1(__float_as_int(x.a), __float_as_int(x.b), idx.a)), __int_as_float(g1(__float_as_int(x.a), __float_as_int(x.b), idx.b))};
    }
    static __device__ __forceinline__ I gatherI(I x, I idx) { return I(g1(x.a, x.b, idx.a), g1(x.a, x.b, idx.b)); }
    static __device__ __forceinline__ F bcast(F x, int k) {
        const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k < 64 ? x.a : x.b), __builtin_amdgcn_readfirstlane(k & 63)));
        return F{v, v};
    }
    static __device__ __forceinline__ F sum(F x) { const float s = D::sum(x.a + x.b); return F{s, s}; }
    static __device__ __forceinline__ F sum_obj(F x) { return sum(x); }
    static __device__ __forceinline__ F vmin(F x) { const float s = D::vmin(__builtin_fminf(x.a, x.b)); return F{s, s}; }
};

}  // namespace pbre
