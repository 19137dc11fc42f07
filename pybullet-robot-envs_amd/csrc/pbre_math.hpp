// pbre_math.hpp -- scalar math helpers shared by every kernel source (device and the host emulation builds).
#pragma once
#include <math.h>
#ifndef PBRE_HD
#define PBRE_HD
#endif

// Floating-point contraction is stated by every header that cares, at its top, and never inherited from whatever was included before
// ('#pragma float_control(push / pop)' is ignored by the amdgcn target, so there is no "restore"): pbre_core.hpp compiles with contraction
// off (its fused operations are explicit L::fma: results must not depend on which solver path a wave takes), the lane-per-env headers
// (pbre_fast.hpp, pbre_objstep.hpp, pbre_lane.hpp) with contraction fast.  pbre_core.hpp ends by switching back to fast, the setting
// every other source of csrc/ is written for.  g++ builds of the host emulation use -ffp-contract=off throughout.
#if defined(__clang__)
#define PBRE_FP_CONTRACT_OFF _Pragma("clang fp contract(off)")
#define PBRE_FP_CONTRACT_FAST _Pragma("clang fp contract(fast)")
#else
#define PBRE_FP_CONTRACT_OFF
#define PBRE_FP_CONTRACT_FAST
#endif

namespace pbre {

// sin and cos of one argument with one shared range reduction (Cody-Waite, pi/2 in three parts) and degree-7 / degree-8 minimax
// polynomials on [-pi/4, pi/4]: absolute error <= 1.3e-7 for |x| < 1e3 (joint angles, Euler half-angles and yaw samples are all below
// 2 pi), ~28 instructions for the pair against ~240 (with a Payne-Hanek slow path each) for separate sinf() and cosf() calls and ~130
// for sincosf().  A Panda step evaluates 23 pairs, an iCub IK iteration 10; the straight-line setup code of k_fast shrank by a fifth.
// Plain C++: the host emulation runs the same arithmetic.
static PBRE_HD void sincos_f(float x, float& sn, float& cs) {
    const float kf = rintf(x * 0.63661977236758134f);
    float r = fmaf(-kf, 1.5707855224609375f, x);          // pi/2 = 1.5707855224609375 + 1.0804334124e-5 + 6.0770999344e-11 (fdlibm's split)
    r = fmaf(-kf, 1.0804334124e-5f, r);
    r = fmaf(-kf, 6.0770999344e-11f, r);
    const float z = r * r;
    const float ps = fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f);
    const float s0 = fmaf(r * z, ps, r);                   // r + r^3 (...)
    const float pc = fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f);
    const float c0 = fmaf(z * z, pc, fmaf(-0.5f, z, 1.f));
    const int k = (int)kf;
    const bool swap = (k & 1) != 0;
    const float sv = swap ? c0 : s0, cv = swap ? s0 : c0;
    sn = (k & 2) ? -sv : sv;
    cs = ((k + 1) & 2) ? -cv : cv;
}

}  // namespace pbre
