// pbre_core.hpp -- the env.step() hot path, written once against a "lane backend" L.
//
// Mapping: one environment = one lane group of shape S (pbre_tables.hpp): a 16-lane DPP row for the
// Panda (4 envs per wave), a 32-lane half-wave for the iCub (2 envs per wave), 128 virtual lanes on one wave for the iCub
// with hands (two per physical lane).  Lane k of a group owns generalized coordinate k:
//     lanes 0..NJ-1     robot joints                         -- link/joint data of link k
//     lanes LC..LC+2    object linear velocity x,y,z         lanes LC+3..LC+5 object angular velocity
//     lane  L1          constant 1 (carries -rhs of a contact row through the row dot product)
// All solver data lives in VGPRs; cross-lane traffic is DPP (all-reduce) and ds_bpermute / v_readlane
// (broadcast/gather inside the group).  No LDS allocation, no scratch.
//
// L provides: F (float per lane), I (int per lane), B (predicate per lane) and the ops used
// below, plus Robot (the view of values that only exist on robot lanes), RowStore (where the contact rows live) and
// uni / setlane / fma_lo.  Device backends: lanes_device.hpp (F = float; DevLanes128: F = two floats).  Host backend
// (CPU tests only): tests/host_emu/lanes_host.hpp (F = W floats).  Control flow is group-uniform by
// construction; L::any() is only ever used to skip work that is a no-op when false.
//
// Replaces, per env (reference file:line):
//   apply_action          pybullet_robot_envs/envs/panda_envs/panda_push_gym_env.py:189-242
//   p.setJointMotorControl2  .../panda_env.py:293-310
//   p.stepSimulation      panda_push_gym_env.py:236  (Bullet multibody step, restated; see oracle/)
//   get_observation       panda_env.py:141-193; world_env.py:109-126
//   get_extended_observation / _termination / _compute_reward   panda_push_gym_env.py:150-187, 301-331
#pragma once
#include <math.h>
#include <type_traits>
#include <utility>
#include "pbre_tables.hpp"
#ifndef PBRE_HD
#define PBRE_HD
#endif
#include "pbre_math.hpp"

#ifndef PBRE_HD
#define PBRE_HD
#endif
#ifndef PBRE_UNROLL
#define PBRE_UNROLL
#endif
#ifndef PBRE_OPAQUE
#define PBRE_OPAQUE(p)       // device builds: hide a pointer's value from the optimiser (no instruction)
#endif
#ifndef PBRE_COUNT_BAD       // ++*p from any number of lanes (device: atomicAdd)
#define PBRE_COUNT_BAD(p) (++*(p))
#endif
#ifndef PBRE_OBJV_SYNC       // a kernel whose `objv` side record is produced by a sibling wave of the same block (k_row_list): the block barrier
#define PBRE_OBJV_SYNC(poison) // behind which it is complete; nothing where a kernel of its own produced it earlier (kw_obj) and on the host
#endif
#ifndef PBRE_PROBE           // phase timing of one wave (tools/phase_probe.py builds with -DPBRE_PHASE_PROBE); nothing otherwise
#define PBRE_PROBE(k)
#define PBRE_PROBE_DECL
#endif
#ifndef PBRE_FREE_SWITCH_FRAC        // the fraction of its bound an applied impulse may reach on the clamp-free stages (see free_far_inside)
#define PBRE_FREE_SWITCH_FRAC 0.5f
#endif
#ifndef PBRE_FREE_SWITCH             // 1: a wave on the clamp-free motor stages goes on with the clamping ones once an impulse is past PBRE_FREE_SWITCH_FRAC of its
#define PBRE_FREE_SWITCH 0           // bound, instead of starting over when one leaves it (see free_far_inside).  Bit-identical rows; k_fused at 131072 envs max 292 ->
#endif                               // 240 us, mean 165.5 -> 163.2, but median 149.5 -> 156.1, and at 16384 envs mean 123.9 -> 125.5 (profiles/r06zj_switch_ab.txt): off.
#ifndef PBRE_FRIC_FOLD               // 1: the staged friction rows of the two-chain sweeps take their "normal impulse is 0: skip" from the bounds (see cstage)
#define PBRE_FRIC_FOLD 1
#endif
#ifndef PBRE_RO_LIMIT_SPECIAL        // 1: a robot-only wave with exactly ONE joint at a limit runs a copy of its sweeps with that joint's limit row inlined -- no
#define PBRE_RO_LIMIT_SPECIAL 1      // per-joint scalar test + branch between the rows (9 x ~15 cycles per sweep for a lone wave)
#endif
#ifndef PBRE_PROBE_PATH      // (probe builds: which solver path a wave took)
#define PBRE_PROBE_PATH(k)
#endif

// No implicit FMA contraction in this file.  Core::step solves the same rows through different code paths chosen per WAVE (two zipped chains,
// the plain loop, the object-table-only loop, ...), and which envs share a wave of the complex-env list depends on the order in which
// kernels appended them -- on timing.  The paths are the same arithmetic row by row; with the compiler free to fuse a multiply and an add
// here but not there they differed in the last bit, and a complex env's result depended on its wave-mates (found by running two
// identical engines side by side: tools/diag_fast3.py).  Every fused operation in this file is an explicit L::fma.
PBRE_FP_CONTRACT_OFF
namespace pbre {

template <class L, class SH = Shape16>
struct Core {
    using F = typename L::F;
    using I = typename L::I;
    using B = typename L::B;
    using Tables = TablesT<SH>;
    static constexpr int W = SH::W, NJ = SH::NJ, LC = SH::LC, L1 = SH::L1, NSUB = SH::NSUB, NLEV = SH::NLEV, STATE = SH::STATE;
    static constexpr int NC_OT = SH::NC_OT, NC_RO = SH::NC_RO, NC_RT = SH::NC_RT, NC = SH::NC, NMW = SH::NMW, NTIP = SH::NTIP, TIP0 = SH::TIP0;
    // values that only exist on robot lanes (rows of M^-1, the motor / limit row a lane owns) use the backend's robot-lane
    // view: L itself, except for a backend that packs two virtual lanes into one physical lane (DevLanes128), where it is
    // the plain 64-lane backend of the low half
    using LR = typename L::Robot;
    using FR = typename LR::F;
    using IR = typename LR::I;
    using BR = typename LR::B;
    struct Mask { I w[NMW]; };         // ancestor / subtree bit mask of this lane, 32 joints per word
    static PBRE_HD Mask loadmask(const int (*m)[W]) { Mask r; PBRE_UNROLL for (int k = 0; k < NMW; k++) r.w[k] = L::loadI(m[k]); return r; }
    static PBRE_HD B mbit(const Mask& m, int i) { return L::bit(m.w[i >> 5], i & 31); }
    static PBRE_HD B mbiti(const Mask& m, I k) {     // bit k of the mask, k a lane value in 0..NJ-1
        B r = L::biti(m.w[0], k);
        PBRE_UNROLL for (int w = 1; w < NMW; w++) r = L::bor(L::band(L::lti(k, 32 * w), r), L::band(L::gei(k, 32 * w), L::biti(m.w[w], k)));
        return r;
    }

    struct V3 { F x, y, z; };
    struct Q4 { F x, y, z, w; };
    struct M3 { F m[9]; };
    struct Sp { V3 a, l; };            // spatial vector: angular, linear (world frame, about world origin)

    // ---------------------------------------------------------------- small math
    static PBRE_HD V3 v3(F x, F y, F z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
    static PBRE_HD V3 add(const V3& a, const V3& b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
    static PBRE_HD V3 sub(const V3& a, const V3& b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
    static PBRE_HD V3 scl(const V3& a, F s) { return v3(a.x * s, a.y * s, a.z * s); }
    static PBRE_HD F dot(const V3& a, const V3& b) { return L::fma(a.x, b.x, L::fma(a.y, b.y, a.z * b.z)); }
    static PBRE_HD V3 cross(const V3& a, const V3& b) {
        return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
    }
    static PBRE_HD F norm(const V3& a) { return L::sqrt(dot(a, a)); }
    static PBRE_HD V3 mv(const M3& A, const V3& v) {
        return v3(L::fma(A.m[0], v.x, L::fma(A.m[1], v.y, A.m[2] * v.z)),
                  L::fma(A.m[3], v.x, L::fma(A.m[4], v.y, A.m[5] * v.z)),
                  L::fma(A.m[6], v.x, L::fma(A.m[7], v.y, A.m[8] * v.z)));
    }
    static PBRE_HD V3 mtv(const M3& A, const V3& v) {
        return v3(L::fma(A.m[0], v.x, L::fma(A.m[3], v.y, A.m[6] * v.z)),
                  L::fma(A.m[1], v.x, L::fma(A.m[4], v.y, A.m[7] * v.z)),
                  L::fma(A.m[2], v.x, L::fma(A.m[5], v.y, A.m[8] * v.z)));
    }
    static PBRE_HD M3 mm(const M3& A, const M3& Bm) {
        M3 C;
        PBRE_UNROLL for (int i = 0; i < 3; i++)
            PBRE_UNROLL for (int j = 0; j < 3; j++)
                C.m[i*3+j] = L::fma(A.m[i*3], Bm.m[j], L::fma(A.m[i*3+1], Bm.m[3+j], A.m[i*3+2] * Bm.m[6+j]));
        return C;
    }
    static PBRE_HD V3 selv(B c, const V3& a, const V3& b) { return v3(L::sel(c, a.x, b.x), L::sel(c, a.y, b.y), L::sel(c, a.z, b.z)); }
    static PBRE_HD V3 bcastv(const V3& a, int i) { return v3(L::bcast(a.x, i), L::bcast(a.y, i), L::bcast(a.z, i)); }
    static PBRE_HD V3 bcastvI(const V3& a, I i) { return v3(L::gather(a.x, i), L::gather(a.y, i), L::gather(a.z, i)); }
    // pick component by lane: x for (lane%3==0 of the triple starting at `base`) ...
    static PBRE_HD F pick3(const V3& a, I lane, int base) {
        return L::sel(L::eqi(lane, base), a.x, L::sel(L::eqi(lane, base + 1), a.y, a.z));
    }
    static PBRE_HD M3 quat_R(const Q4& q) {
        M3 R; F x = q.x, y = q.y, z = q.z, w = q.w; F two = L::c(2.f), one = L::c(1.f);
        R.m[0] = one - two * (y*y + z*z); R.m[1] = two * (x*y - w*z);       R.m[2] = two * (x*z + w*y);
        R.m[3] = two * (x*y + w*z);       R.m[4] = one - two * (x*x + z*z); R.m[5] = two * (y*z - w*x);
        R.m[6] = two * (x*z - w*y);       R.m[7] = two * (y*z + w*x);       R.m[8] = one - two * (x*x + y*y);
        return R;
    }
    static PBRE_HD Q4 qmul(const Q4& a, const Q4& b) {
        Q4 o;
        o.x = a.w*b.x + a.x*b.w + a.y*b.z - a.z*b.y;
        o.y = a.w*b.y - a.x*b.z + a.y*b.w + a.z*b.x;
        o.z = a.w*b.z + a.x*b.y - a.y*b.x + a.z*b.w;
        o.w = a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z;
        return o;
    }
    // btMatrix3x3::getRotation (Shepperd); branch-free over the four cases
    static PBRE_HD Q4 R_quat(const M3& R) {
        F tr = R.m[0] + R.m[4] + R.m[8], one = L::c(1.f), h = L::c(.5f);
        // case trace > 0
        F s0 = L::sqrt(L::max(tr + one, L::c(1e-30f))); F k0 = h / s0;
        Q4 a; a.w = s0 * h; a.x = (R.m[7] - R.m[5]) * k0; a.y = (R.m[2] - R.m[6]) * k0; a.z = (R.m[3] - R.m[1]) * k0;
        // case i = 0,1,2 (largest diagonal)
        Q4 c[3];
        PBRE_UNROLL for (int i = 0; i < 3; i++) {
            int j = (i + 1) % 3, k = (i + 2) % 3;
            F s = L::sqrt(L::max(R.m[i*4] - R.m[j*4] - R.m[k*4] + one, L::c(1e-30f))); F kk = h / s;
            F t[3]; t[i] = s * h; t[j] = (R.m[j*3+i] + R.m[i*3+j]) * kk; t[k] = (R.m[k*3+i] + R.m[i*3+k]) * kk;
            c[i].x = t[0]; c[i].y = t[1]; c[i].z = t[2]; c[i].w = (R.m[k*3+j] - R.m[j*3+k]) * kk;
        }
        // i = R00 < R11 ? (R11 < R22 ? 2 : 1) : (R00 < R22 ? 2 : 0)
        B c01 = L::lt(R.m[0], R.m[4]), c12 = L::lt(R.m[4], R.m[8]), c02 = L::lt(R.m[0], R.m[8]);
        B use2 = L::bor(L::band(c01, c12), L::band(L::bnot(c01), c02));
        B use1 = L::band(c01, L::bnot(c12));
        B pos = L::gt(tr, L::c(0.f));
        Q4 o;
        o.x = L::sel(pos, a.x, L::sel(use2, c[2].x, L::sel(use1, c[1].x, c[0].x)));
        o.y = L::sel(pos, a.y, L::sel(use2, c[2].y, L::sel(use1, c[1].y, c[0].y)));
        o.z = L::sel(pos, a.z, L::sel(use2, c[2].z, L::sel(use1, c[1].z, c[0].z)));
        o.w = L::sel(pos, a.w, L::sel(use2, c[2].w, L::sel(use1, c[1].w, c[0].w)));
        return o;
    }
    // pybullet getQuaternionFromEuler / getEulerFromQuaternion
    static PBRE_HD Q4 euler_quat(const V3& e) {
        F h = L::c(.5f);
        F cr, sr, cp, sp, cy, sy;
        L::sincos(e.x * h, sr, cr); L::sincos(e.y * h, sp, cp); L::sincos(e.z * h, sy, cy);
        Q4 q; q.x = sr*cp*cy - cr*sp*sy; q.y = cr*sp*cy + sr*cp*sy; q.z = cr*cp*sy - sr*sp*cy; q.w = cr*cp*cy + sr*sp*sy;
        return q;
    }
    static PBRE_HD V3 quat_euler(const Q4& q) {
        F x = q.x, y = q.y, z = q.z, w = q.w, two = L::c(2.f);
        F sqx = x*x, sqy = y*y, sqz = z*z, squ = w*w;
        F sarg = L::c(-2.f) * (x*z - w*y);
        B lo = L::le(sarg, L::c(-0.99999f)), hi = L::ge(sarg, L::c(0.99999f));
        F roll = L::atan2(two * (y*z + w*x), squ - sqx - sqy + sqz);
        F pitch = L::asin(L::max(L::min(sarg, L::c(1.f)), L::c(-1.f)));
        F yaw = L::atan2(two * (x*y + w*z), squ + sqx - sqy - sqz);
        F hp = L::c(1.57079632679489662f);
        F ylo = two * L::atan2(x, L::c(0.f) - y), yhi = two * L::atan2(L::c(0.f) - x, y);
        V3 e;
        e.x = L::sel(L::bor(lo, hi), L::c(0.f), roll);
        e.y = L::sel(lo, L::c(0.f) - hp, L::sel(hi, hp, pitch));
        e.z = L::sel(lo, ylo, L::sel(hi, yhi, yaw));
        return e;
    }
    static PBRE_HD F clampf(F x, F lo, F hi) { return L::min(L::max(x, lo), hi); }

    // ---------------------------------------------------------------- kinematics
    struct Kin {
        M3 R; V3 p;          // world pose of this lane's link frame
        Sp S;                // joint motion axis (world, about world origin)
    };

    // Forward kinematics of all robot lanes by pointer jumping over the ancestor tables.
    static PBRE_HD void fk(const Tables& T, F q, Kin& K) {
        I jt = L::loadI(T.jtype);
        B rev = L::eqi(jt, 1), pri = L::eqi(jt, 2);
        V3 ax = v3(L::load(T.axis[0]), L::load(T.axis[1]), L::load(T.axis[2]));
        M3 R0; PBRE_UNROLL for (int k = 0; k < 9; k++) R0.m[k] = L::load(T.R0[k]);
        V3 p0 = v3(L::load(T.p0[0]), L::load(T.p0[1]), L::load(T.p0[2]));
        // Rodrigues rotation about the joint axis (identity for non-revolute lanes)
        F th = L::sel(rev, q, L::c(0.f));
        F c, s;
        L::sincos(th, s, c);
        F C = L::c(1.f) - c;
        M3 Rj;
        Rj.m[0] = c + ax.x*ax.x*C;      Rj.m[1] = ax.x*ax.y*C - ax.z*s; Rj.m[2] = ax.x*ax.z*C + ax.y*s;
        Rj.m[3] = ax.y*ax.x*C + ax.z*s; Rj.m[4] = c + ax.y*ax.y*C;      Rj.m[5] = ax.y*ax.z*C - ax.x*s;
        Rj.m[6] = ax.z*ax.x*C - ax.y*s; Rj.m[7] = ax.z*ax.y*C + ax.x*s; Rj.m[8] = c + ax.z*ax.z*C;
        M3 R = mm(R0, Rj);
        V3 d = mv(R0, ax);
        F qs = L::sel(pri, q, L::c(0.f));
        V3 p = v3(L::fma(d.x, qs, p0.x), L::fma(d.y, qs, p0.y), L::fma(d.z, qs, p0.z));
        PBRE_UNROLL for (int lev = 0; lev < NLEV; lev++) {
            I a = L::loadI(T.anc[lev]);
            B ok = L::gei(a, 0);
            if (!L::any(ok)) break;
            I ai = L::maxi(a, 0);
            M3 Ra; PBRE_UNROLL for (int k = 0; k < 9; k++) Ra.m[k] = L::gather(R.m[k], ai);
            V3 pa = bcastvI(p, ai);
            M3 Rn = mm(Ra, R);
            V3 pn = add(pa, mv(Ra, p));
            PBRE_UNROLL for (int k = 0; k < 9; k++) R.m[k] = L::sel(ok, Rn.m[k], R.m[k]);
            p = selv(ok, pn, p);
        }
        K.R = R; K.p = p;
        V3 aw = mv(R, ax);
        V3 z = v3(L::c(0.f), L::c(0.f), L::c(0.f));
        K.S.a = selv(rev, aw, z);
        K.S.l = selv(rev, cross(p, aw), selv(pri, aw, z));
    }

    // inclusive sum over the ancestor chain (pointer jumping): out_j = sum_{i anc-or-self j} x_i
    static PBRE_HD Sp chain_sum(const Tables& T, Sp x) {
        PBRE_UNROLL for (int lev = 0; lev < NLEV; lev++) {
            I a = L::loadI(T.anc[lev]);
            B ok = L::gei(a, 0);
            if (!L::any(ok)) break;
            I ai = L::maxi(a, 0);
            V3 ga = bcastvI(x.a, ai), gl = bcastvI(x.l, ai);
            F zero = L::c(0.f);
            x.a = add(x.a, selv(ok, ga, v3(zero, zero, zero)));
            x.l = add(x.l, selv(ok, gl, v3(zero, zero, zero)));
        }
        return x;
    }
    static PBRE_HD Sp crossm(const Sp& v, const Sp& m) {   // motion cross product v x m
        Sp o; o.a = cross(v.a, m.a); o.l = add(cross(v.a, m.l), cross(v.l, m.a)); return o;
    }

    // ---------------------------------------------------------------- collision helpers
    // sphere vs oriented box; returns signed distance, n = world normal box->sphere, pb = point on box
    static PBRE_HD F sphere_box(const V3& sc, F sr, const V3& bc, const M3& Rb, const V3& h, V3& n, V3& pb) {
        V3 dl = mtv(Rb, sub(sc, bc));
        V3 cl = v3(clampf(dl.x, L::c(0.f) - h.x, h.x), clampf(dl.y, L::c(0.f) - h.y, h.y), clampf(dl.z, L::c(0.f) - h.z, h.z));
        V3 df = sub(dl, cl);
        F len = norm(df);
        B inside = L::lt(len, L::c(1e-9f));
        F il = L::c(1.f) / L::max(len, L::c(1e-30f));
        V3 n_out = scl(df, il);
        // inside branch: nearest face
        F ex = h.x - L::abs(dl.x), ey = h.y - L::abs(dl.y), ez = h.z - L::abs(dl.z);
        B ax_y = L::lt(ey, ex);                       // strict '<' scanning x,y,z keeps the first minimum
        F best = L::sel(ax_y, ey, ex);
        B ax_z = L::lt(ez, best);
        best = L::sel(ax_z, ez, best);
        B is_x = L::band(L::bnot(ax_y), L::bnot(ax_z)), is_y = L::band(ax_y, L::bnot(ax_z));
        F one = L::c(1.f), zero = L::c(0.f);
        F sx = L::sel(L::ge(dl.x, zero), one, zero - one), sy = L::sel(L::ge(dl.y, zero), one, zero - one), sz = L::sel(L::ge(dl.z, zero), one, zero - one);
        V3 n_in = v3(L::sel(is_x, sx, zero), L::sel(is_y, sy, zero), L::sel(ax_z, sz, zero));
        V3 cl_in = v3(L::sel(is_x, sx * h.x, cl.x), L::sel(is_y, sy * h.y, cl.y), L::sel(ax_z, sz * h.z, cl.z));
        V3 nl = selv(inside, n_in, n_out);
        V3 c2 = selv(inside, cl_in, cl);
        n = mv(Rb, nl);
        pb = add(bc, mv(Rb, c2));
        return L::sel(inside, zero - best - sr, len - sr);
    }

    // sphere vs the round object primitives (Params::obj_shape 1 sphere / 2 cylinder about local z; pbre_objstep.hpp: Shapes, oracle:
    // sphere_shape): same contract as sphere_box.  Branch-free per lane (the shape itself is a batch constant).
    static PBRE_HD F sphere_round(int shape, const V3& sc, F sr, const V3& bc, const M3& Rb, const V3& h, V3& n, V3& pb) {
        const F one = L::c(1.f), zero = L::c(0.f);
        V3 d = sub(sc, bc);
        if (shape == 1) {
            F len = norm(d);
            B deg = L::lt(len, L::c(1e-9f));
            F il = one / L::max(len, L::c(1e-30f));
            n = v3(L::sel(deg, zero, d.x * il), L::sel(deg, zero, d.y * il), L::sel(deg, one, d.z * il));
            pb = add(bc, scl(n, h.x));
            return len - h.x - sr;
        }
        V3 dl = mtv(Rb, d);
        F rho = L::sqrt(L::fma(dl.x, dl.x, dl.y * dl.y));
        B ax0 = L::bnot(L::gt(rho, L::c(1e-12f)));
        F ir = one / L::max(rho, L::c(1e-30f));
        F ux = L::sel(ax0, one, dl.x * ir), uy = L::sel(ax0, zero, dl.y * ir);
        F rc = L::min(rho, h.x), zc = clampf(dl.z, zero - h.z, h.z);
        V3 cl = v3(ux * rc, uy * rc, zc);
        V3 df = sub(dl, cl);
        F len = norm(df);
        B inside = L::lt(len, L::c(1e-9f));
        F il = one / L::max(len, L::c(1e-30f));
        V3 n_out = scl(df, il);
        F er = h.x - rho, ez = h.z - L::abs(dl.z);
        B lat = L::le(er, ez);                         // leave through the lateral surface rather than a cap
        F sz = L::sel(L::ge(dl.z, zero), one, zero - one);
        V3 n_in = v3(L::sel(lat, ux, zero), L::sel(lat, uy, zero), L::sel(lat, zero, sz));
        V3 cl_in = v3(L::sel(lat, ux * h.x, dl.x), L::sel(lat, uy * h.x, dl.y), L::sel(lat, dl.z, sz * h.z));
        V3 nl = selv(inside, n_in, n_out);
        V3 c2 = selv(inside, cl_in, cl);
        n = mv(Rb, nl);
        pb = add(bc, mv(Rb, c2));
        return L::sel(inside, zero - L::sel(lat, er, ez) - sr, len - sr);
    }
    // Candidate contact point of the object against its support surface owned by lane v = 0..7 (offset from the centre, world axes) and
    // whether the shape uses that slot (pbre_objstep.hpp: Shapes::candidate, same rules)
    static PBRE_HD V3 shape_candidate(int shape, const V3& oh, const M3& Ro, I lane, B& used) {
        const F one = L::c(1.f), zero = L::c(0.f);
        if (shape == 0) {
            F sgx = L::sel(L::bit(lane, 0), one, zero - one), sgy = L::sel(L::bit(lane, 1), one, zero - one), sgz = L::sel(L::bit(lane, 2), one, zero - one);
            used = L::lti(lane, 8);
            return mv(Ro, v3(sgx * oh.x, sgy * oh.y, sgz * oh.z));
        }
        if (shape == 1) { used = L::eqi(lane, 0); return v3(zero, zero, zero - oh.x); }
        const F s = L::sel(L::bit(lane, 2), one, zero - one);
        const B b0 = L::bit(lane, 0), b1 = L::bit(lane, 1);          // rim slot k = lane & 3: 0, 1, 2 fixed points, 3 the lowest point
        const B k0 = L::band(L::bnot(b0), L::bnot(b1)), k1 = L::band(b0, L::bnot(b1));
        const F cs = L::sel(k0, one, L::c(-0.5f));
        const F sn = L::sel(k0, zero, L::sel(k1, L::c(0.86602540378443865f), L::c(-0.86602540378443865f)));
        const F dx = zero - Ro.m[6], dy = zero - Ro.m[7];
        const F len = L::sqrt(L::fma(dx, dx, dy * dy));
        const B low_ok = L::ge(len, L::c(1e-6f));
        const F il = one / L::max(len, L::c(1e-30f));
        const B is_low = L::band(b0, b1);
        const F lx = L::sel(is_low, oh.x * dx * il, oh.x * cs), ly = L::sel(is_low, oh.x * dy * il, oh.x * sn);
        used = L::band(L::lti(lane, 8), L::bor(L::bnot(is_low), low_ok));
        return mv(Ro, v3(lx, ly, s * oh.z));
    }

    // sphere vs a convex-hull object (Params::obj_shape 3, pbre_set_object_hull; oracle: sphere_hull): same contract as sphere_box.  H: the
    // hull table (pbre_tables.hpp), read at group-uniform addresses.  A sphere whose centre is farther from the object's origin than the
    // hull's bounding radius + its own radius + the margin cannot be a contact: its lanes get that lower bound as distance (> margin, never
    // selected) and, when no lane of the wave is nearer, the face loop is skipped.  Otherwise per lane: the nearest face plane if the centre
    // is behind every face (inside: the box / cylinder rule), else the closest point over the face triangles (Ericson, Real-Time Collision
    // Detection 5.1.5, written with selects in the priority order of its early returns: vertex A, B, edge AB, vertex C, edge AC, BC, face).
    static PBRE_HD F sphere_hull(const float* H, int nf, float rb, const V3& sc, F sr, const V3& bc, const M3& Rb, F margin, V3& n, V3& pb) {
        const F one = L::c(1.f), zero = L::c(0.f);
        V3 d = sub(sc, bc);
        V3 p = mtv(Rb, d);
        F len0 = norm(p);
        F far = len0 - L::c(rb) - sr;
        B near = L::lt(far, margin);
        F il0 = one / L::max(len0, L::c(1e-30f));
        n = v3(L::sel(L::gt(len0, zero), d.x * il0, zero), L::sel(L::gt(len0, zero), d.y * il0, zero), L::sel(L::gt(len0, zero), d.z * il0, one));
        pb = add(bc, scl(n, L::c(rb)));
        if (!L::any(near)) return far;
        F best2 = L::c(3e38f), maxsd = L::c(-3e38f);
        V3 bcp = v3(zero, zero, zero), nin = v3(zero, zero, one);
        for (int f = 0; f < nf; f++) {
            const float* T = H + HULL_T0 + 12 * f;
            const V3 a = v3(L::loadu(T + 0), L::loadu(T + 1), L::loadu(T + 2)), ab = v3(L::loadu(T + 3), L::loadu(T + 4), L::loadu(T + 5));
            const V3 ac = v3(L::loadu(T + 6), L::loadu(T + 7), L::loadu(T + 8)), nr = v3(L::loadu(T + 9), L::loadu(T + 10), L::loadu(T + 11));
            const V3 ap = sub(p, a);
            const F sd = dot(nr, ap);
            const B deeper = L::gt(sd, maxsd);
            maxsd = L::sel(deeper, sd, maxsd); nin = selv(deeper, nr, nin);
            const F d1 = dot(ab, ap), d2 = dot(ac, ap);
            const V3 bp = sub(ap, ab), cp = sub(ap, ac);
            const F d3 = dot(ab, bp), d4 = dot(ac, bp), d5 = dot(ab, cp), d6 = dot(ac, cp);
            const F vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
            const F den = one / (va + vb + vc);
            F v = vb * den, w = vc * den;                                                   // face interior
            const F e43 = d4 - d3, e56 = d5 - d6;
            const B rBC = L::band(L::le(va, zero), L::band(L::ge(e43, zero), L::ge(e56, zero)));
            const F wbc = e43 / (e43 + e56);
            v = L::sel(rBC, one - wbc, v); w = L::sel(rBC, wbc, w);
            const B rAC = L::band(L::le(vb, zero), L::band(L::ge(d2, zero), L::le(d6, zero)));
            v = L::sel(rAC, zero, v); w = L::sel(rAC, d2 / (d2 - d6), w);
            const B rC = L::band(L::ge(d6, zero), L::le(d5, d6));
            v = L::sel(rC, zero, v); w = L::sel(rC, one, w);
            const B rAB = L::band(L::le(vc, zero), L::band(L::ge(d1, zero), L::le(d3, zero)));
            v = L::sel(rAB, d1 / (d1 - d3), v); w = L::sel(rAB, zero, w);
            const B rB = L::band(L::ge(d3, zero), L::le(d4, d3));
            v = L::sel(rB, one, v); w = L::sel(rB, zero, w);
            const B rA = L::band(L::le(d1, zero), L::le(d2, zero));
            v = L::sel(rA, zero, v); w = L::sel(rA, zero, w);
            const V3 q = add(a, add(scl(ab, v), scl(ac, w)));
            const V3 e = sub(p, q);
            const F e2 = dot(e, e);
            const B closer = L::lt(e2, best2);                                              // (a NaN from a degenerate triangle's 0 / 0 never wins)
            best2 = L::sel(closer, e2, best2); bcp = selv(closer, q, bcp);
        }
        const B inside = L::le(maxsd, zero);
        const F len = L::sqrt(best2);
        const B deg = L::lt(len, L::c(1e-12f));
        const F il = one / L::max(len, L::c(1e-30f));
        const V3 eo = sub(p, bcp);
        V3 n_out = v3(L::sel(deg, zero, eo.x * il), L::sel(deg, zero, eo.y * il), L::sel(deg, one, eo.z * il));
        V3 nl = selv(inside, nin, n_out);
        V3 cl = selv(inside, sub(p, scl(nin, maxsd)), bcp);                                  // inside: the centre's projection onto the nearest face plane
        F dist = L::sel(inside, maxsd - sr, len - sr);
        V3 nw = mv(Rb, nl), pw = add(bc, mv(Rb, cl));
        n = selv(near, nw, n); pb = selv(near, pw, pb);
        return L::sel(near, dist, far);
    }

    // Select the `cap` smallest-distance candidates (dist < margin) among lanes with `valid`;
    // returns the per-lane rank (0..cap-1 in lane order) or -1.  Ties resolve to the lowest lane.
    static PBRE_HD I select_k(F dist, B valid, F margin, int cap, I lane) {
        B cand = L::band(valid, L::lt(dist, margin));
        if (!L::any(cand)) return L::ci(-1);         // nothing within the margin anywhere in the wave (the usual case)
        B chosen = L::bfalse();
        for (int r = 0; r < cap; r++) {
            F key = L::sel(L::band(cand, L::bnot(chosen)), dist, L::c(3e38f));
            F mn = L::vmin(key);
            B hit = L::band(L::band(cand, L::bnot(chosen)), L::eq(key, mn));
            hit = L::band(hit, L::lt(mn, L::c(1e38f)));
            // lowest lane among hits
            F lk = L::sel(hit, L::itof(lane), L::c(999.f));
            F lm = L::vmin(lk);
            chosen = L::bor(chosen, L::band(hit, L::eq(lk, lm)));
        }
        // rank = number of chosen lanes below this lane
        F cf = L::sel(chosen, L::c(1.f), L::c(0.f));
        F rank = L::c(0.f);
        PBRE_UNROLL for (int i = 0; i < W; i++) {
            F ci = L::bcast(cf, i);
            rank = rank + L::sel(L::lti(L::ci(i), lane), ci, L::c(0.f));
        }
        return L::seli(chosen, L::ftoi(rank), L::ci(-1));
    }

    // ---------------------------------------------------------------- contact slot (group-uniform values)
    struct Contact {
        B act; V3 n, pA, pB; F dist, mu; I owner;
    };

    // fetch the contact whose rank == r from the candidate lanes (all fields become group-uniform)
    static PBRE_HD Contact fetch(I rank, int r, const V3& n, const V3& pA, const V3& pB, F dist, F mu, I owner, I lane) {
        B mine = L::eqi(rank, r);
        if (!L::any(mine)) {                         // empty slot in every group of the wave: no gathers
            Contact e;
            const F z = L::c(0.f);
            e.act = L::bfalse(); e.n = v3(z, z, z); e.pA = e.n; e.pB = e.n; e.dist = z; e.mu = z; e.owner = L::ci(0);
            return e;
        }
        F lk = L::sel(mine, L::itof(lane), L::c(999.f));
        F lm = L::vmin(lk);
        Contact c;
        c.act = L::lt(lm, L::c(998.f));
        I src = L::ftoi(L::min(lm, L::c((float)(W - 1))));
        c.n = bcastvI(n, src); c.pA = bcastvI(pA, src); c.pB = bcastvI(pB, src);
        c.dist = L::gather(dist, src); c.mu = L::gather(mu, src);
        c.owner = L::gatherI(owner, src);
        return c;
    }

    // ---------------------------------------------------------------- the step
    // mode bits
    enum { M_ACTION = 1, M_OBS = 2, M_TASK = 4, M_TGT = 8, M_INITD = 16,
           M_INNER = 32 };   // a non-final iteration of the apply_action loop (action_repeat > 1): termination test + counter, no outputs

    // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in every iteration
    // (what `#pragma unroll` only promises up to its size threshold)
    template <class Fn, int... Ks>
    static PBRE_HD void for_seq_(Fn&& f, std::integer_sequence<int, Ks...>) { (f(std::integral_constant<int, Ks>{}), ...); }
    template <int N, class Fn>
    static PBRE_HD void for_seq(Fn&& f) { for_seq_(f, std::make_integer_sequence<int, N>{}); }

    struct Rows {             // register-resident solver data
        FR Mi[NJ];            // row of M^-1 (lane k: Minv[k][j])
        FR m_dinv, m_rhs;     // motor row owned by this lane (rhs already multiplied by dinv)
        FR l_j, l_rhs;        // limit row owned by this lane: J' = dir*dinv (0 if inactive), rhs'
        FR l_dir;
        FR m_app, l_app;      // applied impulse of the motor / limit row this lane owns
        FR m_lim;             // impulse bound of the motor row
        // contact rows: row 6 c + 2 d (J') and 6 c + 2 d + 1 (B) of contact c, direction d (normal, two tangents).  The store is
        // the backend's: registers, or wave-private LDS where the rows would not fit the register file (DevLanes128)
        typename L::template RowStore<6 * NC> rs;
        F an[NC], a1[NC], a2[NC];
        F mu[NC];
        B act[NC];
    };

    // app / lim are group-uniform; L::uni tells a backend that stores a value in more than one register per lane so
    // OBJ: the row only touches the object (object-table contacts): its J' is zero on every robot lane, which lets a backend
    // use a cheaper all-reduce (L::sum_obj; same value bit for bit)
    // (both return the row's impulse change, group-uniform: what the residual test of step<RT> looks at)
    template <bool OBJ = false>
    static PBRE_HD F row(F Jp, F Bv, F& app, F lo, F hi, F& dv) {
        app = L::uni(app);
        F t = (OBJ && LC >= 16) ? L::sum_obj(Jp * dv) : L::sum(Jp * dv);      // (sum_obj: the object lanes lie in the upper 16-lane row)
        F s = L::med3(app - t, lo, hi);
        F d = s - app; app = s;
        dv = L::fma(d, Bv, dv);
        return d;
    }
    template <bool OBJ = false>
    static PBRE_HD F frow(F Jp, F Bv, F& app, F lim, F& dv) {   // friction row, skipped when normal impulse <= 0
        app = L::uni(app); lim = L::uni(lim);
        F t = (OBJ && LC >= 16) ? L::sum_obj(Jp * dv) : L::sum(Jp * dv);      // (sum_obj: the object lanes lie in the upper 16-lane row)
        F s = L::med3(app - t, L::c(0.f) - lim, lim);
        s = L::sel(L::gt(lim, L::c(0.f)), s, app);
        F d = s - app; app = s;
        dv = L::fma(d, Bv, dv);
        return d;
    }

    // One simulation step for one env group.  st: pointer to the env's 48-float record.
    // act: pointer to this env's action row or nullptr.  out: this env's [obs_dim+2] row or nullptr.
    // objv: nullptr, or this group's W-float side record whose object-twist lanes hold the object's twist after a step without
    // robot-object contact (pbre_objstep.hpp); used by the groups that have no such contact.
    // st_out: where the new state goes (default: back into st).  The row kernels' idle rows read a pristine record and write a scratch one,
    // so that what they compute in lockstep with the real rows stays the cheapest step there is (see k_row_list).
    // RT (pbre_physics.solver_residual_threshold > 0; Bullet's m_leastSquaresResidualThreshold [EXT-UNVERIFIED], oracle:
    // orc_params.solver_residual_threshold): a group leaves the sweep loop after the first sweep whose largest velocity-level row change
    // |delta impulse / jacDiagABInv|, over all of its rows, is <= P.res_lim.  Its velocity vector is snapshotted there; the wave goes on
    // until its last group is through, and what a group computes does not depend on its wave-mates.  sw: where the group's sweep count
    // goes (or null).  Callers pass objv = nullptr: the test is over ALL rows of an env, so the object's rows stay in this solve.
    template <bool RT = false>
    static PBRE_HD void step(const Tables& T, const Params& P, float* st, const float* act, float* out, int mode, int flags,
                             const float* tgt = nullptr, unsigned long long env_id = 0, const float* objv = nullptr, float* st_out = nullptr,
                             int* sw = nullptr) {
        const I lane = L::lane();
        const F zero = L::c(0.f), one = L::c(1.f);
        const B robot = L::lti(lane, NJ);
        const B is_lin = L::band(L::gei(lane, LC), L::lti(lane, LC + 3));
        const B is_ang = L::band(L::gei(lane, LC + 3), L::lti(lane, LC + 6));
        const B obj_lane = L::bor(is_lin, is_ang);
        const bool obj_on = !(flags & 1);
        const F dt = L::c(P.dt), inv_dt = L::c(P.inv_dt);

        PBRE_PROBE_DECL
        F Qr = L::load(st), Vr = L::load(st + W), Xr = L::loadm(st + 2 * W, L::lti(lane, 16));
        F q = L::sel(robot, Qr, zero);

        // ---- motor targets (apply_action): q_des = clip(q + 0.05 a, ll, ul) for actuated lanes, else hold at home
        F lower = L::load(T.lower), upper = L::load(T.upper);
        F qdes = L::load(T.home), kp = L::load(T.kp_hold), kd = L::load(T.kd_hold);
        FR fscale = LR::c(1.f);
        F vmx = zero;                // MREC: maxVelocity of the motor (0: unlimited)
        if (SH::MREC) {
            // PyBullet's motors persist between calls: target | kp | force scale of every joint live in the env's motor record
            // (written by apply_action below, by the IK kernel and by the finger commands open_hand / pre_grasp / grasp)
            float* mrec = const_cast<float*>(tgt);
            qdes = L::load(mrec); kp = L::load(mrec + W);
            F fs = L::load(mrec + 2 * W);
            vmx = L::load(mrec + 3 * W);
            if (mode & M_ACTION) {
                // iCubEnv.apply_action, joint branch (icub_env.py:341-361): absolute targets clipped to the joint limits,
                // positionGain 0.5, default force
                I ai = L::loadI(T.act_idx);
                B al = L::gei(ai, 0);
                F a = L::loadx(act, ai, al);
                qdes = L::sel(al, clampf(a, lower, upper), qdes);
                kp = L::sel(al, L::load(T.kp_act), kp); fs = L::sel(al, one, fs); vmx = L::sel(al, zero, vmx);
                L::store(mrec, qdes); L::store(mrec + W, kp); L::store(mrec + 2 * W, fs); L::store(mrec + 3 * W, vmx);
            }
            fscale = L::lo(fs);
        } else {
        if (mode & M_TGT) qdes = L::loadm(tgt, robot);     // IK mode: targets from the IK buffer, hold gains
        if (mode & M_ACTION) {
            I ai = L::loadI(T.act_idx);
            B al = L::gei(ai, 0);
            F a = L::loadx(act, ai, al);
            F tgt = clampf(L::fma(a, L::c(P.act_scale), q), lower, upper);
            qdes = L::sel(al, tgt, qdes);
            kp = L::load(T.kp_act); kd = L::load(T.kd_act);
        }
        }

        // ---- object pose/twist as group-uniform values
        V3 op = v3(L::bcast(Qr, LC), L::bcast(Qr, LC + 1), L::bcast(Qr, LC + 2));
        Q4 oq; oq.x = L::bcast(Qr, LC + 3); oq.y = L::bcast(Qr, LC + 4); oq.z = L::bcast(Qr, LC + 5); oq.w = L::bcast(Qr, LC + 6);
        V3 ov = v3(L::bcast(Vr, LC), L::bcast(Vr, LC + 1), L::bcast(Vr, LC + 2));
        V3 ow = v3(L::bcast(Vr, LC + 3), L::bcast(Vr, LC + 4), L::bcast(Vr, LC + 5));
        M3 Ro = quat_R(oq);

        // ---- kinematics at q_t
        PBRE_PROBE(0);      // loads, motor targets
        Kin K; fk(T, q, K);
        PBRE_PROBE(1);      // forward kinematics
        F qd = L::sel(robot, Vr, zero);

        // ---- rigid-body quantities of this lane's sub-bodies, velocities, bias forces (world-frame RNEA)
        Sp Sq; Sq.a = scl(K.S.a, qd); Sq.l = scl(K.S.l, qd);
        Sp Vs = chain_sum(T, Sq);                         // spatial velocity of this link
        Sp cj = crossm(Vs, Sq);
        Sp As = chain_sum(T, cj);                         // velocity-product acceleration
        As.l.z = As.l.z - L::c(P.gz);                     // gravity as base acceleration
        F cm = zero; V3 ch = v3(zero, zero, zero); F cI[6] = {zero, zero, zero, zero, zero, zero};   // own spatial inertia about world origin
        Sp Fo; Fo.a = v3(zero, zero, zero); Fo.l = v3(zero, zero, zero);
        PBRE_UNROLL for (int b = 0; b < NSUB; b++) {
            F m = L::load(T.sb_m[b]);
            V3 cl = v3(L::load(T.sb_c[b][0]), L::load(T.sb_c[b][1]), L::load(T.sb_c[b][2]));
            V3 c = add(K.p, mv(K.R, cl));
            // world inertia Iw = R I R^T (symmetric)
            F Ixx = L::load(T.sb_I[b][0]), Iyy = L::load(T.sb_I[b][1]), Izz = L::load(T.sb_I[b][2]);
            F Ixy = L::load(T.sb_I[b][3]), Ixz = L::load(T.sb_I[b][4]), Iyz = L::load(T.sb_I[b][5]);
            M3 Il; Il.m[0] = Ixx; Il.m[1] = Ixy; Il.m[2] = Ixz; Il.m[3] = Ixy; Il.m[4] = Iyy; Il.m[5] = Iyz; Il.m[6] = Ixz; Il.m[7] = Iyz; Il.m[8] = Izz;
            M3 RI = mm(K.R, Il);
            M3 Iw;
            PBRE_UNROLL for (int i = 0; i < 3; i++)
                PBRE_UNROLL for (int j = 0; j < 3; j++)
                    Iw.m[i*3+j] = L::fma(RI.m[i*3], K.R.m[j*3], L::fma(RI.m[i*3+1], K.R.m[j*3+1], RI.m[i*3+2] * K.R.m[j*3+2]));
            // Newton-Euler of the sub-body
            V3 w = Vs.a;
            V3 vc = add(Vs.l, cross(w, c));
            V3 ac = add(add(As.l, cross(As.a, c)), cross(w, vc));
            // (Panda task envs: V[15] = 1 + this env's robot link damping, 0 = the batch value; pbre_set_physics_per_env)
            F r_kl = L::c(P.kl);
            if (W == 16 && !SH::MREC) { const F v15 = L::bcast(Vr, 15); r_kl = L::sel(L::gt(v15, zero), v15 - one, r_kl); }
            F sl = L::fma(r_kl, norm(vc), r_kl);           // Bullet velocity damping K1 + K2|v|
            V3 f = scl(add(ac, scl(vc, sl)), m);
            V3 Iww = mv(Iw, w);
            F sa = L::fma(L::c(P.ka), norm(w), L::c(P.ka));
            V3 nc = add(add(mv(Iw, As.a), cross(w, Iww)), scl(Iww, sa));
            Fo.a = add(Fo.a, add(nc, cross(c, f)));
            Fo.l = add(Fo.l, f);
            // spatial inertia about the world origin: m, h = m c, Io = Iw + m (c.c 1 - c c^T)
            cm = cm + m;
            ch = add(ch, scl(c, m));
            F cc = dot(c, c);
            cI[0] = cI[0] + L::fma(m, cc - c.x*c.x, Iw.m[0]); cI[1] = cI[1] + L::fma(m, cc - c.y*c.y, Iw.m[4]); cI[2] = cI[2] + L::fma(m, cc - c.z*c.z, Iw.m[8]);
            cI[3] = cI[3] + L::fma(zero - m, c.x*c.y, Iw.m[1]); cI[4] = cI[4] + L::fma(zero - m, c.x*c.z, Iw.m[2]); cI[5] = cI[5] + L::fma(zero - m, c.y*c.z, Iw.m[5]);
        }
        PBRE_PROBE(2);      // velocities, bias forces, inertias (RNEA)
        // ---- subtree sums (bias force + composite inertia): broadcast loop with descendant masks
        const Mask dmask = loadmask(T.dmask);
        Sp Fs; Fs.a = v3(zero, zero, zero); Fs.l = v3(zero, zero, zero);
        F Cm = zero; V3 Ch = v3(zero, zero, zero); F CI[6] = {zero, zero, zero, zero, zero, zero};
        PBRE_UNROLL for (int i = 0; i < NJ; i++) {
            B in = mbit(dmask, i);
            Fs.a = add(Fs.a, selv(in, bcastv(Fo.a, i), v3(zero, zero, zero)));
            Fs.l = add(Fs.l, selv(in, bcastv(Fo.l, i), v3(zero, zero, zero)));
            Cm = Cm + L::sel(in, L::bcast(cm, i), zero);
            Ch = add(Ch, selv(in, bcastv(ch, i), v3(zero, zero, zero)));
            PBRE_UNROLL for (int k = 0; k < 6; k++) CI[k] = CI[k] + L::sel(in, L::bcast(cI[k], i), zero);
        }
        F tau = zero - (dot(K.S.a, Fs.a) + dot(K.S.l, Fs.l)) - L::load(T.jdamp) * qd;   // -bias - joint damping

        PBRE_PROBE(3);      // subtree sums
        // ---- CRBA: G = Ic S (own), H[row=lane][i]
        M3 Io; Io.m[0] = CI[0]; Io.m[1] = CI[3]; Io.m[2] = CI[4]; Io.m[3] = CI[3]; Io.m[4] = CI[1]; Io.m[5] = CI[5]; Io.m[6] = CI[4]; Io.m[7] = CI[5]; Io.m[8] = CI[2];
        Sp G; G.a = add(mv(Io, K.S.a), cross(Ch, K.S.l)); G.l = add(scl(K.S.l, Cm), cross(K.S.a, Ch));
        const Mask amask = loadmask(T.amask);
        const IR laneR = LR::lane();
        const FR zeroR = LR::c(0.f), oneR = LR::c(1.f);
        Rows R;
        R.rs.init();
        PBRE_UNROLL for (int i = 0; i < NJ; i++) {
            V3 Sa = bcastv(K.S.a, i), Sl = bcastv(K.S.l, i), Ga = bcastv(G.a, i), Gl = bcastv(G.l, i);
            F up = dot(Sa, G.a) + dot(Sl, G.l);            // i is an ancestor-or-self of this lane
            F dn = dot(K.S.a, Ga) + dot(K.S.l, Gl);        // this lane is an ancestor of i
            R.Mi[i] = L::lo(L::sel(mbit(amask, i), up, L::sel(mbit(dmask, i), dn, zero)));
        }
        if (P.jd_dt != 0.f) {     // implicit joint damping: M + dt C (pbre_physics.implicit_joint_damping)
            const FR add = LR::c(P.jd_dt) * LR::load(T.jdamp);
            PBRE_UNROLL for (int i = 0; i < NJ; i++) R.Mi[i] = LR::sel(LR::eqi(laneR, i), R.Mi[i] + add, R.Mi[i]);
        }
        // unused robot lanes (fewer than 9 DoF): unit diagonal keeps the inverse well defined
        {
            const BR nojoint = LR::eqi(LR::loadI(T.jtype), 0);
            PBRE_UNROLL for (int i = 0; i < NJ; i++) R.Mi[i] = LR::sel(LR::band(LR::eqi(laneR, i), nojoint), oneR, R.Mi[i]);
        }

        PBRE_PROBE(4);      // CRBA
        // ---- M^-1 by in-place Gauss-Jordan (SPD, no pivoting), one matrix row per lane
        if (NJ > 40) {
            // 60 x 60: a rolled pivot loop (fully unrolled it is ~20k instructions and beyond the compiler's unroll budget, which
            // would leave R in scratch memory).  The columns rotate left by one per step, so the pivot column is always Mi[0] and
            // every register index stays a compile-time constant; after NJ steps the columns are back in place.  Same
            // arithmetic per element as the unrolled form below.
            for (int c = 0; c < NJ; c++) {
                FR pc = LR::bcast(R.Mi[0], c);
                FR inv = oneR / pc;
                BR isc = LR::eqi(laneR, c);
                FR f = R.Mi[0];
                FR piv = LR::sel(isc, inv, zeroR - f * inv);
                PBRE_UNROLL for (int k = 1; k < NJ; k++) {
                    FR rc = LR::bcast(R.Mi[k], c) * inv;
                    R.Mi[k - 1] = LR::sel(isc, rc, LR::fma(zeroR - f, rc, R.Mi[k]));
                }
                R.Mi[NJ - 1] = piv;
            }
        } else
        PBRE_UNROLL for (int c = 0; c < NJ; c++) {
            FR pc = LR::bcast(R.Mi[c], c);
            FR inv = oneR / pc;
            BR isc = LR::eqi(laneR, c);
            FR f = R.Mi[c];
            PBRE_UNROLL for (int k = 0; k < NJ; k++) {
                if (k == c) continue;
                FR rc = LR::bcast(R.Mi[k], c) * inv;
                R.Mi[k] = LR::sel(isc, rc, LR::fma(zeroR - f, rc, R.Mi[k]));
            }
            R.Mi[c] = LR::sel(isc, inv, zeroR - f * inv);
        }

        PBRE_PROBE(5);      // M^-1
        // ---- unconstrained velocities v* (ABA equivalent): qdd = M^-1 tau
        F qdd;
        {
            const FR tauR = L::lo(tau);
            FR acc = zeroR;
            PBRE_UNROLL for (int j = 0; j < NJ; j++) acc = LR::fma(R.Mi[j], LR::bcast(tauR, j), acc);
            qdd = L::wide(acc);
        }
        const F vmax = L::c(P.vmax);
        F vstar = clampf(L::fma(dt, qdd, qd), zero - vmax, vmax);
        // per-env object parameters of the Panda task envs (domain randomisation, pbre_set_physics_per_env): X[12] mass, X[13] lateral
        // friction, X[15] 1 + linear damping, 0 = the batch value; the inertia scales with the mass.  (W = 16 only: on the iCub X[12],
        // X[13] are the distances of the push reward.)
        F o_m = L::c(P.obj_m), o_mu = L::c(P.obj_mu), o_kl = L::c(P.kl);
        if (W == 16) {
            const F x12 = L::bcast(Xr, 12), x13 = L::bcast(Xr, 13), x15 = L::bcast(Xr, 15);
            o_m = L::sel(L::gt(x12, zero), x12, o_m); o_mu = L::sel(L::gt(x13, zero), x13, o_mu); o_kl = L::sel(L::gt(x15, zero), x15 - one, o_kl);
        }
        // object: gravity, damping, gyroscopic torque
        const F isc = o_m / L::c(P.obj_m);
        M3 Iinv; V3 oI = v3(L::c(P.obj_I[0]) * isc, L::c(P.obj_I[1]) * isc, L::c(P.obj_I[2]) * isc);
        {
            M3 D; PBRE_UNROLL for (int i = 0; i < 3; i++) { D.m[i*3] = Ro.m[i*3] / oI.x; D.m[i*3+1] = Ro.m[i*3+1] / oI.y; D.m[i*3+2] = Ro.m[i*3+2] / oI.z; }
            PBRE_UNROLL for (int i = 0; i < 3; i++)
                PBRE_UNROLL for (int j = 0; j < 3; j++)
                    Iinv.m[i*3+j] = L::fma(D.m[i*3], Ro.m[j*3], L::fma(D.m[i*3+1], Ro.m[j*3+1], D.m[i*3+2] * Ro.m[j*3+2]));
        }
        F vobj = zero;
        if (obj_on) {
            V3 wl = mtv(Ro, ow);
            V3 Lw = mv(Ro, v3(wl.x * oI.x, wl.y * oI.y, wl.z * oI.z));   // I_w w
            F sl = L::fma(o_kl, norm(ov), o_kl);
            V3 al = v3(zero - sl * ov.x, zero - sl * ov.y, L::c(P.gz) - sl * ov.z);
            F sa = L::fma(L::c(P.ka), norm(ow), L::c(P.ka));
            V3 tq = sub(scl(cross(ow, Lw), zero - one), scl(Lw, sa));
            V3 aa = mv(Iinv, tq);
            V3 vl = v3(L::fma(dt, al.x, ov.x), L::fma(dt, al.y, ov.y), L::fma(dt, al.z, ov.z));
            V3 va = v3(L::fma(dt, aa.x, ow.x), L::fma(dt, aa.y, ow.y), L::fma(dt, aa.z, ow.z));
            vobj = L::sel(is_lin, pick3(vl, lane, LC), L::sel(is_ang, pick3(va, lane, LC + 3), zero));
            vobj = clampf(vobj, zero - vmax, vmax);
        }
        vstar = L::sel(robot, vstar, vobj);                 // generalized v* of all 15 DoF (lane 15: 0)

        PBRE_PROBE(6);      // v*, object dynamics
        // ---- collision detection at q_t
        // Per-lane candidates stay alive through the row setup; the group-uniform Contact of a slot is fetched where its
        // rows are built (all NC of them at once would be 13 values x NC of register pressure on top of the M^-1 rows).
        const F margin = L::c(P.margin);
        I so, rk_ot, rk_ro, rk_rt;
        V3 up, vx, vB, n_ro, pB_ro, pA_ro, n_rt, pB_rt, pA_rt;
        F vd, d_ro, d_rt, smu;
        {
            // candidates: sphere lane s vs object / table; vertex lane v (0..7) vs support surface
            so = L::loadI(T.s_owner);
            B sv = L::nei(L::loadI(T.s_valid), 0);
            M3 Rs; PBRE_UNROLL for (int k = 0; k < 9; k++) Rs.m[k] = L::gather(K.R.m[k], so);
            V3 ps = bcastvI(K.p, so);
            V3 sc = add(ps, mv(Rs, v3(L::load(T.s_c[0]), L::load(T.s_c[1]), L::load(T.s_c[2]))));
            F sr = L::load(T.s_r);
            smu = L::load(T.s_mu);
            V3 oh = v3(L::c(P.obj_h[0]), L::c(P.obj_h[1]), L::c(P.obj_h[2]));
            // object vertices (box) / candidate points of a round object
            B cand_used;
            F top = L::c(P.tab_c[2] + P.tab_h[2]), bot = L::c(P.tab_c[2] - P.tab_h[2]);
            auto support = [&](const V3& x) -> F {      // height of the support surface under a world point: table top inside the footprint (unless below the slab), else ground
                B infoot = L::band(L::le(L::abs(x.x - L::c(P.tab_c[0])), L::c(P.tab_h[0])), L::le(L::abs(x.y - L::c(P.tab_c[1])), L::c(P.tab_h[1])));
                return L::sel(L::band(infoot, L::gt(x.z, bot)), top, L::c(P.ground_z));
            };
            if (P.obj_shape == 3) {
                // convex hull: every vertex is a candidate.  Lane v holds vertex v + pass * W; with more vertices than lanes each pass keeps its
                // NC_OT deepest (select_k), their survivors are gathered onto lanes pass * NC_OT + rank -- vertex order is preserved -- and the
                // selection below runs over those: the NC_OT deepest of all vertices, ties to the lower index, in vertex order (oracle:
                // select_contacts over obj_hull[])
                constexpr int PASSES = (HULL_MAXV + W - 1) / W;
                static_assert(PASSES * NC_OT <= W, "the survivors of every pass fit one lane group");
                auto vertex = [&](int pass, B& used) -> V3 {
                    used = L::lti(lane, L::ci(P.hull_nv - pass * W));
                    const I at = L::ftoi(L::itof(lane) * L::c(4.f));                 // (lane arithmetic through the float unit: the lane backends have no integer operators)
                    const float* hv = P.hull + HULL_V0 + 4 * pass * W;
                    return add(op, mv(Ro, v3(L::loadx(hv, at, used), L::loadx(hv + 1, at, used), L::loadx(hv + 2, at, used))));
                };
                if (PASSES == 1 || P.hull_nv <= W) {
                    vx = vertex(0, cand_used);
                } else {
                    const F z = L::c(0.f);
                    vx = v3(z, z, z); cand_used = L::bfalse();
                    for (int pass = 0; pass < PASSES; pass++) {
                        B used_p;
                        const V3 vp = vertex(pass, used_p);
                        const F vdp = vp.z - support(vp);
                        const I rkp = select_k(vdp, used_p, L::c(P.margin), NC_OT, lane);
                        for (int c = 0; c < NC_OT; c++) {
                            const Contact cc = fetch(rkp, c, vp, vp, vp, vdp, z, L::ci(0), lane);
                            const B here = L::band(L::eqi(lane, L::ci(pass * NC_OT + c)), cc.act);
                            vx = selv(here, cc.pA, vx); cand_used = L::bor(cand_used, here);
                        }
                    }
                }
            } else vx = add(op, shape_candidate(P.obj_shape, oh, Ro, lane, cand_used));
            F hs = support(vx);
            vd = vx.z - hs;
            up = v3(zero, zero, one);
            I none = L::ci(-1);
            rk_ot = none; rk_ro = none;
            d_ro = L::c(1.f);
            if (obj_on) {
                rk_ot = select_k(vd, cand_used, margin, NC_OT, lane);
                d_ro = P.obj_shape == 0 ? sphere_box(sc, sr, op, Ro, oh, n_ro, pB_ro)
                     : (P.obj_shape == 3 ? sphere_hull(P.hull, P.hull_nf, P.hull_rb, sc, sr, op, Ro, margin, n_ro, pB_ro) : sphere_round(P.obj_shape, sc, sr, op, Ro, oh, n_ro, pB_ro));
                pA_ro = add(pB_ro, scl(n_ro, d_ro));
                rk_ro = select_k(d_ro, sv, margin, NC_RO, lane);
            } else { n_ro = up; pB_ro = up; pA_ro = up; }
            M3 Idm; PBRE_UNROLL for (int k = 0; k < 9; k++) Idm.m[k] = (k % 4 == 0) ? one : zero;
            d_rt = sphere_box(sc, sr, v3(L::c(P.tab_c[0]), L::c(P.tab_c[1]), L::c(P.tab_c[2])), Idm,
                                v3(L::c(P.tab_h[0]), L::c(P.tab_h[1]), L::c(P.tab_h[2])), n_rt, pB_rt);
            pA_rt = add(pB_rt, scl(n_rt, d_rt));
            rk_rt = select_k(d_rt, sv, margin, NC_RT, lane);
            vB = v3(vx.x, vx.y, hs);
        }
        PBRE_PROBE(7);      // collision detection
        auto contact_of = [&](int c) -> Contact {
            if (c < NC_OT) return fetch(rk_ot, c, up, vx, vB, vd, o_mu * L::c(P.tab_mu), L::ci(0), lane);
            if (c < NC_OT + NC_RO) return fetch(rk_ro, c - NC_OT, n_ro, pA_ro, pB_ro, d_ro, smu * o_mu, so, lane);
            return fetch(rk_rt, c - NC_OT - NC_RO, n_rt, pA_rt, pB_rt, d_rt, smu * L::c(P.tab_mu), so, lane);
        };
        I owner_ro[NC_RO];           // link lane of each robot-object contact (fingertip bookkeeping)
        // Groups without a robot-object contact: robot rows and object rows share no unknown, the object's half of the solve was done
        // by kw_obj (one thread per env).  `split`: lanes of such groups; split_all: the whole wave -- then the object-table rows are
        // not even built.  In a mixed wave a split group's object rows are solved along with the others' and their result dropped,
        // so what an env computes never depends on the env it shares a wave with.
        const bool use_objv = objv != nullptr && obj_on;
        B split = L::lti(lane, 0);
        bool split_all = false;
        if (use_objv) {
            const Contact c0 = contact_of(NC_OT);       // slots fill in rank order: slot 0 is in use iff there is any such contact
            split = L::bnot(c0.act);
            split_all = !L::any(c0.act);
        }

        // ---- constraint rows
        FR m_diag = oneR;            // RT: (M^-1)_jj on lane j = 1 / dinv of joint j's motor and limit rows
        F r_den[NC][3];              // RT: 1 / dinv of the contact rows (group-uniform; dead code without RT)
        // motors (btMultiBodyJointMotor, POSITION_CONTROL): velocity error kp (q_des - q)/dt - kd v*
        {
            FR diag = oneR;                                 // this lane's diagonal entry of M^-1 (1 on lanes without a joint)
            PBRE_UNROLL for (int j = 0; j < NJ; j++) diag = LR::sel(LR::eqi(laneR, j), R.Mi[j], diag);
            F dinv = L::wide(oneR / diag);
            if (RT) m_diag = diag;
            B live = L::band(robot, L::nei(L::loadI(T.jtype), 0));
            R.m_dinv = L::lo(L::sel(live, dinv, zero));
            F verr = kp * (qdes - q) * inv_dt - kd * vstar;
            if (SH::MREC) {
                // setJointMotorControl2(maxVelocity=v): Bullet clamps the motor row's target velocity kp dq/dt + (1 - kd) v to +-v
                // (btMultiBodyJointMotor m_rhsClamp, set from maxVelocity [EXT-UNVERIFIED]); the velocity error is target - v*
                const F vt = L::fma(kp * (qdes - q), inv_dt, (one - kd) * vstar);
                verr = L::sel(L::gt(vmx, zero), clampf(vt, zero - vmx, vmx) - vstar, verr);
            }
            R.m_rhs = L::lo(L::sel(live, verr * dinv, zero));
            // joint limits (btMultiBodyJointLimitConstraint): row exists only while violated
            F pl = q - lower, pu = upper - q;
            B lo_v = L::band(live, L::le(pl, zero)), up_v = L::band(live, L::band(L::bnot(lo_v), L::le(pu, zero)));
            F dir = L::sel(lo_v, one, L::sel(up_v, zero - one, zero));
            F pen = L::sel(lo_v, pl, pu);
            R.l_dir = L::lo(dir);
            R.l_j = L::lo(dir * dinv);                     // J' = dir * dinv (dir^2 = 1 so dinv is unchanged)
            R.l_rhs = L::lo(L::sel(L::bor(lo_v, up_v), (zero - pen * L::c(P.erp) * inv_dt - dir * vstar) * dinv, zero));
            R.m_app = zeroR; R.l_app = zeroR;
            R.m_lim = LR::c(P.motor_imp) * fscale;         // setJointMotorControl `force` (default for every joint but grasping fingers)
            if constexpr (SH::W >= 64) {                   // (the floating-base model is a 26-DoF one: the 64-lane shapes)
                // the base constraint's rows (createConstraint(JOINT_FIXED), icub_env.py:95-101) are bounded by ITS maxForce, not by a motor's:
                // the virtual joints' hold motors carry it in the table (Tables::mforce)
                const FR mf = L::lo(L::load(T.mforce));
                R.m_lim = LR::sel(LR::gt(mf, zeroR), mf * LR::c(P.dt), R.m_lim);
            }
        }
        const BR any_limit = LR::ne(R.l_dir, zeroR);
        // contacts
        const F inv_m = one / o_m;
        PBRE_UNROLL for (int c = 0; c < NC; c++) {
            const int type = c < NC_OT ? 0 : (c < NC_OT + NC_RO ? 1 : 2);
            R.an[c] = zero; R.a1[c] = zero; R.a2[c] = zero;
            PBRE_UNROLL for (int d = 0; d < 3; d++) r_den[c][d] = zero;
            if (type == 0 && split_all) {
                R.act[c] = L::lti(lane, 0); R.mu[c] = L::uni(zero);
                PBRE_UNROLL for (int k = 0; k < 6; k++) R.rs.put(6 * c + k, zero);
                continue;
            }
            const Contact cc = contact_of(c);
            if (type == 1) owner_ro[c - NC_OT] = cc.owner;
            R.act[c] = cc.act; R.mu[c] = L::uni(L::sel(cc.act, cc.mu, zero));
            if (!L::any(cc.act)) { PBRE_UNROLL for (int k = 0; k < 6; k++) R.rs.put(6 * c + k, zero); continue; }
            // btPlaneSpace1
            V3 n = cc.n, t1, t2;
            {
                B big = L::gt(L::abs(n.z), L::c(0.70710678118654752f));
                F a1 = n.y*n.y + n.z*n.z, k1 = one / L::sqrt(L::max(a1, L::c(1e-30f)));
                V3 p1 = v3(zero, zero - n.z * k1, n.y * k1);
                V3 q1 = v3(a1 * k1, zero - n.x * p1.z, n.x * p1.y);
                F a2 = n.x*n.x + n.y*n.y, k2 = one / L::sqrt(L::max(a2, L::c(1e-30f)));
                V3 p2 = v3(zero - n.y * k2, n.x * k2, zero);
                V3 q2 = v3(zero - n.z * p2.y, n.z * p2.x, a2 * k2);
                t1 = selv(big, p1, p2); t2 = selv(big, q1, q2);
            }
            B onchain = L::band(robot, mbiti(dmask, cc.owner));   // this joint moves the contact link (the link is in its subtree)
            V3 rO = sub(type == 0 ? cc.pA : cc.pB, op);
            PBRE_UNROLL for (int d = 0; d < 3; d++) {
                const V3& dir = d == 0 ? n : (d == 1 ? t1 : t2);
                F J = zero;
                if (type != 0) {
                    F jr = dot(dir, add(K.S.l, cross(K.S.a, cc.pA)));
                    J = L::sel(onchain, jr, zero);
                }
                if (type != 2) {
                    V3 ra = cross(rO, dir);
                    F jo = L::sel(is_lin, pick3(dir, lane, LC), L::sel(is_ang, pick3(ra, lane, LC + 3), zero));
                    J = J + (type == 0 ? jo : zero - jo);
                }
                J = L::sel(cc.act, J, zero);
                // B = M^-1 J^T
                F Bv = zero;
                if (type != 0) {
                    const FR JR = L::lo(J);
                    FR Br = zeroR;
                    PBRE_UNROLL for (int j = 0; j < NJ; j++) Br = LR::fma(R.Mi[j], LR::bcast(JR, j), Br);
                    Bv = L::sel(robot, L::wide(Br), zero);
                }
                if (type != 2) {
                    V3 Ja = v3(L::bcast(J, LC + 3), L::bcast(J, LC + 4), L::bcast(J, LC + 5));
                    V3 Ba = mv(Iinv, Ja);
                    Bv = Bv + L::sel(is_lin, J * inv_m, L::sel(is_ang, pick3(Ba, lane, LC + 3), zero));
                }
                F denom = L::sum(J * Bv);
                F rel = L::sum(J * vstar);
                F dinv = L::sel(cc.act, one / L::sel(cc.act, denom, one), zero);
                r_den[c][d] = L::uni(L::sel(cc.act, denom, zero));
                F rhs;
                if (d == 0) {   // setupMultiBodyContactConstraint, restitution 0
                    F pen = cc.dist + L::c(P.slop);
                    B sep = L::gt(pen, zero);
                    F perr = L::sel(sep, zero, zero - pen * L::c(P.erp) * inv_dt);
                    F verr = L::sel(sep, zero - rel - pen * inv_dt, zero - rel);
                    rhs = (perr + verr) * dinv;
                } else rhs = (zero - rel) * dinv;
                F Jp = L::sel(L::eqi(lane, L1), zero - rhs, J * dinv);
                Jp = L::sel(cc.act, Jp, zero);
                R.rs.put(6 * c + 2 * d, Jp); R.rs.put(6 * c + 2 * d + 1, Bv);
            }
        }

        PBRE_PROBE(8);      // constraint rows
        // ---- projected Gauss-Seidel (Bullet order: non-contact rows alternate direction, normals, frictions)
        F dv = L::sel(L::eqi(lane, L1), one, zero);
        const F big = L::c(1e10f);
        const FR llim = LR::c(P.limit_imp), nmlim = zeroR - R.m_lim;
        // a motor / limit row touches one DoF only: every lane evaluates the row it owns, lane j's update is the one
        // applied (Gauss-Seidel order is kept by the sequence of calls)
        constexpr bool FREE_ROWS = SH::W == 32 || SH::MREC;       // iCub shapes: clamp-free motor rows first, see below (16-lane Panda rows: no gain)
        // RT bookkeeping: lane j of m_dsw / l_dsw holds the impulse change of joint j's motor / limit row in the current sweep, lsr the
        // largest |impulse change / dinv| of its contact rows so far (group-uniform)
        FR m_dsw = zeroR, l_dsw = zeroR;
        F lsr = zero;
        auto motor = [&](int j) {
            FR t = LR::fma(R.m_dinv, L::lo(dv), zeroR - R.m_rhs);
            FR s = LR::med3(R.m_app - t, nmlim, R.m_lim);
            FR d = s - R.m_app;
            R.m_app = LR::setlane(R.m_app, j, s);
            if constexpr (RT) m_dsw = LR::setlane(m_dsw, j, d);
            dv = L::fma_lo(LR::bcast(d, j), R.Mi[j], dv);
        };
        auto limit = [&](int j) {
            FR t = LR::fma(R.l_j, L::lo(dv), zeroR - R.l_rhs);
            FR s = LR::med3(R.l_app - t, zeroR, llim);
            FR d = s - R.l_app;
            R.l_app = LR::setlane(R.l_app, j, s);
            if constexpr (RT) l_dsw = LR::setlane(l_dsw, j, d);
            dv = L::fma_lo(LR::bcast(d * R.l_dir, j), R.Mi[j], dv);      // (bcast_row here: no gain with IK control, where limit rows are frequent)
        };
        auto res_of = [&](int c, int d, F dd) { if constexpr (RT) lsr = L::max(lsr, L::abs(dd * r_den[c][d])); };
        bool on[NC];                         // some group of the wave has contact c
        unsigned on_bits = 0u;               // the same as a scalar bit mask: tested with one SALU instruction per slot inside the loop
        PBRE_UNROLL for (int c = 0; c < NC; c++) { on[c] = L::any(R.act[c]); on_bits |= on[c] ? (1u << c) : 0u; }
        auto contacts = [&]() {
            PBRE_UNROLL for (int c = 0; c < NC; c++) if ((on_bits >> c) & 1u) {
                if (c < NC_OT) res_of(c, 0, row<true>(R.rs.get(6 * c), R.rs.get(6 * c + 1), R.an[c], zero, big, dv));
                else res_of(c, 0, row<false>(R.rs.get(6 * c), R.rs.get(6 * c + 1), R.an[c], zero, big, dv));
            }
            PBRE_UNROLL for (int c = 0; c < NC; c++) if ((on_bits >> c) & 1u) {
                F lim = R.mu[c] * R.an[c];
                if (c < NC_OT) {
                    res_of(c, 1, frow<true>(R.rs.get(6 * c + 2), R.rs.get(6 * c + 3), R.a1[c], lim, dv));
                    res_of(c, 2, frow<true>(R.rs.get(6 * c + 4), R.rs.get(6 * c + 5), R.a2[c], lim, dv));
                } else {
                    res_of(c, 1, frow<false>(R.rs.get(6 * c + 2), R.rs.get(6 * c + 3), R.a1[c], lim, dv));
                    res_of(c, 2, frow<false>(R.rs.get(6 * c + 4), R.rs.get(6 * c + 5), R.a2[c], lim, dv));
                }
            }
        };
        // bit j: joint j is at a limit in some group of the wave; the limit rows of all other joints are exact no-ops (J' = rhs = 0)
        // and are skipped (iCub, IK control: +22 %)
        const unsigned long long lim_bits = LR::lanebits(any_limit);
        const bool has_limit = lim_bits != 0ull;
        // sweeps over the limit rows: per joint on the small shapes; on the 60-DoF shape per block of 10 joints (there a scalar
        // test per row costs more than the skipped rows save, but whole idle blocks -- arms away from their limits while the
        // resting fingers sit on theirs -- are worth skipping)
        constexpr int LB = 10;
        auto limits_fwd = [&]() {
            if (NJ <= 40) { PBRE_UNROLL for (int j = 0; j < NJ; j++) if ((lim_bits >> j) & 1ull) limit(j); }
            else {
                PBRE_UNROLL for (int b0 = 0; b0 < NJ; b0 += LB)
                    if ((lim_bits >> b0) & ((1ull << LB) - 1ull)) { PBRE_UNROLL for (int j = b0; j < b0 + LB && j < NJ; j++) limit(j); }
            }
        };
        auto limits_bwd = [&]() {
            if (NJ <= 40) { PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) if ((lim_bits >> j) & 1ull) limit(j); }
            else {
                PBRE_UNROLL for (int b0 = ((NJ - 1) / LB) * LB; b0 >= 0; b0 -= LB)
                    if ((lim_bits >> b0) & ((1ull << LB) - 1ull)) { PBRE_UNROLL for (int j = (b0 + LB < NJ ? b0 + LB : NJ) - 1; j >= b0; j--) limit(j); }
            }
        };
        // The usual wave: the object rests on the table with all four object-table slots in use and the robot touches nothing.
        // That case gets its own copy of the loop without the per-slot branches (and their mask bookkeeping); rows of a group
        // that lacks one of the contacts are exact no-ops either way.
        bool only_ot = true;
        PBRE_UNROLL for (int c = 0; c < NC; c++) only_ot = only_ot && (on[c] == (c < NC_OT));
        auto contacts_ot = [&]() {
            PBRE_UNROLL for (int c = 0; c < NC_OT; c++) res_of(c, 0, row<true>(R.rs.get(6 * c), R.rs.get(6 * c + 1), R.an[c], zero, big, dv));
            PBRE_UNROLL for (int c = 0; c < NC_OT; c++) {
                F lim = R.mu[c] * R.an[c];
                res_of(c, 1, frow<true>(R.rs.get(6 * c + 2), R.rs.get(6 * c + 3), R.a1[c], lim, dv));
                res_of(c, 2, frow<true>(R.rs.get(6 * c + 4), R.rs.get(6 * c + 5), R.a2[c], lim, dv));
            }
        };
        // RT: end of sweep `it` (cur: the velocity vector as it stands).  Bullet's test, btSequentialImpulseConstraintSolver::
        // solveGroupCacheFriendlyIterations: leastSquaresResidual <= m_leastSquaresResidualThreshold (both squared there).  A group that
        // passes keeps a snapshot of its vector (and, where fingertip forces are reported, of its robot-object normal impulses); true once
        // every group of the wave is through.
        B rt_done = L::bfalse();
        F rt_dv = zero, rt_used = L::c((float)P.iters);
        F rt_an[NC_RO];
        PBRE_UNROLL for (int c = 0; c < NC_RO; c++) rt_an[c] = zero;
        auto sweep_end = [&](int it, F cur) -> bool {
            const FR mres = LR::max(LR::abs(m_dsw * m_diag), LR::abs(l_dsw * m_diag));
            const F res = L::max(L::sel(robot, L::wide(mres), zero), L::uni(lsr));
            const F g = zero - L::vmin(zero - res);
            const B newly = L::band(L::bnot(rt_done), L::le(g, L::c(P.res_lim)));
            if (L::any(newly)) {
                rt_dv = L::sel(newly, cur, rt_dv);
                rt_used = L::sel(newly, L::c((float)(it + 1)), rt_used);
                if (NTIP > 0) { PBRE_UNROLL for (int c = 0; c < NC_RO; c++) rt_an[c] = L::sel(newly, R.an[NC_OT + c], rt_an[c]); }
                rt_done = L::bor(rt_done, newly);
            }
            m_dsw = zeroR; l_dsw = zeroR; lsr = zero;
            return !L::any(L::bnot(rt_done));
        };
        // iCub shapes: in waves without robot contact rows the motor rows first run WITHOUT their clamp.  While a motor stays
        // inside its impulse bound (PyBullet's default force of 1e5 N is far out of these robots' reach) Bullet's row is
        // delta = rhs' - dinv dv_j, applied += delta: one fma whose result goes straight to the broadcast, instead of the
        // fma - sub - med3 - sub chain that bounds this loop (three waves per SIMD, each row waiting for the previous one's broadcast).
        // The applied impulses are still accumulated, off that chain, and tested against the bound after every sweep (a motor's
        // impulse only changes in its own row, so every value it takes is seen); if one ever leaves the bound the solve is repeated
        // with clamping rows.  Those (motor_x) return exactly the same delta for a row whose clamp does not bind, so what an env
        // computes does not depend on which of the two paths its wave took.
        // clamping row in delta form: clamp(applied + delta) - applied = clamp(delta, lo - applied, hi - applied); an unclamped row
        // returns delta itself, bit for bit, and the two bounds are off the row-to-row chain (fma -> med3 -> broadcast -> fma)
        const FR m_ndinv_x = zeroR - R.m_dinv;
        auto motor_x = [&](int j) {
            FR nt = LR::fma(m_ndinv_x, L::lo(dv), R.m_rhs);
            FR d = LR::med3(nt, nmlim - R.m_app, R.m_lim - R.m_app);
            R.m_app = LR::setlane(R.m_app, j, R.m_app + d);
            if constexpr (RT) m_dsw = LR::setlane(m_dsw, j, d);
            dv = L::fma_lo(LR::bcast(d, j), R.Mi[j], dv);
        };
        // (a shape with one env per wave has no neighbour to be independent of: its clamping rows stay the plain ones)
        auto mrow = [&](int j) { if (SH::W <= 32) motor_x(j); else motor(j); };
        bool solved = false;
        // (not attempted while some motor of the wave is force-limited -- grasping fingers, force 10: those do reach their bound)
        if (!RT && FREE_ROWS && (only_ot || on_bits == 0u) && !LR::any(LR::lt(R.m_lim, LR::c(P.motor_imp)))) {
            const FR m_ndinv = zeroR - R.m_dinv;
            bool over = false;
            FR dsel = zeroR;              // lane j: the delta of row j in the current sweep
            auto motor_free = [&](int j) {
                FR d = LR::fma(m_ndinv, L::lo(dv), R.m_rhs);
                dsel = LR::setlane(dsel, j, d);
                dv = L::fma_lo(LR::bcast_row(d, j), R.Mi[j], dv);
            };
            auto in_bound = [&]() {       // end of a sweep over the motor rows
                R.m_app = R.m_app + dsel;
                return !LR::any(LR::bnot(LR::le(LR::abs(R.m_app), R.m_lim)));      // (a NaN fails the test as well)
            };
            if (only_ot) {
                for (int it = 0; it < P.iters; it += 2) {
                    PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) motor_free(j);
                    if (!in_bound()) { over = true; break; }
                    if (has_limit) limits_bwd();
                    contacts_ot();
                    if (it + 1 >= P.iters) break;
                    if (has_limit) limits_fwd();
                    PBRE_UNROLL for (int j = 0; j < NJ; j++) motor_free(j);
                    if (!in_bound()) { over = true; break; }
                    contacts_ot();
                }
            } else {
                for (int it = 0; it < P.iters; it += 2) {
                    PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) motor_free(j);
                    if (!in_bound()) { over = true; break; }
                    if (has_limit) limits_bwd();
                    if (it + 1 >= P.iters) break;
                    if (has_limit) limits_fwd();
                    PBRE_UNROLL for (int j = 0; j < NJ; j++) motor_free(j);
                    if (!in_bound()) { over = true; break; }
                }
            }
            solved = !over;
            if (over) {             // start over with clamping rows
                dv = L::sel(L::eqi(lane, L1), one, zero);
                R.m_app = zeroR; R.l_app = zeroR;
                PBRE_UNROLL for (int c = 0; c < NC; c++) { R.an[c] = zero; R.a1[c] = zero; R.a2[c] = zero; }
            }
        }
        // 16-lane rows (the Panda's complex envs, k_row_list / k_step): the solver loop is one wave's serial chain -- every row waits for the
        // previous row's update of dv -- and that chain is what a stationary batch's step waits for.  But rows that share no
        // unknown commute exactly: motor / limit / robot-table rows only touch the robot lanes of dv, object-table rows only the object
        // lanes (J' and B of a row are exact zeros on the other body's lanes).  So the velocity vector is kept as two registers, dvr (robot
        // lanes) and dvo (object lanes), both with the constant-one lane, and a sweep runs a robot-only and an object-only group of rows
        // as two chains whose operations are issued in turn (`zip`: the compiler keeps the source order, a lone wave then issues one
        // chain's operation while the other's is in flight); robot-object rows run on the merged vector in between.  Bullet's order
        //     M L | OTn ROn RTn | OTf ROf RTf        becomes        (RTf' M || OTn) L ROn (RTn || OTf) ROf
        // (RTf': the robot-table friction rows of the previous sweep; odd sweeps: L before M): the same arithmetic on the same values row
        // by row, bit for bit (tests: the two orders compared on contact-rich states).
#ifndef PBRE_TWO_CHAIN
#define PBRE_TWO_CHAIN 1
#endif
#ifndef PBRE_ROBOT_ONLY_CHAIN      // (0: A/B -- such waves take the loops with per-slot tests below, as before round 4)
#define PBRE_ROBOT_ONLY_CHAIN 1
#endif
        constexpr bool TWO_CHAIN = PBRE_TWO_CHAIN && SH::W == 16 && !SH::MREC;
        bool solved2 = false, robot_only_path = false;
        if constexpr (TWO_CHAIN) {
        const bool ot_all = (on_bits & ((1u << NC_OT) - 1u)) == ((1u << NC_OT) - 1u);
        // (with `objv`: a wave none of whose groups has a robot-object contact has not built the object-table rows -- ot_all is false and it
        // sweeps its robot rows alone below; a wave with such a group runs the zipped sweeps over everything, as without `objv`)
        // robot_only: no object row of any kind in the wave -- the object is another wave's (k_row_list's object wave: every group of this
        // wave is free of robot-object contact) or absent (PBRE_F_NO_OBJECT).  The sweep is then the robot chain alone, M L RTn RTf, as
        // straight-line code from the same stage functions (the loop with per-slot tests spent a third of its time in its branches: a
        // lone wave's dependent instruction takes ~4.5 cycles, a scalar test + branch between two rows ~15).
        const bool robot_only = !solved && (on_bits & ((1u << (NC_OT + NC_RO)) - 1u)) == 0u && PBRE_ROBOT_ONLY_CHAIN;
        // (round 5: a wave WITH robot-object rows takes the zipped sweeps whether or not all four object-table slots are in use -- a pushed
        // cube tips onto an edge, 2-3 contacts -- : the rows of an unused slot are exact no-ops (J' = B = 0), and the phase probe had such
        // waves, more than one per step at 131072 envs, on the loops with per-slot tests at 295 k cycles against the zipped 269 k)
        const bool with_ro = ((on_bits >> NC_OT) & ((1u << NC_RO) - 1u)) != 0u;
        if (!solved && (ot_all || robot_only || with_ro)) {
            solved2 = true;
            const unsigned ro_bits = (on_bits >> NC_OT) & ((1u << NC_RO) - 1u), rt_bits = (on_bits >> (NC_OT + NC_RO)) & ((1u << NC_RT) - 1u);
#ifdef PBRE_TWO_CHAIN_TRACE
            PBRE_TWO_CHAIN_TRACE(ro_bits, rt_bits, has_limit);      // (host test builds: which row patterns took this path)
#endif
            F dvr = dv, dvo = dv;         // 1 on the constant-one lane, 0 elsewhere
            // ---- a motor row (motor_x) in 3 stages on dvr
            // Round 6, last session: the row as few INSTRUCTIONS as it can be -- a lone wave issues one per ~4.5 cycles whether or not it depends
            // on its predecessor (profiles/r06_ubench_valu.txt), so the bookkeeping "off the chain" cost as much as the chain.  The two clamp
            // bounds are taken once per sweep (motors_begin: a motor's applied impulse only changes in its own row, once per sweep), the
            // deltas are collected in m_dsel and added to the applied impulses once per sweep (motors_end), and the row's tail -- record the
            // delta, broadcast it, update the vector -- is L::row_tail's three instructions (v_fmac with a DPP row broadcast as its source).
            // Same operations on the same operands as before: bit-identical rows (GPU tests: the plain loops against these paths).
            FR m_nt = zeroR, m_d = zeroR, m_lo = zeroR, m_hi = zeroR;
            // FREE (round 5): the same row WITHOUT its clamp, in 2 stages -- delta = rhs' - dinv dv_j goes straight to the broadcast.  While no
            // motor reaches its impulse bound (PyBullet's default force, 1e5 N, is far out of reach) this is the clamping row's delta bit for
            // bit (a med3 whose bounds do not bind returns its first operand), at 4 instructions instead of 6 on the wave's serial chain.  The
            // deltas of a sweep are collected in m_dsel, added to the applied impulses after the sweep's motor rows and tested against the
            // bound there (a motor's impulse only changes in its own row: every value it takes is seen); if one ever leaves it, the solve
            // starts over with the clamping stages -- same mechanism as the wide shapes' FREE_ROWS above.
            FR m_dsel = zeroR;
            auto mstage = [&](auto jc, auto sc, auto freec) {
                constexpr int j = decltype(jc)::value, st = decltype(sc)::value;
                constexpr bool FREE = decltype(freec)::value;
                if constexpr (FREE) {
                    if constexpr (st == 0) m_nt = LR::fma(m_ndinv_x, L::lo(dvr), R.m_rhs);
                    else dvr = L::template row_tail<j>(m_dsel, m_nt, m_nt, R.Mi[j], dvr);
                } else {
                if constexpr (st == 0) m_nt = LR::fma(m_ndinv_x, L::lo(dvr), R.m_rhs);
                else if constexpr (st == 1) m_d = LR::med3(m_nt, m_lo, m_hi);
                else dvr = L::template row_tail<j>(m_dsel, m_d, m_d, R.Mi[j], dvr);
                }
            };
            // before / after the motor rows of a sweep on the clamping stages (the clamp-free ones: free_in_bound)
            auto motors_begin = [&]() { m_lo = nmlim - R.m_app; m_hi = R.m_lim - R.m_app; };
            auto motors_end = [&]() {
                R.m_app = R.m_app + m_dsel;
                if constexpr (RT) m_dsw = m_dsel;      // (the sweep's motor deltas, for the residual test at its end)
            };
            // end of a sweep's clamp-free motor rows: false if an applied impulse has left its bound (a NaN fails the test too)
            auto free_in_bound = [&]() {
                if constexpr (RT) m_dsw = m_dsel;      // (the sweep's motor deltas, for the residual test at its end)
                R.m_app = R.m_app + m_dsel;
                return !LR::any(LR::bnot(LR::le(LR::abs(R.m_app), R.m_lim)));
            };
            // ... and (round 6, PBRE_FREE_SWITCH) true while every applied impulse is still far inside it.  A wave that starts over is the longest of
            // its step at TWICE the chain, and 27 % of the stationary steps at 131072 envs have one (tools/wave_trace.py, profiles/r06zf_wave_trace.txt;
            // r06ze_tail_ab.txt: k_fused max 290 us against 238, mean 166.1 against 161.9, with the clamp-free stages off altogether).  With the switch
            // a wave leaves the clamp-free stages for the clamping ones at the next boundary between sweep pairs once an impulse is past
            // PBRE_FREE_SWITCH_FRAC of its bound -- nothing has been clamped yet, so the clamp-free rows so far ARE the clamping rows bit for bit and
            // the sweeps simply go on; starting over is left for an impulse that jumps past the bound within one pair.  Measured, the two cancel.
            auto free_far_inside = [&]() { return !LR::any(LR::bnot(LR::le(LR::abs(R.m_app), R.m_lim * LR::c(PBRE_FREE_SWITCH_FRAC)))); };
            auto limit2 = [&](int j) {
                FR t = LR::fma(R.l_j, L::lo(dvr), zeroR - R.l_rhs);
                FR s_ = LR::med3(R.l_app - t, zeroR, llim);
                FR d = s_ - R.l_app;
                R.l_app = LR::setlane(R.l_app, j, s_);
                if constexpr (RT) l_dsw = LR::setlane(l_dsw, j, d);
                dvr = L::fma_lo(LR::bcast(d * R.l_dir, j), R.Mi[j], dvr);
            };
            // ---- a contact row (row<> / frow<>) in 9 stages on one of the vectors; Fly: the row in flight
            struct Fly { F p, s, lo, hi, res; };      // (res, RT: the chain's largest |impulse change / dinv| of the sweep)
            Fly fr, fo;                   // robot chain / object chain
            fr.res = zero; fo.res = zero;
            // ROW: index of the row's J' in the row store (B follows); FRIC: friction row (bound +-lim, skipped while the normal impulse is 0)
            // OBJ: an object-table row.  Its J' is non-zero on lanes LC..L1 only, all in the upper half of the 16-lane row (LC >= 8), so
            // three butterfly stages already leave the whole sum on those lanes (the fourth would add the lower half's exact zero); the
            // lower lanes then carry 0 through the row -- their impulse stays 0 and B is 0 there -- which costs one stage less: NSO stages
            auto cstage = [&](Fly& f, F& vec, auto rowc, F& app, auto sc, auto fricc, auto objc, F lim) {
                constexpr int ROW = decltype(rowc)::value, st0 = decltype(sc)::value;
                constexpr bool FRIC = decltype(fricc)::value, OBJ = decltype(objc)::value;
                constexpr int st = (OBJ && st0 >= 4) ? st0 + 1 : st0;         // OBJ rows skip stage 4 (row_mirror)
                if constexpr (st == 0) {
                    if (FRIC) {
                        // PBRE_FRIC_FOLD: "skipped while the normal impulse is 0" as the row's BOUNDS -- med3(x, app, app) = app, the value the select
                        // behind the clamp returned -- taken here, off the row's chain of dependent instructions (one link less per friction row:
                        // a lone wave's dependent instruction costs 8.25 cycles, profiles/r06_ubench_valu.txt)
                        if (PBRE_FRIC_FOLD) { const B on_ = L::gt(lim, zero); f.lo = L::sel(on_, zero - lim, app); f.hi = L::sel(on_, lim, app); }
                        else { f.lo = zero - lim; f.hi = lim; }
                    } else { f.lo = zero; f.hi = big; }
                    f.p = R.rs.get(ROW) * vec;
                }
                else if constexpr (st <= 4) f.p = L::sum_step(f.p, st - 1);
                else if constexpr (st == 5) f.s = app - f.p;
                else if constexpr (st == 6) { f.s = L::med3(f.s, f.lo, f.hi); if (FRIC && !PBRE_FRIC_FOLD) f.s = L::sel(L::gt(f.hi, zero), f.s, app); }
                else if constexpr (st == 7) { f.p = f.s - app; app = f.s; if constexpr (RT) f.res = L::max(f.res, L::abs(f.p * r_den[ROW / 6][(ROW % 6) / 2])); }
                else vec = L::fma(f.p, R.rs.get(ROW + 1), vec);
            };
            static_assert(LC >= 8 && L1 < 16, "object lanes in the upper half of the row");
            constexpr int NSR = 9, NSO = 8;       // stages of a contact row on the robot chain / of an object-table row
            // normal row of contact C / friction row D of contact C, stage ST
            auto nstage = [&](Fly& f, F& vec, auto cc, auto sc) {
                constexpr int C = decltype(cc)::value;
                cstage(f, vec, std::integral_constant<int, 6 * C>{}, R.an[C], sc, std::false_type{}, std::integral_constant<bool, (C < NC_OT)>{}, zero);
            };
            auto fstage = [&](Fly& f, F& vec, auto cc, auto dc, auto sc) {
                constexpr int C = decltype(cc)::value, D = decltype(dc)::value;
                F lim = zero;
                if constexpr (decltype(sc)::value == 0) lim = R.mu[C] * R.an[C];
                cstage(f, vec, std::integral_constant<int, 6 * C + 2 + 2 * D>{}, D == 0 ? R.a1[C] : R.a2[C], sc, std::true_type{}, std::integral_constant<bool, (C < NC_OT)>{}, lim);
            };
            constexpr int NRT0 = NC_OT + NC_RO;
            // (RTf' M || OTn): robot stream = [2 NC_RT friction rows of the previous sweep] + NJ motor rows, object stream = NC_OT normal rows
            auto phase_a = [&](auto rev_c, auto e_c, auto freec) {
                constexpr bool REV = decltype(rev_c)::value, WITH_E = decltype(e_c)::value;
                constexpr int MS = decltype(freec)::value ? 2 : 3;       // stages of a motor row
                constexpr int NE = WITH_E ? 2 * NC_RT * NSR : 0, NR = NE + MS * NJ, NO = NSO * NC_OT, NZ = NR > NO ? NR : NO;
                if constexpr (!decltype(freec)::value) motors_begin();
                for_seq<NZ>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < NE) fstage(fr, dvr, std::integral_constant<int, NRT0 + (k / NSR) / 2>{}, std::integral_constant<int, (k / NSR) % 2>{}, std::integral_constant<int, k % NSR>{});
                    else if constexpr (k < NR) { constexpr int r = (k - NE) / MS; mstage(std::integral_constant<int, (REV ? NJ - 1 - r : r)>{}, std::integral_constant<int, (k - NE) % MS>{}, freec); }
                    if constexpr (k < NO) nstage(fo, dvo, std::integral_constant<int, k / NSO>{}, std::integral_constant<int, k % NSO>{});
                });
                if constexpr (!decltype(freec)::value) motors_end();
            };
            // (RTn || OTf)
            auto phase_c = [&](auto rt_c) {
                constexpr bool RTAB = decltype(rt_c)::value;
                constexpr int NR = RTAB ? NSR * NC_RT : 0, NO = NSO * 2 * NC_OT, NZ = NR > NO ? NR : NO;
                for_seq<NZ>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if constexpr (k < NR) nstage(fr, dvr, std::integral_constant<int, NRT0 + k / NSR>{}, std::integral_constant<int, k % NSR>{});
                    if constexpr (k < NO) fstage(fo, dvo, std::integral_constant<int, (k / NSO) / 2>{}, std::integral_constant<int, (k / NSO) % 2>{}, std::integral_constant<int, k % NSO>{});
                });
            };
            // NRO > 0: the wave uses exactly the first NRO robot-object slots (slots fill in rank order) -- no per-slot scalar test and branch
            // between the rows (a test + branch costs a lone wave ~15 cycles, three dependent instructions' worth); 0: test every slot
            auto coupled = [&](bool fric, auto nroc) {       // robot-object rows on the merged vector
                constexpr int NRO = decltype(nroc)::value;
                F dvc = L::sel(obj_lane, dvo, dvr);
                PBRE_UNROLL for (int c = NC_OT; c < NC_OT + (NRO ? NRO : NC_RO); c++) if (NRO || ((on_bits >> c) & 1u)) {
                    if (!fric) res_of(c, 0, row<false>(R.rs.get(6 * c), R.rs.get(6 * c + 1), R.an[c], zero, big, dvc));
                    else {
                        F lim = R.mu[c] * R.an[c];
                        res_of(c, 1, frow<false>(R.rs.get(6 * c + 2), R.rs.get(6 * c + 3), R.a1[c], lim, dvc));
                        res_of(c, 2, frow<false>(R.rs.get(6 * c + 4), R.rs.get(6 * c + 5), R.a2[c], lim, dvc));
                    }
                }
                dvr = L::sel(obj_lane, zero, dvc); dvo = L::sel(robot, zero, dvc);
            };
            // (the rows of a robot-table slot no group of the wave uses are exact no-ops: J' = B = 0)
            // RT: end of a sweep of the two chains (their residuals merged into lsr, the vector as it stands: object lanes from dvo)
            auto chains_end = [&](int it) -> bool {
                lsr = L::max(lsr, L::max(fr.res, fo.res)); fr.res = zero; fo.res = zero;
                return sweep_end(it, L::sel(obj_lane, dvo, dvr));
            };
            auto rt_f = [&]() { for_seq<2 * NC_RT * NSR>([&](auto kc) { constexpr int k = decltype(kc)::value; fstage(fr, dvr, std::integral_constant<int, NRT0 + (k / NSR) / 2>{}, std::integral_constant<int, (k / NSR) % 2>{}, std::integral_constant<int, k % NSR>{}); }); };
            // the sweeps; FREE: with the clamp-free motor stages -- returns false if a motor impulse left its bound on the way (the caller
            // then starts over with the clamping stages)
            auto run_chains = [&](auto freec, auto nroc) -> bool {
                constexpr bool FREE = decltype(freec)::value;
                constexpr int MS = FREE ? 2 : 3;
                constexpr int NRO = decltype(nroc)::value;
            if (NRO == 0 && robot_only) {
                robot_only_path = true;
                auto motors = [&](auto rev_c) {
                    constexpr bool REV = decltype(rev_c)::value;
                    if constexpr (!FREE) motors_begin();
                    for_seq<MS * NJ>([&](auto kc) { constexpr int k = decltype(kc)::value; mstage(std::integral_constant<int, (REV ? NJ - 1 - k / MS : k / MS)>{}, std::integral_constant<int, k % MS>{}, freec); });
                    if constexpr (!FREE) motors_end();
                };
                auto rt_n = [&]() { for_seq<NC_RT * NSR>([&](auto kc) { constexpr int k = decltype(kc)::value; nstage(fr, dvr, std::integral_constant<int, NRT0 + k / NSR>{}, std::integral_constant<int, k % NSR>{}); }); };
                // LJ >= 0: joint LJ is the only joint of the wave at a limit (PBRE_RO_LIMIT_SPECIAL) -- its row inlined, no per-joint tests; -1: test every joint
                auto ro_sweeps = [&](auto ljc) -> bool {
                    constexpr int LJ = decltype(ljc)::value;
                    for (int it = 0; it < P.iters; it += 2) {
                        motors(std::true_type{});
                        if constexpr (FREE) { if (!free_in_bound()) return false; }
                        if constexpr (LJ >= 0) limit2(LJ);
                        else if (has_limit) { PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) if ((lim_bits >> j) & 1ull) limit2(j); }
                        if (rt_bits) { rt_n(); rt_f(); }
                        if constexpr (RT) { if (chains_end(it)) break; }
                        if (it + 1 >= P.iters) break;
                        if constexpr (LJ >= 0) limit2(LJ);
                        else if (has_limit) { PBRE_UNROLL for (int j = 0; j < NJ; j++) if ((lim_bits >> j) & 1ull) limit2(j); }
                        motors(std::false_type{});
                        if constexpr (FREE) { if (!free_in_bound()) return false; }
                        if (rt_bits) { rt_n(); rt_f(); }
                        if constexpr (RT) { if (chains_end(it + 1)) break; }
                    }
                    return true;
                };
                bool ran = false, ok = true;
                if constexpr (PBRE_RO_LIMIT_SPECIAL && !FREE) {
                    const int lj = __builtin_popcountll(lim_bits) == 1 ? __builtin_ctzll(lim_bits) : -1;
                    for_seq<NJ>([&](auto jc) { if (!ran && lj == decltype(jc)::value) { ok = ro_sweeps(jc); ran = true; } });
                }
                if (!ran) ok = ro_sweeps(std::integral_constant<int, -1>{});
                if (!ok) return false;
                dv = dvr;
            } else {
            const bool e_zip = !RT && rt_bits != 0u && !has_limit;       // the RTf rows ride along with the next sweep's motor rows (RT: a sweep ends where Bullet's does)
            auto mid = [&]() {            // the rest of a sweep after its motor / limit / OT-normal rows
                if (NRO || ro_bits) coupled(false, nroc);
                if (rt_bits) phase_c(std::true_type{}); else phase_c(std::false_type{});
                if (NRO || ro_bits) coupled(true, nroc);
                if (rt_bits && !e_zip) rt_f();
            };
            // the sweeps from `it0` (even) on, in pairs, with the clamp-free (fc = true) or the clamping motor stages.  Returns 0: done (all sweeps, or
            // the residual exit), 1: clamp-free stages only -- an applied impulse is past PBRE_FREE_SWITCH_FRAC of its bound: go on from `it0` (updated)
            // with the clamping stages, 2: clamp-free stages only -- an impulse has left its bound: start over.  (Two loops, one per kind of stage,
            // and NOT one loop with a test per phase: a branch between the phases splits the basic block the two chains' stages are interleaved in --
            // measured, a coupled wave's sweeps 173 -> 197 us, profiles/r06zi_wave_trace_switch.txt.)
            auto pairs = [&](auto fc, int& it0) -> int {
                constexpr bool FC = decltype(fc)::value;
                for (int it = it0; it < P.iters; it += 2) {
                    if (e_zip && it > 0) phase_a(std::true_type{}, std::true_type{}, fc); else phase_a(std::true_type{}, std::false_type{}, fc);
                    if constexpr (FC) { if (!free_in_bound()) return 2; }
                    if (has_limit) { PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) if ((lim_bits >> j) & 1ull) limit2(j); }
                    mid();
                    if constexpr (RT) { if (chains_end(it)) return 0; }
                    if (it + 1 >= P.iters) return 0;
                    if (has_limit) {          // odd sweep: limits first, then the motors in forward order
                        PBRE_UNROLL for (int j = 0; j < NJ; j++) if ((lim_bits >> j) & 1ull) limit2(j);
                    }
                    if (e_zip) phase_a(std::false_type{}, std::true_type{}, fc); else phase_a(std::false_type{}, std::false_type{}, fc);
                    if constexpr (FC) { if (!free_in_bound()) return 2; }
                    mid();
                    if constexpr (RT) { if (chains_end(it + 1)) return 0; }
                    if constexpr (FC) { if (PBRE_FREE_SWITCH && !free_far_inside()) { it0 = it + 2; PBRE_PROBE_PATH(10); return 1; } }
                }
                return 0;
            };
            int it0 = 0;
            if constexpr (FREE) {
                const int r = pairs(std::true_type{}, it0);
                if (r == 2) return false;
                if (r == 1) (void)pairs(std::false_type{}, it0);
            } else (void)pairs(std::false_type{}, it0);
            if (e_zip) rt_f();            // the last sweep's
            dv = L::sel(obj_lane, dvo, dvr);
            }
                return true;
            };
#ifndef PBRE_RT_FREE_STAGES          // 1: the clamp-free motor stages under the residual exit too.  Bit-identical (checksums of 300 steps at 16384 and
#define PBRE_RT_FREE_STAGES 0        // 131072 envs) and no faster: 0.2207 against 0.2169 ms per step at 131072 envs, 0.1913 / 0.1910 at 16384
#endif                               // (profiles/r06zc_rt_ab.txt) -- the RT row waves' time is not their motor rows.  Off.
#ifndef PBRE_FREE_MOTOR_STAGES      // 0: always the clamping stages (the default since the round-6 diet of the clamping row: 5 instructions on a 3-link chain against the
#define PBRE_FREE_MOTOR_STAGES 0      // clamp-free row's 4 on 2 -- what is left to gain no longer pays for the waves that start over: 131072 envs stationary 0.1490 -> 0.1447 ms,
#endif                                // 16384 envs 0.0987 -> 0.0983, three interleaved runs, profiles/r06u_free_stages_ab.txt); 2: clamp-free first in EVERY two-chain wave;
                                      // 1 (rounds 5-6): clamp-free first in the waves with robot-object rows only (measured then, profiles/r05_chain_ab4.txt and
                                      // r05_phase_probe_free_motor_stages.txt: a coupled env's sweeps get 17 % shorter and 2 % of those waves start over; among the
                                      // envs WITHOUT robot-object contact 12 % start over -- an arm pressed onto the table drives a blocked position motor
                                      // into its bound within the 150 sweeps -- and a wave that starts over is the longest of its step: 16384 envs 0.133 -> 0.140 ms)
            // clamp-free first unless a motor of the wave is force-limited (those do reach their bound) or the residual exit is on
            bool done2 = false;
            // (the usual coupled wave: one or two robot-object slots in use)
#ifndef PBRE_NRO_SPECIAL      // (0: A/B -- per-slot tests in every wave)
#define PBRE_NRO_SPECIAL 1
#endif
            auto run_n = [&](auto freec) -> bool {
                if (PBRE_NRO_SPECIAL && !robot_only && ro_bits == 1u) return run_chains(freec, std::integral_constant<int, 1>{});
                if (PBRE_NRO_SPECIAL && !robot_only && ro_bits == 3u) return run_chains(freec, std::integral_constant<int, 2>{});
                return run_chains(freec, std::integral_constant<int, 0>{});
            };
            // (round 6, PBRE_RT_FREE_STAGES: also under the residual exit -- a group's snapshot is taken from rows that returned the clamping rows'
            // deltas bit for bit, and a wave that starts over forgets its snapshots: what a group computes does not depend on the path its wave took)
            if (PBRE_FREE_MOTOR_STAGES && (PBRE_FREE_MOTOR_STAGES == 2 || (!robot_only && ro_bits != 0u)) && (!RT || PBRE_RT_FREE_STAGES) && !LR::any(LR::lt(R.m_lim, LR::c(P.motor_imp)))) {
                done2 = run_n(std::true_type{});
                if (!done2) {             // start over with clamping rows
                    PBRE_PROBE_PATH(11);
                    dvr = dv; dvo = dv; m_dsel = zeroR;
                    R.m_app = zeroR; R.l_app = zeroR;
                    PBRE_UNROLL for (int c = 0; c < NC; c++) { R.an[c] = zero; R.a1[c] = zero; R.a2[c] = zero; }
                    if constexpr (RT) {
                        rt_done = L::bfalse(); rt_dv = zero; rt_used = L::c((float)P.iters);
                        PBRE_UNROLL for (int c = 0; c < NC_RO; c++) rt_an[c] = zero;
                        m_dsw = zeroR; l_dsw = zeroR; lsr = zero; fr.res = zero; fo.res = zero;
                    }
                }
            }
            if (!done2) (void)run_n(std::false_type{});
        }
        }
        if (solved || solved2) {
        } else if (only_ot) {
            for (int it = 0; it < P.iters; it += 2) {
                PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) mrow(j);
                if (has_limit) limits_bwd();
                contacts_ot();
                if constexpr (RT) { if (sweep_end(it, dv)) break; }
                if (it + 1 >= P.iters) break;
                if (has_limit) limits_fwd();
                PBRE_UNROLL for (int j = 0; j < NJ; j++) mrow(j);
                contacts_ot();
                if constexpr (RT) { if (sweep_end(it + 1, dv)) break; }
            }
        } else if (on_bits == 0u) {      // no contact row in the wave (the object rows are kw_obj's, or there is no object)
            for (int it = 0; it < P.iters; it += 2) {
                PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) mrow(j);
                if (has_limit) limits_bwd();
                if constexpr (RT) { if (sweep_end(it, dv)) break; }
                if (it + 1 >= P.iters) break;
                if (has_limit) limits_fwd();
                PBRE_UNROLL for (int j = 0; j < NJ; j++) mrow(j);
                if constexpr (RT) { if (sweep_end(it + 1, dv)) break; }
            }
        } else
        for (int it = 0; it < P.iters; it += 2) {
            // even iteration: reversed non-contact order
            PBRE_UNROLL for (int j = NJ - 1; j >= 0; j--) mrow(j);
            if (has_limit) limits_bwd();
            contacts();
            if constexpr (RT) { if (sweep_end(it, dv)) break; }
            if (it + 1 >= P.iters) break;
            // odd iteration: forward order
            if (has_limit) limits_fwd();
            PBRE_UNROLL for (int j = 0; j < NJ; j++) mrow(j);
            contacts();
            if constexpr (RT) { if (sweep_end(it + 1, dv)) break; }
        }
        if constexpr (RT) {           // a group that left the loop early: its snapshot
            dv = L::sel(rt_done, rt_dv, dv);
            if (NTIP > 0) { PBRE_UNROLL for (int c = 0; c < NC_RO; c++) R.an[NC_OT + c] = L::sel(rt_done, rt_an[c], R.an[NC_OT + c]); }
            if (sw && L::lane0()) *sw = (int)L::first(rt_used);
        }

        PBRE_PROBE_PATH(solved ? 12 : (solved2 ? (robot_only_path ? 13 : 14) : 15));      // clamp-free / robot-only chain / two zipped chains / loops with per-slot tests
#ifdef PBRE_TRACE_ROWS
        PBRE_TRACE_ROWS(on_bits, has_limit);      // (tools/wave_trace.py builds: what the wave's sweeps were made of)
#endif
        PBRE_PROBE(9);      // the sweeps
        // ---- velocity + position update (semi-implicit Euler; quaternion exponential map for the object)
        F vnew = clampf(vstar + L::sel(L::eqi(lane, L1), zero, dv), zero - vmax, vmax);
        if (use_objv) {
            float poison = 0.f;      // NaN when a bounded wait for the side record ran out (k_fused's 64-thread grid): the env-step goes to the NaN / Inf guard
            PBRE_OBJV_SYNC(poison);
            vnew = L::sel(L::band(split, obj_lane), L::load(objv), vnew);
            if (poison != 0.f) vnew = L::c(poison);
        }
        B dyn = obj_on ? L::bor(robot, obj_lane) : robot;
        F Vn = L::sel(dyn, vnew, Vr);
        B posl = obj_on ? L::lti(lane, LC + 3) : robot;
        F Qn = L::sel(posl, L::fma(dt, vnew, Qr), Qr);
        if (NTIP > 0) {
            // check_contact_fingertips / check_collision (icub_env_with_hands.py:246-318): per fingertip the mean normal force
            // (applied impulse / dt, getContactPoints()[9]) of its contact points with the object, the number of tips in
            // contact and the number of robot-object contact points; kept in the Q record behind the object pose
            const I tip = L::loadI(T.tip_of);
            F tf[NTIP > 0 ? NTIP : 1], tc[NTIP > 0 ? NTIP : 1];
            PBRE_UNROLL for (int t = 0; t < NTIP; t++) { tf[t] = zero; tc[t] = zero; }
            F nro = zero;
            PBRE_UNROLL for (int c = NC_OT; c < NC_OT + NC_RO; c++) {
                if (!obj_on || !L::any(R.act[c])) continue;
                const I ts = L::gatherI(tip, owner_ro[c - NC_OT]);
                const F f = R.an[c] * inv_dt;
                nro = nro + L::sel(R.act[c], one, zero);
                PBRE_UNROLL for (int t = 0; t < NTIP; t++) {
                    const B is = L::band(R.act[c], L::eqi(ts, t));
                    tf[t] = tf[t] + L::sel(is, f, zero); tc[t] = tc[t] + L::sel(is, one, zero);
                }
            }
            F ntip = zero;
            PBRE_UNROLL for (int t = 0; t < NTIP; t++) {
                const B hit = L::gt(tc[t], zero);
                ntip = ntip + L::sel(hit, one, zero);
                Qn = L::sel(L::eqi(lane, TIP0 + t), L::sel(hit, tf[t] / L::max(tc[t], one), zero), Qn);
            }
            Qn = L::sel(L::eqi(lane, TIP0 + NTIP), ntip, L::sel(L::eqi(lane, TIP0 + NTIP + 1), nro, Qn));
        }
        if (obj_on) {
            V3 wn = v3(L::bcast(vnew, LC + 3), L::bcast(vnew, LC + 4), L::bcast(vnew, LC + 5));
            F ang = norm(wn);
            F cap = L::c(0.78539816339744831f) * inv_dt;
            ang = L::sel(L::gt(ang * dt, L::c(0.78539816339744831f)), cap, ang);
            F small = L::c(0.5f) * dt - dt * dt * dt * L::c(0.020833333333f) * ang * ang;
            F sh_, ch_;
            L::sincos(L::c(0.5f) * ang * dt, sh_, ch_);
            F sc_ = L::sel(L::lt(ang, L::c(0.001f)), small, sh_ / L::max(ang, L::c(1e-30f)));
            Q4 dq; dq.x = wn.x * sc_; dq.y = wn.y * sc_; dq.z = wn.z * sc_; dq.w = ch_;
            Q4 nq = qmul(dq, oq);
            F in = one / L::sqrt(nq.x*nq.x + nq.y*nq.y + nq.z*nq.z + nq.w*nq.w);
            F qc = L::sel(L::eqi(lane, LC + 3), nq.x, L::sel(L::eqi(lane, LC + 4), nq.y, L::sel(L::eqi(lane, LC + 5), nq.z, nq.w)));
            Qn = L::sel(L::band(L::gei(lane, LC + 3), L::lti(lane, LC + 7)), qc * in, Qn);
        }
        {   // NaN / Inf guard: a non-finite entry of the incoming state (x * 0 is NaN for x = NaN, +-Inf) -- which the solver's
            // clamps would otherwise turn into finite garbage -- is added to every position of the new state: 0 normally, NaN then.
            // observe() / Fast::finish find it there, count the env-step and (PBRE_F_AUTO_RESET) restart the env.
            const F finv = L::sum(L::fma(Qr, zero, Vr * zero));
            Qn = L::sel(posl, Qn + finv, Qn);
        }
        {   // action_repeat > 1: a group that left the apply_action loop earlier in this env.step() (X[14]) does not simulate
            const B skip = L::ne(L::bcast(Xr, 14), zero);
            Qn = L::sel(skip, Qr, Qn); Vn = L::sel(skip, Vr, Vn);
        }
        { float* so = st_out ? st_out : st; L::store(so, Qn); L::store(so + W, Vn); }
        PBRE_PROBE(10);     // integration, store

        if (mode & (M_OBS | M_TASK)) {
            // the observation re-reads the tables through a pointer the optimiser cannot identify with T: otherwise the table
            // values both phases use (joint frames, ancestor tables, ...) stay in registers across the whole solver loop
            const Tables* Tp = &T;
            PBRE_OPAQUE(Tp);
            observe(*Tp, P, st_out ? st_out : st, Qn, Vn, Xr, out, mode, flags, env_id);
        }
    }

    // Geometric part of the observation of a state (Q, V, X = its three lane records).
    struct Obs { V3 ee, eul, vn, op, oe, rel, er, tg; F q; };
    static PBRE_HD Obs geom(const Tables& T, const Params& P, F Qn, F Vn, F Xr) {
        const I lane = L::lane();
        const F zero = L::c(0.f);
        const B robot = L::lti(lane, NJ);
        Obs o;
        o.q = L::sel(robot, Qn, zero);
        F qd = L::sel(robot, Vn, zero);
        Kin K; fk(T, o.q, K);
        Sp Sq; Sq.a = scl(K.S.a, qd); Sq.l = scl(K.S.l, qd);
        Sp Vs = chain_sum(T, Sq);
        const int eo = T.ee_owner;
        M3 Re; PBRE_UNROLL for (int k = 0; k < 9; k++) Re.m[k] = L::bcast(K.R.m[k], eo);
        V3 pe = bcastv(K.p, eo);
        M3 Eo; PBRE_UNROLL for (int k = 0; k < 9; k++) Eo.m[k] = L::c(T.ee_R[k]);
        M3 Ree = mm(Re, Eo);
        o.ee = add(pe, mv(Re, v3(L::c(T.ee_p[0]), L::c(T.ee_p[1]), L::c(T.ee_p[2]))));
        V3 vee = add(bcastv(Vs.l, eo), cross(bcastv(Vs.a, eo), o.ee));
        o.eul = quat_euler(R_quat(Ree));
        o.op = v3(L::bcast(Qn, LC), L::bcast(Qn, LC + 1), L::bcast(Qn, LC + 2));
        Q4 oq; oq.x = L::bcast(Qn, LC + 3); oq.y = L::bcast(Qn, LC + 4); oq.z = L::bcast(Qn, LC + 5); oq.w = L::bcast(Qn, LC + 6);
        o.oe = quat_euler(oq);
        // object pose in the hand frame via the Euler round trip the reference performs (panda_push_gym_env.py:168-174)
        Q4 qh = euler_quat(o.eul), qo = euler_quat(o.oe);
        o.rel = mtv(quat_R(qh), sub(o.op, o.ee));
        Q4 qhi; qhi.x = zero - qh.x; qhi.y = zero - qh.y; qhi.z = zero - qh.z; qhi.w = qh.w;
        o.er = quat_euler(qmul(qhi, qo));
        o.tg = v3(L::bcast(Xr, 0), L::bcast(Xr, 1), L::bcast(Xr, 2));
        // Panda: normalised EE velocity (panda_env.py:174-178); iCub: raw (icub_env.py:233-236)
        o.vn = P.robot >= 1 ? vee : v3(vee.x / L::c(0.04f), (vee.y - L::c(0.01f)) / L::c(0.07f), vee.z / L::c(0.03f));
        return o;
    }
    static PBRE_HD V3 selv3(B c, const V3& a, const V3& b) { return selv(c, a, b); }

    // Observation / reward / termination of the current state (Qn, Vn, Xr = the three state records).  With
    // PBRE_F_AUTO_RESET (flags & 2) a finished env is re-initialised right here (snapshot reset): the
    // transition's reward and done flag are returned together with the first observation of the next episode.
    static PBRE_HD void observe(const Tables& T, const Params& P, float* st, F Qn, F Vn, F Xr, float* out, int mode,
                                int flags = 0, unsigned long long env_id = 0) {
        const I lane = L::lane();
        const F zero = L::c(0.f), one = L::c(1.f);
        Obs o = geom(T, P, Qn, Vn, Xr);
        const V3 ee = o.ee, op = o.op, tg = o.tg;

        F reward = zero, done = zero;
        // NaN / Inf guard: a non-finite position of the new state (see step()): counted, returned as reward 0 / done 1, restarted with
        // PBRE_F_AUTO_RESET
        const B posn = (flags & 1) ? L::lti(lane, NJ) : L::lti(lane, LC + 7);
        const B bad = L::ne(L::sum(L::sel(posn, Qn * zero, zero)), zero);
        if (L::any(bad)) { if (L::first(L::sel(bad, one, zero)) != 0.f && L::lane0() && P.bad_count) PBRE_COUNT_BAD(P.bad_count); }
        if (mode & M_INITD) {
            // iCubPushGymEnv.reset (icub_push_gym_env.py:124-127): distances the normalised reward divides by
            F d1 = norm(sub(ee, op)), d2 = norm(sub(op, tg));
            F Xn = L::sel(L::eqi(lane, 12), d1, L::sel(L::eqi(lane, 13), d2, Xr));
            L::storem(st + 2 * W, Xn, L::lti(lane, 16));
        }
        if (mode & M_TASK) {
            // _termination + counter (panda_push_gym_env.py:239-242, 301-316) and _compute_reward (:318-331)
            F d1 = norm(sub(ee, op)), d2 = norm(sub(op, tg));
            F dsucc = P.task >= 1 ? d2 : d1;
            B succ = L::le(dsucc, L::c(P.dist_min));
            F cnt = L::bcast(Xr, 3), term = L::bcast(Xr, 4);
            F mx = L::c((float)P.max_steps);
            B left;          // `if self._termination(): break` fired in this iteration of the apply_action loop
            if (P.task == 2) {
                // goal env (panda_push_gym_goal_env.py:89-122): _termination is only the step budget, success does
                // not latch; done = budget or success; sparse reward -(d > threshold)
                left = L::gt(cnt, mx);
                cnt = L::sel(L::gt(cnt, mx), cnt, cnt + one);
                done = L::sel(L::bor(succ, L::gt(cnt, mx)), one, zero);
                reward = L::sel(succ, zero, zero - one);
            } else {
                B d0 = L::bor(L::bor(succ, L::ne(term, zero)), L::gt(cnt, mx));
                left = d0;
                cnt = L::sel(d0, cnt, cnt + one);
                term = L::sel(succ, one, term);
                B dn = L::bor(L::bor(succ, L::ne(term, zero)), L::gt(cnt, mx));
                done = L::sel(dn, one, zero);
                F base = P.task == 1 ? zero - d1 - d2 : zero - d1;
                if (P.robot >= 1) {
                    // iCub (icub_reach_gym_env.py:318-330: the bonus is added; icub_push_gym_env.py:346-373: reward types 0 / 1)
                    if (P.task == 0) reward = base + L::sel(succ, L::c(1000.f) + (L::c(100.f) - d1 * L::c(80.f)), zero);
                    else {
                        if (P.reward_type != 0) {
                            F r1 = L::c(0.125f) * (one - d1 / L::bcast(Xr, 12));
                            F r2 = L::c(0.25f) * (one - d2 / L::bcast(Xr, 13));
                            base = r1 + L::sel(L::gt(d1, L::c(0.1f)), zero, r2);
                        }
                        reward = base + L::sel(succ, L::c(1000.f), zero);
                    }
                } else
                reward = L::sel(succ, L::c(1000.f) + (L::c(100.f) - dsucc * L::c(80.f)), base);
            }
            done = L::sel(bad, one, done); reward = L::sel(bad, zero, reward);
            const F lf = (mode & M_INNER) ? L::sel(left, one, zero) : zero;      // consumed by the remaining iterations, cleared by the last one
            F Xn = L::sel(L::eqi(lane, 3), cnt, L::sel(L::eqi(lane, 4), term, L::sel(L::eqi(lane, 14), lf, Xr)));
            L::storem(st + 2 * W, Xn, L::lti(lane, 16));

            if ((flags & 2) && !(mode & M_INNER) && L::any(L::ne(done, zero))) {
                // ---- snapshot reset of the finished groups: the settled state of reset_simulation is invariant under the
                // sampled object x, y, yaw (flat table, vertical drop), so the next episode starts from the settled robot pose
                // and object height recorded at the last full reset, with freshly sampled pose and target
                const B fin = L::ne(done, zero);
                if (L::first(done) != 0.f && L::lane0()) snapshot_reset(T, P, env_id, st);    // scalar code, one lane per group
                L::fence();
                F Q2 = L::load(st), V2 = L::load(st + W), X2 = L::loadm(st + 2 * W, L::lti(lane, 16));
                const Obs o2 = geom(T, P, Q2, V2, X2);
                if (P.robot >= 1 && P.task >= 1) {      // icub_push_gym_env.py:124-127
                    F e1 = norm(sub(o2.ee, o2.op)), e2 = norm(sub(o2.op, o2.tg));
                    F Xn2 = L::sel(L::eqi(lane, 12), e1, L::sel(L::eqi(lane, 13), e2, X2));
                    L::storem(st + 2 * W, Xn2, L::band(fin, L::lti(lane, 16)));
                }
                o.ee = selv3(fin, o2.ee, o.ee); o.eul = selv3(fin, o2.eul, o.eul); o.vn = selv3(fin, o2.vn, o.vn);
                o.op = selv3(fin, o2.op, o.op); o.oe = selv3(fin, o2.oe, o.oe); o.rel = selv3(fin, o2.rel, o.rel);
                o.er = selv3(fin, o2.er, o.er); o.tg = selv3(fin, o2.tg, o.tg); o.q = L::sel(fin, o2.q, o.q);
            }
        }
        if (out) {
            // row-major [obs | reward | done]; obs layout SURVEY Appendix C
            const int nd = T.n_obs_j;
            const int od0 = 9 + nd + 12 + (P.task >= 1 ? 3 : 0);
            const int od = od0 + (NTIP > 0 ? NTIP + 2 : 0);
            if (NTIP > 0) L::storem(out + od0 - TIP0, Qn, L::band(L::gei(lane, TIP0), L::lti(lane, TIP0 + NTIP + 2)));   // fingertip forces, counts
            F head = L::sel(L::eqi(lane, 0), o.ee.x, L::sel(L::eqi(lane, 1), o.ee.y, L::sel(L::eqi(lane, 2), o.ee.z,
                     L::sel(L::eqi(lane, 3), o.eul.x, L::sel(L::eqi(lane, 4), o.eul.y, L::sel(L::eqi(lane, 5), o.eul.z,
                     L::sel(L::eqi(lane, 6), o.vn.x, L::sel(L::eqi(lane, 7), o.vn.y, o.vn.z))))))));
            L::storem(out, head, L::lti(lane, 9));
            { I oi = L::loadI(T.obs_idx); L::storex(out + 9, oi, o.q, L::gei(oi, 0)); }
            float* o2p = out + 9 + nd;
            F tail = L::sel(L::eqi(lane, 0), o.op.x, L::sel(L::eqi(lane, 1), o.op.y, L::sel(L::eqi(lane, 2), o.op.z,
                     L::sel(L::eqi(lane, 3), o.oe.x, L::sel(L::eqi(lane, 4), o.oe.y, L::sel(L::eqi(lane, 5), o.oe.z,
                     L::sel(L::eqi(lane, 6), o.rel.x, L::sel(L::eqi(lane, 7), o.rel.y, L::sel(L::eqi(lane, 8), o.rel.z,
                     L::sel(L::eqi(lane, 9), o.er.x, L::sel(L::eqi(lane, 10), o.er.y, L::sel(L::eqi(lane, 11), o.er.z,
                     L::sel(L::eqi(lane, 12), o.tg.x, L::sel(L::eqi(lane, 13), o.tg.y, o.tg.z))))))))))))));
            L::storem(o2p, tail, L::lti(lane, P.task >= 1 ? 15 : 12));
            F rd = L::sel(L::eqi(lane, 0), reward, done);
            L::storem(out + od, rd, L::lti(lane, 2));
        }
    }

    // ---------------------------------------------------------------- Cartesian control (use_IK = 1)
    // apply_action, IK branch (icub_reach_gym_env.py:204-230 + icub_env.py:262-330; panda_push_gym_env.py:197-222 +
    // panda_env.py:229-291): accumulate the scaled action on the commanded hand pose (X[6..11]), clip rotation and
    // workspace, then damped-least-squares IK from the current joint angles (restated in oracle/pbre_oracle.c orc_ik:
    // dq = J^T (J J^T + lambda^2 I)^-1 e over the joints of the chain to the end effector, <= ik_iters iterations, stop
    // when the position error < ik_res).  Lane k owns Jacobian column k; the 6x6 normal matrix is 21 group all-reduces.
    // reset: targets of the home hand pose (robot.reset, icub_env.py:147-148).  Writes tgt[0..NJ) and X[6..11].
    static PBRE_HD void ik_targets(const Tables& T, const Params& P, float* st, const float* act, float* tgt, bool reset) {
        const I lane = L::lane();
        const F zero = L::c(0.f), one = L::c(1.f);
        const B robot = L::lti(lane, NJ);
        F Qr = L::load(st), Xr = L::loadm(st + 2 * W, L::lti(lane, 16));
        F q = L::sel(robot, Qr, zero);
        const F q0 = q;
        V3 pos, eul;
        if (reset) {
            pos = v3(L::c(P.home_hand[0]), L::c(P.home_hand[1]), L::c(P.home_hand[2]));
            eul = v3(L::c(P.home_hand[3]), L::c(P.home_hand[4]), L::c(P.home_hand[5]));
        } else {
            const F ps = L::c(P.ik_ps), rs = L::c(P.ik_rs);
            pos = v3(L::fma(L::loadu(act), ps, L::bcast(Xr, 6)), L::fma(L::loadu(act + 1), ps, L::bcast(Xr, 7)), L::fma(L::loadu(act + 2), ps, L::bcast(Xr, 8)));
            eul = v3(L::bcast(Xr, 9), L::bcast(Xr, 10), L::bcast(Xr, 11));
            if (P.ik_abs) {
                // robot-level apply_action (icub_env.py:262-300): the action is the hand pose itself
                pos = v3(L::loadu(act), L::loadu(act + 1), L::loadu(act + 2));
                if (P.ctrl_ori) {
                    eul.x = clampf(L::loadu(act + 3), L::c(P.eu_lim[0][0]), L::c(P.eu_lim[0][1]));
                    eul.y = clampf(L::loadu(act + 4), L::c(P.eu_lim[1][0]), L::c(P.eu_lim[1][1]));
                    eul.z = clampf(L::loadu(act + 5), L::c(P.eu_lim[2][0]), L::c(P.eu_lim[2][1]));
                }
            } else
            if (P.ctrl_ori) {
                eul.x = clampf(L::fma(L::loadu(act + 3), rs, eul.x), L::c(P.eu_lim[0][0]), L::c(P.eu_lim[0][1]));
                eul.y = clampf(L::fma(L::loadu(act + 4), rs, eul.y), L::c(P.eu_lim[1][0]), L::c(P.eu_lim[1][1]));
                eul.z = clampf(L::fma(L::loadu(act + 5), rs, eul.z), L::c(P.eu_lim[2][0]), L::c(P.eu_lim[2][1]));
            }
        }
        V3 cp = pos;                      // the pose handed to the IK; at reset the stored pose stays the unclipped home pose
        if (P.robot != 0 || (!reset && !P.ik_abs)) {     // pandaEnv.apply_action clips z only (panda_env.py:243-247); the task env / iCub clip x, y, z
            cp.x = clampf(cp.x, L::c(P.rws[0][0]), L::c(P.rws[0][1]));
            cp.y = clampf(cp.y, L::c(P.rws[1][0]), L::c(P.rws[1][1]));
        }
        cp.z = clampf(cp.z, L::c(P.rws[2][0]), L::c(P.rws[2][1]));
        if (!reset) pos = cp;
        {
            F Xn = L::sel(L::eqi(lane, 6), pos.x, L::sel(L::eqi(lane, 7), pos.y, L::sel(L::eqi(lane, 8), pos.z,
                   L::sel(L::eqi(lane, 9), eul.x, L::sel(L::eqi(lane, 10), eul.y, eul.z)))));
            // a group that already left the apply_action loop of this env.step() (action_repeat > 1, X[14]) keeps its hand pose
            const B live = L::eq(L::bcast(Xr, 14), zero);
            L::storem(st + 2 * W, Xn, L::band(live, L::band(L::gei(lane, 6), L::lti(lane, 12))));
        }
        const M3 Rt = quat_R(euler_quat(eul));
        const V3 tp = add(cp, mv(Rt, v3(L::c(P.ik_off[0]), L::c(P.ik_off[1]), L::c(P.ik_off[2]))));
        const B chain = L::band(robot, L::nei(L::loadI(T.on_chain), 0));
        const int eo = T.ee_owner;
        M3 Eo; PBRE_UNROLL for (int k = 0; k < 9; k++) Eo.m[k] = L::c(T.ee_R[k]);
        const F l2 = L::c(P.ik_l2);
        for (int it = 0; it < P.ik_iters; it++) {
            Kin K; fk(T, q, K);
            M3 Re; PBRE_UNROLL for (int k = 0; k < 9; k++) Re.m[k] = L::bcast(K.R.m[k], eo);
            const V3 pe = add(bcastv(K.p, eo), mv(Re, v3(L::c(T.ee_lp[0]), L::c(T.ee_lp[1]), L::c(T.ee_lp[2]))));
            F e[6];
            { V3 d = sub(tp, pe); e[0] = d.x; e[1] = d.y; e[2] = d.z; }
            const B go = L::ge(L::sqrt(L::fma(e[0], e[0], L::fma(e[1], e[1], e[2] * e[2]))), L::c(P.ik_res));
            if (!L::any(go)) break;
            {   // orientation error as a world-frame rotation vector: axis-angle of Rt (Re Eo)^T
                const M3 Ree = mm(Re, Eo);
                M3 Rr;
                PBRE_UNROLL for (int i = 0; i < 3; i++)
                    PBRE_UNROLL for (int j = 0; j < 3; j++)
                        Rr.m[i*3+j] = L::fma(Rt.m[i*3], Ree.m[j*3], L::fma(Rt.m[i*3+1], Ree.m[j*3+1], Rt.m[i*3+2] * Ree.m[j*3+2]));
                F sx = Rr.m[7] - Rr.m[5], sy = Rr.m[2] - Rr.m[6], sz = Rr.m[3] - Rr.m[1];
                F s2 = L::sqrt(L::fma(sx, sx, L::fma(sy, sy, sz * sz))) * L::c(0.5f);
                F c2 = (Rr.m[0] + Rr.m[4] + Rr.m[8] - one) * L::c(0.5f);
                F ang = L::atan2(s2, c2);
                F f = L::sel(L::gt(s2, L::c(1e-9f)), ang / (L::c(2.f) * L::max(s2, L::c(1e-30f))), L::c(0.5f));
                e[3] = f * sx; e[4] = f * sy; e[5] = f * sz;
            }
            F J[6];
            {
                V3 jl = add(K.S.l, cross(K.S.a, pe));
                J[0] = L::sel(chain, jl.x, zero); J[1] = L::sel(chain, jl.y, zero); J[2] = L::sel(chain, jl.z, zero);
                J[3] = L::sel(chain, K.S.a.x, zero); J[4] = L::sel(chain, K.S.a.y, zero); J[5] = L::sel(chain, K.S.a.z, zero);
            }
            // A = J J^T + lambda^2 I (symmetric), Cholesky A = G G^T, solve A y = e
            F G[6][6], y[6];
            PBRE_UNROLL for (int a = 0; a < 6; a++)
                PBRE_UNROLL for (int b = 0; b <= a; b++) {
                    F sum = L::sum(J[a] * J[b]);
                    if (a == b) sum = sum + l2;
                    PBRE_UNROLL for (int k = 0; k < b; k++) sum = sum - G[a][k] * G[b][k];
                    G[a][b] = a == b ? L::sqrt(sum) : sum / G[b][b];
                }
            PBRE_UNROLL for (int a = 0; a < 6; a++) { F sum = e[a]; PBRE_UNROLL for (int k = 0; k < a; k++) sum = sum - G[a][k] * y[k]; y[a] = sum / G[a][a]; }
            PBRE_UNROLL for (int a = 5; a >= 0; a--) { F sum = y[a]; PBRE_UNROLL for (int k = a + 1; k < 6; k++) sum = sum - G[k][a] * y[k]; y[a] = sum / G[a][a]; }
            F dq = zero;
            PBRE_UNROLL for (int a = 0; a < 6; a++) dq = L::fma(J[a], y[a], dq);
            const F qn = L::sel(L::band(go, chain), q + dq, q);
            const bool moved = L::any(L::ne(qn, q));
            q = qn;
            // no joint angle of any group of the wave changed (targets out of reach: the damped step is below half an ulp): every further
            // iteration would repeat this one, leaving gives the same targets bit for bit
            if (!moved) break;
        }
        // joints off the chain: the iCub sends them to their rest pose (icub_env.py:316-317), PyBullet returns the Panda's
        // current finger positions
        F qdes = L::sel(chain, q, L::sel(L::nei(L::loadI(T.blocked), 0), L::load(T.home), q0));
        if (SH::MREC) {
            // setJointMotorControlArray(all joints, positionGains 0.2) (icub_env.py:319-336, panda_env.py:275-282); with max_vel the
            // iCub commands every joint with positionGain 0.2 and maxVelocity (:338-346), the Panda its 7 arm joints with PyBullet's
            // default gain and maxVelocity (panda_env.py:284-290): P.cmd_nj / cmd_kp / cmd_vmax
            const B cmd = P.cmd_nj > 0 ? L::band(robot, L::lti(lane, P.cmd_nj)) : L::lti(lane, W);
            L::storem(tgt, qdes, L::band(cmd, robot));
            L::storem(tgt + W, P.cmd_kp > 0.f ? L::c(P.cmd_kp) : L::load(T.kp_hold), cmd);
            L::storem(tgt + 2 * W, one, cmd);
            L::storem(tgt + 3 * W, L::c(P.cmd_vmax), cmd);
        } else
            L::storem(tgt, qdes, robot);
    }

    // ---------------------------------------------------------------- reset (initial state before the settle steps)
    static PBRE_HD void philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned o[4]) {
        for (int r = 0; r < 10; r++) {
            unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
            unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
    }
    static PBRE_HD float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
    static PBRE_HD float clamps(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

    // robot.reset + WorldEnv._sample_pose (reference panda_env.py:51-79, world_env.py:145-176): scalar, one call per env
    static PBRE_HD void init_state(const Tables& T, const Params& P, unsigned long long env_id, unsigned episode, float* st) {
        float* ob = st + LC;                       // object position (3) + quaternion (4) inside the Q record
        float* X = st + 2 * W;
        // (the per-env object parameters X[12], X[13], X[15] of a Panda task env are not part of the episode: a reset keeps them)
        const float k12 = st[2 * W + 12], k13 = st[2 * W + 13], k15 = st[2 * W + 15], v15 = st[W + 15];      // (V[15]: the robot's per-env link damping)
        for (int k = 0; k < STATE; k++) st[k] = 0.f;
        if (W == 16) { st[2 * W + 12] = k12; st[2 * W + 13] = k13; st[2 * W + 15] = k15; st[W + 15] = v15; }
        for (int k = 0; k < T.ndof; k++) st[k] = T.home[k];
        const float x_min = P.ws[0][0] + 0.05f, x_max = P.ws[0][1] - 0.1f;
        const float y_min = P.ws[1][0] + 0.05f, y_max = P.ws[1][1] - 0.05f;
        float px = x_min + 0.5f * (x_max - x_min), py = y_min + 0.5f * (y_max - y_min);
        const float pz = P.h_table + 0.07f;
        // (the robot-level scenes -- shapes with a motor record: helloworld_icub.py:51, helloworld_panda.py:78 -- load their object with
        // p.loadURDF(path, position): identity orientation, not WorldEnv's yaw of pi/4)
        float yaw = SH::MREC ? 0.f : 0.78539816339744831f;
        if (P.obj_std > 0.f) {
            unsigned r[4];
            philox((unsigned)env_id, (unsigned)(env_id >> 32), episode, 0u, P.seed_lo, P.seed_hi, r);
            px += -P.obj_std + 2.f * P.obj_std * u01(r[0]);
            py += -P.obj_std + 2.f * P.obj_std * u01(r[1]);
            yaw = -0.78539816339744831f + 1.57079632679489662f * u01(r[2]);
        }
        ob[0] = clamps(px, x_min, x_max); ob[1] = clamps(py, y_min, y_max); ob[2] = pz;
        ob[3] = 0.f; ob[4] = 0.f; sincos_f(0.5f * yaw, ob[5], ob[6]);      // (pbre_math.hpp: the same values as Fast::finish's in-kernel restart)
        X[5] = (float)(int)episode;       // 0xFFFFFFFF marks a record that was never reset (episode -1)
        if (P.use_ik) for (int k = 0; k < 6; k++) X[6 + k] = P.home_hand[k];
    }
    // sample_tg_pose (reference panda_push_gym_env.py:333-360) on the settled object position
    static PBRE_HD void sample_target(const Params& P, unsigned long long env_id, unsigned episode, float* st) {
        if (P.task < 1) return;
        const float* ob = st + LC;
        float* X = st + 2 * W;
        const float tx_min = P.ws[0][0] + 0.07f, tx_max = P.ws[0][1] - 0.07f;
        float tx = ob[0] + 0.05f, ty = ob[1] + 0.05f;
        if (P.tg_std > 0.f) {
            unsigned r[4];
            philox((unsigned)env_id, (unsigned)(env_id >> 32), episode, 1u, P.seed_lo, P.seed_hi, r);
            const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u01(r[1]);
            const float rad = sqrtf(-2.f * logf(u1)) * P.tg_std;
            float su, cu;
            sincos_f(6.28318530717958648f * u2, su, cu);          // (as Fast::finish: every restart path samples identical values)
            tx = ob[0] + rad * cu;
            ty = ob[1] + rad * su;
        }
        X[0] = clamps(tx, tx_min, tx_max); X[1] = clamps(ty, P.ws[1][0], P.ws[1][1]); X[2] = ob[2];
        X[3] = 0.f; X[4] = 0.f;
    }
    // PBRE_F_AUTO_RESET: next episode of a finished env from the settled snapshot (scalar, one call per env)
    static PBRE_HD void snapshot_reset(const Tables& T, const Params& P, unsigned long long env_id, float* st) {
        const unsigned ep = (unsigned)(int)st[2 * W + 5] + 1u;
        init_state(T, P, env_id, ep, st);                         // zeroed record, sampled object x, y, yaw, episode, home hand pose
        for (int k = 0; k < T.ndof; k++) st[k] = T.rst_q[k];      // settled robot pose
        st[LC + 2] = P.rst_objz;                                  // settled object height
        sample_target(P, env_id, ep, st);
    }
};

}  // namespace pbre
PBRE_FP_CONTRACT_FAST       // (pbre_math.hpp: the setting the other sources of csrc/ are written for; the amdgcn target has no push / pop of it)
