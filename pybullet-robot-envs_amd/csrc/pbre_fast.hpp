// pbre_fast.hpp -- lane-per-env fast path of the step (one thread = one env, everything in VGPRs).
//
// step_t<false> handles the common case exactly: no robot collision sphere within the contact margin of the
// object or the table and no joint at/beyond a limit, i.e. the only constraint rows are the 9 joint
// motors and up to 4 object-table contacts (normal +z => Bullet's btPlaneSpace1 friction directions are
// the constants (0,-1,0) and (1,0,0), so the object rows are sparse).  Such an env needs ~850
// instructions per env-step instead of ~17000 in the 16-lane row kernel (pbre_core.hpp), because nothing
// is computed redundantly across lanes and no cross-lane reduction is needed.  step_t<true> adds the robot-contact
// and joint-limit rows for the remaining envs (dense rows, whole register file).  Which variant an env needs is
// decided by classify() on the state it is in.
//
// Same mathematics as pbre_core.hpp (world-frame RNEA + CRBA + explicit M^-1, Bullet row order:
// motors in alternating direction, normals, frictions), same reference call sites.  The kinematic
// topology is a compile-time template argument so every array index folds and all link data stay in
// registers; the numeric model constants are read from `Tables` with wave-uniform addresses (scalar
// loads).  Plain C++: compiled for the device in pbre_capi.hip and for the host in tests/host_emu.
#pragma once
#include <math.h>
#include "pbre_tables.hpp"
#include "pbre_math.hpp"

#ifndef PBRE_HD
#define PBRE_HD
#endif
#ifndef PBRE_UNROLL
#define PBRE_UNROLL
#endif
#ifndef PBRE_REG_BARRIER    // compiler-only memory barrier (device build); nothing on the host
#define PBRE_REG_BARRIER() do {} while (0)
#endif
#ifndef PBRE_LAUNDER        // hide a (uniform) pointer's provenance from the optimiser (device build)
#define PBRE_LAUNDER(p) do {} while (0)
#endif
#ifndef PBRE_CONST_AS       // device build: the constant address space -- wave-uniform loads from it are scalar loads (s_load), whatever stores
#define PBRE_CONST_AS       // precede them; nothing on the host.  See step_t's call of finish().
#endif
#ifndef PBRE_ANY            // wave-uniform "any lane" on the device; identity on the host (one env per call)
#define PBRE_ANY(x) (x)
#endif
#ifndef PBRE_PROBE           // phase timing of one wave (tools/phase_probe.py builds with -DPBRE_PHASE_PROBE); nothing otherwise
#define PBRE_PROBE(k)
#define PBRE_PROBE_DECL
#endif
#ifndef PBRE_NAN_GUARD      // 0: build without the NaN / Inf guard (A/B of its cost)
#define PBRE_NAN_GUARD 1
#endif
#ifndef PBRE_COUNT_BAD      // ++*p from any number of lanes (device: atomicAdd)
#define PBRE_COUNT_BAD(p) (++*(p))
#endif
// y0 += s * x0, y1 += s * x1: with -DPBRE_PK_SQUARE=1 ONE v_pk_fma_f32 on the device (two fmaf otherwise and on the host: the same IEEE
// operations, the same bits).  Round 6 measured that a packed fp32 FMA issues like a scalar one (profiles/r06_ubench_valu.txt; rounds 1-5
// believed 7.2 cycles), so the matrix squarings of the closed forms were rewritten to need half the instructions -- 1900 fewer per k_fast
// wave -- and the step did not get faster: 0.1868 against 0.1870 ms stationary at 131072 envs, 0.1249 / 0.1253 at 16384, rows bit-identical
// (profiles/r06w_pk_square_ab.txt).  The wave is bound by dependent-issue latency, not by its instruction count; the packed build also
// spills more (k_fused 100 -> 196 B of scratch per lane).  Off by default.
#ifndef PBRE_PK_SQUARE
#define PBRE_PK_SQUARE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__) && PBRE_PK_SQUARE
typedef float pbre_f2 __attribute__((ext_vector_type(2)));
#define PBRE_FMA2(s, x0, x1, y0, y1) do { const pbre_f2 r_ = __builtin_elementwise_fma((pbre_f2){(s), (s)}, (pbre_f2){(x0), (x1)}, (pbre_f2){(y0), (y1)}); \
        (y0) = r_.x; (y1) = r_.y; } while (0)
#else
#define PBRE_FMA2(s, x0, x1, y0, y1) do { (y0) = fmaf((s), (x0), (y0)); (y1) = fmaf((s), (x1), (y1)); } while (0)
#endif
#ifndef PBRE_OC_PROBE       // test builds: count the lanes that pass / fail the validity bound of the object block's closed form
#define PBRE_OC_PROBE(ok) do {} while (0)
#endif
#ifndef PBRE_RT_PROBE       // test builds: why a lane did not take the closed form of the residual exit (bit mask) and its state record
#define PBRE_RT_PROBE(why, st) do {} while (0)
#endif
#ifndef PBRE_ROW_SUM_I      // sum of an int over the 16 lanes of a row wave's group (device build: pbre_panda.hpp; sweep<3> is never instantiated on the host)
#define PBRE_ROW_SUM_I(x) (x)
#endif
#ifndef PBRE_PAIR_SYNC      // the robot wave of a pair waits for its object wave (device build: pbre_panda.hpp; never reached on the host)
#define PBRE_PAIR_SYNC(px, ln) do {} while (0)
#endif
#include "pbre_objstep.hpp"

// The lane-per-env code is written as plain a * b + c across statements and is meant to be fused wherever the compiler can (pbre_math.hpp:
// contraction is stated per header): 32.6 k instead of 38.1 k VALU instructions per k_fast wave (profiles/r04_fp_contract_bisect.txt).
// Device and emulation are compared within tolerances, variants of one build bit for bit.
PBRE_FP_CONTRACT_FAST
namespace pbre {

// LDS exchange area of one block of the pair kernel (k_fast_pair, pbre_capi.hip): 64 envs, the robot's half of the step on wave 0, the
// object's half on wave 1.  [value][lane] so that a wave's accesses are conflict-free.
struct PairX {
    float o[7][64];          // object wave -> robot wave: the object's new position (3) and quaternion (4)
    float sc[3 * W][64];     // robot wave, for itself: collision-sphere centres of the new state (tested against the new object pose
                             // once the object wave has delivered it)
    // the robot wave's own note of where its object wave is (written and read by the robot wave only).  g == nullptr: a sibling wave of the
    // block -- `o` is complete behind the block barrier; else a wave of ANOTHER block (k_fused's tail pairs, pbre_panda.hpp): the pair's
    // global record -- this PairX itself, then one word per lane that carries the launch's sequence number `seq` once `o` is complete
    const float* g;
    int seq;
};

// Franka Panda as flattened by build_tables(): 7-revolute chain, two prismatic fingers on lane 6,
// lane 6 carries 3 rigid sub-bodies (link7, hand, grasptarget).
struct TopoPanda {
    static constexpr int ND = 9;
    static constexpr int parent(int j) { return j == 0 ? -1 : (j <= 6 ? j - 1 : 6); }
    static constexpr int jtype(int j) { return j <= 6 ? 1 : 2; }
    static constexpr int nsub(int j) { return j == 6 ? 3 : 1; }
    static constexpr bool is_anc(int i, int j) {   // i ancestor-or-self of j
        return i == j || (j >= 0 && parent(j) >= 0 && is_anc(i, parent(j)));
    }
};

// fast path preconditions that are uniform over the batch (checked once on the host)
inline bool fast_scene_ok(const Params& P) {
    return P.jd_dt == 0.f;                                          // explicit joint damping
    // (a box with unequal principal inertias is stepped by ObjStep inside the simple-env kernel, and its complex envs by the row
    // kernel: P.obj_iso)
}

template <class Topo>
inline bool topo_matches(const Tables& T) {
    if (T.ndof != Topo::ND) return false;
    for (int j = 0; j < Topo::ND; j++) {
        if (T.anc[0][j] != Topo::parent(j) || T.jtype[j] != Topo::jtype(j)) return false;
        for (int b = 0; b < NSUB; b++) if ((T.sb_m[b][j] != 0.f || T.sb_I[b][0][j] != 0.f) != (b < Topo::nsub(j))) return false;
        if (T.jdamp[j] != 0.f) return false;   // the fast path folds joint damping out
    }
    return true;
}

typedef PBRE_CONST_AS Tables CTables;

template <class Topo>
struct Fast {
    static constexpr int ND = Topo::ND;
    struct V3 { float x, y, z; };
    static PBRE_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
    static PBRE_HD V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
    static PBRE_HD V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
    static PBRE_HD V3 scl(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
    static PBRE_HD float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
    static PBRE_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
    static PBRE_HD float norm(V3 a) { return sqrtf(dot(a, a)); }
    struct M3 { float m[9]; };
    static PBRE_HD V3 mv(const M3& A, V3 v) {
        return v3(fmaf(A.m[0], v.x, fmaf(A.m[1], v.y, A.m[2] * v.z)), fmaf(A.m[3], v.x, fmaf(A.m[4], v.y, A.m[5] * v.z)),
                  fmaf(A.m[6], v.x, fmaf(A.m[7], v.y, A.m[8] * v.z)));
    }
    static PBRE_HD V3 mtv(const M3& A, V3 v) {
        return v3(fmaf(A.m[0], v.x, fmaf(A.m[3], v.y, A.m[6] * v.z)), fmaf(A.m[1], v.x, fmaf(A.m[4], v.y, A.m[7] * v.z)),
                  fmaf(A.m[2], v.x, fmaf(A.m[5], v.y, A.m[8] * v.z)));
    }
    static PBRE_HD M3 mm(const M3& A, const M3& B) {
        M3 C;
        PBRE_UNROLL for (int i = 0; i < 3; i++)
            PBRE_UNROLL for (int j = 0; j < 3; j++)
                C.m[i*3+j] = fmaf(A.m[i*3], B.m[j], fmaf(A.m[i*3+1], B.m[3+j], A.m[i*3+2] * B.m[6+j]));
        return C;
    }
    static PBRE_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
#if defined(__HIP_DEVICE_COMPILE__)
    static PBRE_HD float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }   // one v_med3_f32 (lo <= hi, no NaNs)
    // Joint-space velocity vector as (ND+1)/2 register pairs (the last pair padded with a zero): w += d * column is 5 v_pk_fma_f32 for
    // the 9-DoF Panda.  A v_pk_fma_f32 occupies the SIMD for two passes on this chip (profiles/r01_ubench_pkfma.txt), so this is
    // throughput-neutral against 9 scalar FMAs at 2 waves/SIMD, but it halves the dependent chain a lone wave waits for and the
    // pair layout allocates with far fewer spills (44 vs 228-248 B/lane for the 4 pairs + 1 scalar and the scalar layouts).
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct WV { f2 p[(ND + 1) / 2]; };
    static PBRE_HD float wget(const WV& w, int k) { return w.p[k >> 1][k & 1]; }
    static PBRE_HD void wset(WV& w, int k, float v) { w.p[k >> 1][k & 1] = v; }
    static PBRE_HD void waxpy(WV& w, float d, const WV& col) {
        const f2 dd = {d, d};
        PBRE_UNROLL for (int i = 0; i < (ND + 1) / 2; i++) w.p[i] = __builtin_elementwise_fma(dd, col.p[i], w.p[i]);
    }
#else
    static PBRE_HD float med3(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
    struct WV { float v[2 * ((ND + 1) / 2)]; };
    static PBRE_HD float wget(const WV& w, int k) { return w.v[k]; }
    static PBRE_HD void wset(WV& w, int k, float v) { w.v[k] = v; }
    static PBRE_HD void waxpy(WV& w, float d, const WV& col) { for (int k = 0; k < 2 * ((ND + 1) / 2); k++) w.v[k] = fmaf(d, col.v[k], w.v[k]); }
#endif
    static constexpr int sym(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

    static PBRE_HD void sincos_(float x, float& sn, float& cs) { sincos_f(x, sn, cs); }      // pbre_math.hpp

    struct Q4 { float x, y, z, w; };
    static PBRE_HD M3 quat_R(Q4 q) {
        M3 R; float x = q.x, y = q.y, z = q.z, w = q.w;
        R.m[0] = 1.f - 2.f * (y*y + z*z); R.m[1] = 2.f * (x*y - w*z);       R.m[2] = 2.f * (x*z + w*y);
        R.m[3] = 2.f * (x*y + w*z);       R.m[4] = 1.f - 2.f * (x*x + z*z); R.m[5] = 2.f * (y*z - w*x);
        R.m[6] = 2.f * (x*z - w*y);       R.m[7] = 2.f * (y*z + w*x);       R.m[8] = 1.f - 2.f * (x*x + y*y);
        return R;
    }
    static PBRE_HD Q4 qmul(Q4 a, Q4 b) {
        Q4 o;
        o.x = a.w*b.x + a.x*b.w + a.y*b.z - a.z*b.y; o.y = a.w*b.y - a.x*b.z + a.y*b.w + a.z*b.x;
        o.z = a.w*b.z + a.x*b.y - a.y*b.x + a.z*b.w; o.w = a.w*b.w - a.x*b.x - a.y*b.y - a.z*b.z;
        return o;
    }
    static PBRE_HD Q4 R_quat(const M3& R) {   // btMatrix3x3::getRotation
        float tr = R.m[0] + R.m[4] + R.m[8];
        Q4 q;
        if (tr > 0.f) {
            float s = sqrtf(tr + 1.f); q.w = s * .5f; s = .5f / s;
            q.x = (R.m[7] - R.m[5]) * s; q.y = (R.m[2] - R.m[6]) * s; q.z = (R.m[3] - R.m[1]) * s;
        } else {
            // branch-free over the three cases (kept in registers)
            Q4 c[3];
            PBRE_UNROLL for (int i = 0; i < 3; i++) {
                const int j = (i + 1) % 3, k = (i + 2) % 3;
                float s = sqrtf(fmaxf(R.m[i*4] - R.m[j*4] - R.m[k*4] + 1.f, 1e-30f)), kk = .5f / s;
                float t[3]; t[i] = s * .5f; t[j] = (R.m[j*3+i] + R.m[i*3+j]) * kk; t[k] = (R.m[k*3+i] + R.m[i*3+k]) * kk;
                c[i].x = t[0]; c[i].y = t[1]; c[i].z = t[2]; c[i].w = (R.m[k*3+j] - R.m[j*3+k]) * kk;
            }
            const bool c01 = R.m[0] < R.m[4], c12 = R.m[4] < R.m[8], c02 = R.m[0] < R.m[8];
            const bool use2 = (c01 && c12) || (!c01 && c02), use1 = c01 && !c12;
            // (component-wise selects: `use2 ? c[2] : ...` on the structs became an indexed load from an LDS copy of c[], 3 KB per wave)
            q.x = use2 ? c[2].x : (use1 ? c[1].x : c[0].x); q.y = use2 ? c[2].y : (use1 ? c[1].y : c[0].y);
            q.z = use2 ? c[2].z : (use1 ? c[1].z : c[0].z); q.w = use2 ? c[2].w : (use1 ? c[1].w : c[0].w);
        }
        return q;
    }
    static PBRE_HD Q4 euler_quat(V3 e) {
        float cr, sr, cp, sp, cy, sy;
        sincos_(e.x * .5f, sr, cr); sincos_(e.y * .5f, sp, cp); sincos_(e.z * .5f, sy, cy);
        Q4 q; q.x = sr*cp*cy - cr*sp*sy; q.y = cr*sp*cy + sr*cp*sy; q.z = cr*cp*sy - sr*sp*cy; q.w = cr*cp*cy + sr*sp*sy;
        return q;
    }
    static PBRE_HD V3 quat_euler(Q4 q) {
        float x = q.x, y = q.y, z = q.z, w = q.w;
        float sarg = -2.f * (x*z - w*y);
        V3 e;
        if (sarg <= -0.99999f) { e.x = 0.f; e.y = -1.57079632679489662f; e.z = 2.f * atan2f(x, -y); }
        else if (sarg >= 0.99999f) { e.x = 0.f; e.y = 1.57079632679489662f; e.z = 2.f * atan2f(-x, y); }
        else {
            e.x = atan2f(2.f * (y*z + w*x), w*w - x*x - y*y + z*z);
            e.y = asinf(fmaxf(fminf(sarg, 1.f), -1.f));
            e.z = atan2f(2.f * (x*y + w*z), w*w + x*x - y*y - z*z);
        }
        return e;
    }

    struct Kin { M3 R[ND]; V3 p[ND]; V3 Sa[ND], Sl[ND]; };   // link frames, joint axes (world, about world origin)

    template <class TT>
    static PBRE_HD void fk(const TT& T, const float* q, Kin& K) {
        PBRE_UNROLL for (int j = 0; j < ND; j++) {
            V3 ax = v3(T.axis[0][j], T.axis[1][j], T.axis[2][j]);
            M3 R0; PBRE_UNROLL for (int k = 0; k < 9; k++) R0.m[k] = T.R0[k][j];
            V3 p0 = v3(T.p0[0][j], T.p0[1][j], T.p0[2][j]);
            M3 Rl; V3 pl;
            if (Topo::jtype(j) == 1) {
                float c, s; sincos_(q[j], s, c); const float C = 1.f - c;
                M3 Rj;
                Rj.m[0] = c + ax.x*ax.x*C;      Rj.m[1] = ax.x*ax.y*C - ax.z*s; Rj.m[2] = ax.x*ax.z*C + ax.y*s;
                Rj.m[3] = ax.y*ax.x*C + ax.z*s; Rj.m[4] = c + ax.y*ax.y*C;      Rj.m[5] = ax.y*ax.z*C - ax.x*s;
                Rj.m[6] = ax.z*ax.x*C - ax.y*s; Rj.m[7] = ax.z*ax.y*C + ax.x*s; Rj.m[8] = c + ax.z*ax.z*C;
                Rl = mm(R0, Rj); pl = p0;
            } else {
                Rl = R0; V3 d = mv(R0, ax); pl = v3(fmaf(d.x, q[j], p0.x), fmaf(d.y, q[j], p0.y), fmaf(d.z, q[j], p0.z));
            }
            if (Topo::parent(j) < 0) { K.R[j] = Rl; K.p[j] = pl; }
            else { const int pj = Topo::parent(j) < 0 ? 0 : Topo::parent(j); K.R[j] = mm(K.R[pj], Rl); K.p[j] = add(K.p[pj], mv(K.R[pj], pl)); }
            V3 aw = mv(K.R[j], ax);
            if (Topo::jtype(j) == 1) { K.Sa[j] = aw; K.Sl[j] = cross(K.p[j], aw); }
            else { K.Sa[j] = v3(0.f, 0.f, 0.f); K.Sl[j] = aw; }
        }
    }

    // minimum signed distance of any robot collision sphere to the box (c, R, h)
    static PBRE_HD float sphere_box_dist(V3 sc, float sr, V3 bc, const M3& Rb, V3 h) {
        V3 dl = mtv(Rb, sub(sc, bc));
        V3 cl = v3(clampf(dl.x, -h.x, h.x), clampf(dl.y, -h.y, h.y), clampf(dl.z, -h.z, h.z));
        V3 df = sub(dl, cl);
        float len = norm(df);
        float best = fminf(fminf(h.x - fabsf(dl.x), h.y - fabsf(dl.y)), h.z - fabsf(dl.z));
        return len < 1e-9f ? -best - sr : len - sr;
    }

    // ... to the object, whatever its primitive (Params::obj_shape; round objects: Shapes::sphere_round)
    static PBRE_HD float sphere_obj_dist(const Params& P, V3 sc, float sr, V3 bc, const M3& Rb, V3 h) {
        if (P.obj_shape == 0) return sphere_box_dist(sc, sr, bc, Rb, h);
        const float s_[3] = {sc.x, sc.y, sc.z}, c_[3] = {bc.x, bc.y, bc.z}, h_[3] = {h.x, h.y, h.z};
        float n_[3], pb_[3];
        return Shapes::sphere_round(P.obj_shape, s_, sr, c_, Rb.m, h_, n_, pb_);
    }

    enum { M_ACTION = 1, M_OBS = 2, M_TASK = 4, M_TGT = 8,     // M_TGT: motor targets come from the IK target buffer
           M_INNER = 32 };   // a non-final iteration of the apply_action loop (action_repeat > 1): termination test + counter, no outputs

    // full sphere-vs-box test: signed distance, world normal box->sphere, point on the box
    static PBRE_HD float sphere_box(V3 sc, float sr, V3 bc, const M3& Rb, V3 h, V3& n, V3& pb) {
        V3 dl = mtv(Rb, sub(sc, bc));
        V3 cl = v3(clampf(dl.x, -h.x, h.x), clampf(dl.y, -h.y, h.y), clampf(dl.z, -h.z, h.z));
        V3 df = sub(dl, cl);
        const float len = norm(df);
        const bool inside = len < 1e-9f;
        const float il = 1.f / fmaxf(len, 1e-30f);
        const float ex = h.x - fabsf(dl.x), ey = h.y - fabsf(dl.y), ez = h.z - fabsf(dl.z);
        const bool ax_y = ey < ex; float best = ax_y ? ey : ex;
        const bool ax_z = ez < best; best = ax_z ? ez : best;
        const bool is_x = !ax_y && !ax_z, is_y = ax_y && !ax_z;
        const float sx = dl.x >= 0.f ? 1.f : -1.f, sy = dl.y >= 0.f ? 1.f : -1.f, sz = dl.z >= 0.f ? 1.f : -1.f;
        V3 nl = inside ? v3(is_x ? sx : 0.f, is_y ? sy : 0.f, ax_z ? sz : 0.f) : scl(df, il);
        V3 c2 = inside ? v3(is_x ? sx * h.x : cl.x, is_y ? sy * h.y : cl.y, ax_z ? sz * h.z : cl.z) : cl;
        n = mv(Rb, nl); pb = add(bc, mv(Rb, c2));
        return inside ? -best - sr : len - sr;
    }

    // full test against the object, whatever its primitive: signed distance, world normal object -> sphere, point on the object
    static PBRE_HD float sphere_obj(const Params& P, V3 sc, float sr, V3 bc, const M3& Rb, V3 h, V3& n, V3& pb) {
        if (P.obj_shape == 0) return sphere_box(sc, sr, bc, Rb, h, n, pb);
        const float s_[3] = {sc.x, sc.y, sc.z}, c_[3] = {bc.x, bc.y, bc.z}, h_[3] = {h.x, h.y, h.z};
        float n_[3], pb_[3];
        const float d = Shapes::sphere_round(P.obj_shape, s_, sr, c_, Rb.m, h_, n_, pb_);
        n = v3(n_[0], n_[1], n_[2]); pb = v3(pb_[0], pb_[1], pb_[2]);
        return d;
    }
    struct Cand { float dist; int idx; V3 n, pA, pB; float mu; int owner; };
    static PBRE_HD bool better(const Cand& x, const Cand& y) { return x.dist < y.dist || (x.dist == y.dist && x.idx < y.idx); }
    // keep the N best (smallest distance, ties -> lowest sphere index) candidates below the margin, best first: an insertion into
    // a sorted register array with compile-time indices (selects, no dynamic indexing)
    // (two candidates, best and second best: the iCub's lane-per-env code, pbre_lane.hpp)
    static PBRE_HD void keep2(const Cand& c, float margin, Cand& a, Cand& b) {
        if (c.dist < margin) {
            if (better(c, a)) { b = a; a = c; } else if (better(c, b)) b = c;
        }
    }
    template <int N>
    static PBRE_HD void keepn(const Cand& c, float margin, Cand (&k)[N]) {
        if (c.dist < margin) {
            Cand cur = c;
            PBRE_UNROLL for (int i = 0; i < N; i++) {
                if (better(cur, k[i])) { const Cand t = k[i]; k[i] = cur; cur = t; }
            }
        }
    }

    // Env classes.  Every stepping kernel finishes by classifying the NEW state (it has the kinematics at hand), so the
    // next step can launch the right kernel for every env without a pre-pass:
    //   0  simple : no robot collision sphere within the contact margin of object/table, no joint at a limit
    //               -> step_t<false> (fast path: motors + object-table contacts only)
    //   1  complex: robot contacts and/or active joint-limit rows -> step_t<true> (adds dense 9-DoF contact rows and the
    //               limit rows; needs the whole register file, launched only over the list of complex envs)
    // Both variants return the class of the state they produced.
    // Closed form of `ds` double sweeps (rows ND-1..0, then 0..ND-1) of the clamp-free motor rows on the error e = w - t (see step_t).
    // A: M^-1 (triangle), dinv[j] = 1 / A_jj.  A row j is e <- e - (e_j / A_jj) A_j, which zeroes e_j: the second visit of row 0 in a
    // double sweep is the identity, and after the first double sweep e_{ND-1} = 0 for good, so the remaining ds - 1 double sweeps act on
    // the first NH = ND - 1 components through one NH x NH matrix H (built column by column from the same row operations), and
    // H^(ds-1) e is evaluated by binary powering: ~4.2 k FMAs for 150 sweeps of 9 rows instead of ~15 k dependent ones.
    static constexpr int NH = ND - 1;
    static PBRE_HD void motor_closed(const float* A, const float* dinv, float* e, int ds) {
        auto rowop = [&](float* v, int j) {
            const float s = v[j] * dinv[j];
            PBRE_UNROLL for (int k = 0; k < ND; k++) if (k != j) v[k] = fmaf(-s, A[sym(k, j)], v[k]);
            v[j] = 0.f;
        };
        PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) rowop(e, j);
        PBRE_UNROLL for (int j = 1; j < ND; j++) rowop(e, j);
        int r = ds - 1;
        if (r <= 0) return;
        float H[NH][NH], G[NH][NH];
        PBRE_UNROLL for (int c = 0; c < NH; c++) {
            float v[ND];
            PBRE_UNROLL for (int k = 0; k < ND; k++) v[k] = k == c ? 1.f : 0.f;
            PBRE_UNROLL for (int j = NH - 1; j >= 0; j--) if (j <= c) rowop(v, j);     // (rows j > c find v_j = 0: identity)
            PBRE_UNROLL for (int j = 1; j < ND; j++) rowop(v, j);
            PBRE_UNROLL for (int k = 0; k < NH; k++) H[k][c] = v[k];
        }
        auto apply = [&](const float (*X)[NH]) {
            float y[NH];
            PBRE_UNROLL for (int i = 0; i < NH; i++) {
                float a = 0.f;
                PBRE_UNROLL for (int k = 0; k < NH; k++) a = fmaf(X[i][k], e[k], a);
                y[i] = a;
            }
            PBRE_UNROLL for (int i = 0; i < NH; i++) e[i] = y[i];
        };
        auto square = [&](const float (*X)[NH], float (*Y)[NH]) {
            static_assert(NH % 2 == 0, "columns in pairs");
            PBRE_UNROLL for (int i = 0; i < NH; i++)
                PBRE_UNROLL for (int c = 0; c < NH; c += 2) {
                    float a0 = 0.f, a1 = 0.f;
                    PBRE_UNROLL for (int k = 0; k < NH; k++) PBRE_FMA2(X[i][k], X[k][c], X[k][c + 1], a0, a1);
                    Y[i][c] = a0; Y[i][c + 1] = a1;
                }
        };
        for (;;) {
            if (r & 1) apply(H);
            r >>= 1; if (!r) break;
            square(H, G);
            if (r & 1) apply(G);
            r >>= 1; if (!r) break;
            square(G, H);
        }
    }
    // Bullet's residual exit over the clamp-free motor rows WITHOUT running every sweep (round 6; step_t<false, 0, true>).  In: e = w0 - t.
    // A motor row's velocity-level change |delta impulse / jacDiagABInv| is |e_j| at its turn, so the exit test of sweep `it` is
    // max_j |e_j at row j's turn| <= tau (the object's rows pass or fail by `opass`: bit it for it < OK, pass from OK on).  The sweeps
    // alternate direction (rows ND-1..0, then 0..ND-1); after the first pair e_{ND-1} = 0 at every pair boundary and a pair is the
    // NH x NH map H of motor_closed.  Scan: the first pair explicitly; then COARSE steps of eight pairs with H^8 (three squarings), each
    // followed by a trial pair from the state it reaches -- a lane advances while that trial pair does not pass; then explicit pairs
    // from where the lane stopped until one passes (at most eight + the tail).  At most 1 + 9 + 9 explicit pairs instead of 75.
    // The scan finds the first passing sweep among those it visits: every sweep from the lane's last coarse stop on, and the trial
    // pairs before it.  It would miss a sweep that passes INSIDE an earlier coarse block while the block's end does not (a residual
    // that is not decreasing in the sweep index); `mono` is cleared when the trial pairs' residuals do not decrease, and such a lane
    // takes the explicit rows.  Out: e = the error after the exit sweep (or after sweep iters - 1); returns the sweeps used.
    static PBRE_HD int motor_scan(const float* A, const float* dinv, float* e, int iters, float tau, unsigned opass, int OK, bool& mono) {
        auto row = [&](float* v, int j, float& r) {
            r = fmaxf(r, fabsf(v[j]));
            const float s = v[j] * dinv[j];
            PBRE_UNROLL for (int k = 0; k < ND; k++) if (k != j) v[k] = fmaf(-s, A[sym(k, j)], v[k]);
            v[j] = 0.f;
        };
        int used = iters;
        bool found = false;
        float ex[ND];
        PBRE_UNROLL for (int k = 0; k < ND; k++) ex[k] = 0.f;
        auto passes = [&](int it, float r) -> bool { const bool po = it >= OK || ((opass >> (it & 31)) & 1u) != 0u; return r <= tau && po && it < iters; };
        auto commit = [&](int it, float r, const float* v, bool live = true) {
            const bool hit = live && !found && passes(it, r);
            PBRE_UNROLL for (int k = 0; k < ND; k++) ex[k] = hit ? v[k] : ex[k];
            used = hit ? it + 1 : used;
            found = found || hit;
        };
        {   // sweeps 0 and 1
            float r = 0.f;
            PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) row(e, j, r);
            commit(0, r, e);
            r = 0.f;
            PBRE_UNROLL for (int j = 1; j < ND; j++) row(e, j, r);
            commit(1, r, e);
        }
        mono = true;
        const int mlast = (iters >> 1) - 1;           // index of the last pair
        int m = 1;                                      // the pair the lane's state `e` is the start of
        if (mlast >= 1) {
            float H[NH][NH], G[NH][NH];
            PBRE_UNROLL for (int c = 0; c < NH; c++) {
                float v[ND], rr = 0.f;
                PBRE_UNROLL for (int k = 0; k < ND; k++) v[k] = k == c ? 1.f : 0.f;
                PBRE_UNROLL for (int j = NH - 1; j >= 0; j--) if (j <= c) row(v, j, rr);
                PBRE_UNROLL for (int j = 1; j < ND; j++) row(v, j, rr);
                PBRE_UNROLL for (int k = 0; k < NH; k++) H[k][c] = v[k];
            }
            auto square = [&](const float (*X)[NH], float (*Y)[NH]) {
                PBRE_UNROLL for (int i = 0; i < NH; i++)
                    PBRE_UNROLL for (int c = 0; c < NH; c += 2) {
                        float a0 = 0.f, a1 = 0.f;
                        PBRE_UNROLL for (int k = 0; k < NH; k++) PBRE_FMA2(X[i][k], X[k][c], X[k][c + 1], a0, a1);
                        Y[i][c] = a0; Y[i][c + 1] = a1;
                    }
            };
            square(H, G); square(G, H); square(H, G);           // G = H^8
            bool stop = false;
            float r_last = 3e38f;
            for (int c = 0; c < 9; c++) {
                float v[ND], rR = 0.f, rF = 0.f;
                PBRE_UNROLL for (int i = 0; i < NH; i++) {
                    float a = 0.f;
                    PBRE_UNROLL for (int k = 0; k < NH; k++) a = fmaf(G[i][k], e[k], a);
                    v[i] = a;
                }
                v[NH] = 0.f;
                float cand[NH];
                PBRE_UNROLL for (int i = 0; i < NH; i++) cand[i] = v[i];
                PBRE_UNROLL for (int j = NH - 1; j >= 0; j--) row(v, j, rR);
                PBRE_UNROLL for (int j = 1; j < ND; j++) row(v, j, rF);
                const int mt = m + 8;
                const bool exit_here = passes(2 * mt, rR) || passes(2 * mt + 1, rF);
                const bool adv = !found && !stop && !exit_here && mt <= mlast;
                mono = mono && !(adv && !(rF <= r_last));
                r_last = adv ? rF : r_last;
                PBRE_UNROLL for (int i = 0; i < NH; i++) e[i] = adv ? cand[i] : e[i];
                m = adv ? mt : m;
                stop = stop || !adv;
                if (!PBRE_ANY(!stop && !found)) break;
            }
            for (int f = 0; f < 10; f++) {
                const bool live = !found && m <= mlast;
                if (!PBRE_ANY(live)) break;
                float v[ND], r = 0.f;
                PBRE_UNROLL for (int i = 0; i < NH; i++) v[i] = e[i];
                v[NH] = 0.f;
                PBRE_UNROLL for (int j = NH - 1; j >= 0; j--) row(v, j, r);
                commit(2 * m, r, v, live);
                r = 0.f;
                PBRE_UNROLL for (int j = 1; j < ND; j++) row(v, j, r);
                commit(2 * m + 1, r, v, live);
                PBRE_UNROLL for (int i = 0; i < NH; i++) e[i] = live ? v[i] : e[i];
                m = live ? m + 1 : m;
            }
            e[NH] = 0.f;
        }
        PBRE_UNROLL for (int k = 0; k < ND; k++) e[k] = found ? ex[k] : e[k];
        return used;
    }
    // Closed form of `n` sweeps over the object-table rows (see the simple class's solver section in step_t for the derivation and the
    // validity bound).  rx, ry, rz: scaled lever arms of the NK slots; dinv = 1 / |J|^2 per row (0: slot unused, its rows are no-ops);
    // rhs: the normal rows' right-hand sides (already multiplied by dinv); app: impulses applied so far; v, u: linear velocity and scaled
    // angular velocity after the explicit sweeps.  xo: the twist after n more sweeps.  Returns whether no clamp can bind in them.
    // PERLANE (round 6, Bullet's residual exit): `n` differs from lane to lane (0 .. iters - OC_K: the sweeps between the explicit ones and the
    // lane's exit sweep) -- the squarings run until every lane's exponent is used up and the 16th power (the bound's contraction factor) has
    // been seen, a lane applies a power when its bit is set.
    template <bool PERLANE = false>
    static PBRE_HD bool obj_closed(const float (&rx)[NC_OT], const float (&ry)[NC_OT], const float (&rz)[NC_OT], const float (&dinv)[NC_OT][3],
                                   const float (&rhs)[NC_OT], const float (&app)[NC_OT][3], float mu, V3 v, V3 u, int n, float (&xo)[6]) {
        constexpr int NK = NC_OT;
        // row (c, d) as x <- x + (beta - dinv (J.x)) J with J = [dir ; r' x dir]: dir = +z, -y, +x
        auto Jdot = [&](int c, int d, const float* x) -> float {
            if (d == 0) return x[2] + ry[c] * x[3] - rx[c] * x[4];
            if (d == 1) return -x[1] + rz[c] * x[3] - rx[c] * x[5];
            return x[0] + rz[c] * x[4] - ry[c] * x[5];
        };
        auto Jaxpy = [&](int c, int d, float a, float* x) {
            if (d == 0) { x[2] += a; x[3] = fmaf(a, ry[c], x[3]); x[4] = fmaf(-a, rx[c], x[4]); }
            else if (d == 1) { x[1] -= a; x[3] = fmaf(a, rz[c], x[3]); x[5] = fmaf(-a, rx[c], x[5]); }
            else { x[0] += a; x[4] = fmaf(a, rz[c], x[4]); x[5] = fmaf(-a, ry[c], x[5]); }
        };
        // one sweep as an affine map: columns 0..5 of B = S, column 6 = s (row order: the 4 normals, then the friction pairs)
        float B[7][6];
        PBRE_UNROLL for (int k = 0; k < 7; k++) PBRE_UNROLL for (int i = 0; i < 6; i++) B[k][i] = (i == k) ? 1.f : 0.f;
        auto rowop = [&](int c, int d) {
            PBRE_UNROLL for (int k = 0; k < 7; k++) {
                const float t = Jdot(c, d, B[k]);
                const float beta = (k == 6 && d == 0) ? rhs[c] : 0.f;
                Jaxpy(c, d, fmaf(-dinv[c][d], t, beta), B[k]);
            }
        };
        PBRE_UNROLL for (int c = 0; c < NK; c++) rowop(c, 0);
        PBRE_UNROLL for (int c = 0; c < NK; c++) { rowop(c, 1); rowop(c, 2); }
        float x[6] = {v.x, v.y, v.z, u.x, u.y, u.z};
        const float xk[6] = {v.x, v.y, v.z, u.x, u.y, u.z};
        auto apply = [&](const float (*M)[6]) {
            float y[6];
            PBRE_UNROLL for (int i = 0; i < 6; i++) {
                float a = M[6][i];
                PBRE_UNROLL for (int k = 0; k < 6; k++) a = fmaf(M[k][i], x[k], a);
                y[i] = a;
            }
            PBRE_UNROLL for (int i = 0; i < 6; i++) x[i] = y[i];
        };
        auto square = [&](const float (*M)[6], float (*Q)[6]) {      // Q = M o M: S^2, S s + s
            PBRE_UNROLL for (int k = 0; k < 7; k++)
                PBRE_UNROLL for (int i = 0; i < 6; i += 2) {
                    float a0 = k == 6 ? M[6][i] : 0.f, a1 = k == 6 ? M[6][i + 1] : 0.f;
                    PBRE_UNROLL for (int j = 0; j < 6; j++) PBRE_FMA2(M[k][j], M[j][i], M[j][i + 1], a0, a1);
                    Q[k][i] = a0; Q[k][i + 1] = a1;
                }
        };
        auto fro2 = [&](const float (*M)[6]) { float a = 0.f; PBRE_UNROLL for (int k = 0; k < 6; k++) PBRE_UNROLL for (int i = 0; i < 6; i++) a = fmaf(M[k][i], M[k][i], a); return a; };
        float C[7][6];
        float sig2 = 4.f;                  // |S^16|_F^2 (taken when the running power is 16; n >= 32 makes sure it is reached)
        int r = n, pw = 1;
        if constexpr (PERLANE) {
            auto apply_if = [&](const float (*M)[6], bool on) {
                float y[6];
                PBRE_UNROLL for (int i = 0; i < 6; i++) {
                    float a = M[6][i];
                    PBRE_UNROLL for (int k = 0; k < 6; k++) a = fmaf(M[k][i], x[k], a);
                    y[i] = a;
                }
                PBRE_UNROLL for (int i = 0; i < 6; i++) x[i] = on ? y[i] : x[i];
            };
            for (;;) {
                apply_if(B, (r & 1) != 0);
                if (pw == 16) sig2 = fro2(B);
                r >>= 1; if (!PBRE_ANY(r != 0) && pw >= 16) break;
                square(B, C); pw <<= 1;
                apply_if(C, (r & 1) != 0);
                if (pw == 16) sig2 = fro2(C);
                r >>= 1; if (!PBRE_ANY(r != 0) && pw >= 16) break;
                square(C, B); pw <<= 1;
            }
        } else
        for (;;) {
            if (r & 1) apply(B);
            if (pw == 16) sig2 = fro2(B);
            r >>= 1; if (!r) break;
            square(B, C); pw <<= 1;
            if (r & 1) apply(C);
            if (pw == 16) sig2 = fro2(C);
            r >>= 1; if (!r) break;
            square(C, B); pw <<= 1;
        }
        PBRE_UNROLL for (int i = 0; i < 6; i++) xo[i] = x[i];
        // ---- the bound
        float rho2 = 0.f;
        PBRE_UNROLL for (int i = 0; i < 6; i++) { const float e = xk[i] - x[i]; rho2 = fmaf(e, e, rho2); }
        float eta[NK][3], E = 0.f;
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            const float jb = sqrtf(1.f + fmaf(rx[c], rx[c], fmaf(ry[c], ry[c], rz[c] * rz[c])));      // |J_r| <= jb for the three rows of the slot
            PBRE_UNROLL for (int d = 0; d < 3; d++) {
                eta[c][d] = fabsf(fmaf(-dinv[c][d], Jdot(c, d, x), d == 0 ? rhs[c] : 0.f));
                E = fmaf(eta[c][d], jb, E);
            }
        }
        const float sigma = sqrtf(sig2), nf = (float)n;
        // sum over the n sweeps of |y_t| (re-derived with the per-block growth term outside the 1 / (1 - sigma) factor): at block
        // boundaries Y_m <= sigma^m rho + 16 E / (1 - sigma), inside a block |y| <= Y_m + j E (j = 0..15: 120 E per block = 7.5 E per sweep)
        const float T = fmaf(16.f, fmaf(nf, E, sqrtf(rho2)) / (1.f - sigma), 8.f * nf * E);
        bool ok = sigma < 0.9f && T >= 0.f;          // (a NaN anywhere fails one of the comparisons below)
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            const bool used = dinv[c][0] != 0.f;
            const float nmin = app[c][0] - 2.f * fmaf(nf, eta[c][0], T);
            bool okc = nmin > 0.f;
            PBRE_UNROLL for (int d = 1; d < 3; d++) okc = okc && (fabsf(app[c][d]) + 2.f * fmaf(nf, eta[c][d], T) <= mu * nmin);
            ok = ok && (okc || !used);
        }
        return ok;
    }

    // RT: pbre_physics.solver_residual_threshold > 0 (Bullet's exit test of the sweep loop, see step_t); sw: where the number of sweeps the
    // env ran goes (Params::sweeps + the env's local index), or null
    // TT: `Tables` or CTables, the same struct in the constant address space (pbre_capi.hip k_fused: there the tables pointer is not a
    // __restrict__ kernel argument every store is known not to alias, and through a plain pointer the model constants came in as per-lane
    // vector loads of a uniform address -- 50 x4 loads whose results were spilled, 488 B of scratch per lane -- instead of scalar loads)
    template <bool RT = false, class TT = Tables>
    static PBRE_HD int step(const TT& T, const Params& P, float* st, const float* act, float* out, int mode, int flags,
                            unsigned long long env_id, const float* tgt, int* sw = nullptr) {
        if (st[46] != 0.f) return skipped(T, P, st, out, mode, flags, env_id);
        return step_t<false, 0, RT>(T, P, st, act, out, mode, flags, env_id, tgt, nullptr, 0, sw);
    }
    // action_repeat > 1: the env left the apply_action loop in an earlier iteration of this env.step() (`if self._termination():
    // break`, panda_push_gym_env.py:239-240; flag X[14]): no simulation step, only the evaluation of the state it is in
    template <class TT = Tables>
    static PBRE_HD int skipped(const TT& T, const Params& P, float* st, float* out, int mode, int flags, unsigned long long env_id) {
        float q[ND], qd[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = st[j]; qd[j] = st[16 + j]; }
        V3 op = v3(st[9], st[10], st[11]);
        Q4 oq; oq.x = st[12]; oq.y = st[13]; oq.z = st[14]; oq.w = st[15];
        return finish(T, P, st, q, qd, op, oq, out, mode, flags, env_id);
    }
    template <bool RT = false, class TT = Tables>
    static PBRE_HD int step_rc(const TT& T, const Params& P, float* st, const float* act, float* out, int mode, int flags,
                               unsigned long long env_id = 0, const float* tgt = nullptr, int* sw = nullptr) {
        if (st[46] != 0.f) return skipped(T, P, st, out, mode, flags, env_id);
        return step_t<true, 0, RT>(T, P, st, act, out, mode, flags, env_id, tgt, nullptr, 0, sw);
    }
    // ROLE (simple class only): 0 the whole step on one lane (k_fast); 1 / 2 the pair kernel's split of the same step over two waves of
    // a block -- the robot's half (kinematics, dynamics, motor rows, integration, then the observation of the new state) and the
    // object's half (contact candidates, the 150 sweeps over its rows, integration).  In this class the two halves share no unknown, so
    // the split changes no operand of any operation: the results are those of ROLE 0 bit for bit (GPU test).  The object wave hands the
    // new object pose over in LDS (px, lane ln); small batches -- every wave alone on its SIMD -- step in max(robot, object) + the
    // observation instead of their sum.
    // RT (pbre_physics.solver_residual_threshold > 0; PyBullet's solverResidualThreshold, Bullet's m_leastSquaresResidualThreshold
    // [EXT-UNVERIFIED], oracle: orc_params.solver_residual_threshold): the env leaves the sweep loop after the first sweep whose largest
    // velocity-level row change |delta impulse / jacDiagABInv| -- over ALL of its rows: motors, limits, normals, frictions -- is
    // <= P.res_lim.  The test couples the blocks of the simple class, so the motor rows run sequentially next to the object's rows (no
    // closed form, no split over two waves); a lane that has left the loop keeps a snapshot of its velocities while its wave-mates go on
    // (what an env computes does not depend on the lanes it shares a wave with).  The sweeps run are reported through `sw`.
    template <bool RC, int ROLE = 0, bool RT = false, class TT = Tables>
    static PBRE_HD int step_t(const TT& T, const Params& P, float* st, const float* act, float* out, int mode, int flags,
                              unsigned long long env_id, const float* tgt, PairX* px = nullptr, int ln = 0, int* sw = nullptr) {
        static_assert(ROLE == 0 || !RC, "the pair kernel steps the simple class");
        static_assert(ROLE == 0 || !RT, "the residual test is a maximum over all rows of an env: one lane steps the whole env");
        constexpr bool ROBOT = ROLE != 2, OBJECT = ROLE != 1;
        constexpr int NR = RC ? NC_RO + NC_RT : 1;      // robot-contact slots: [0, NC_RO) object, [NC_RO, NR) table
        constexpr int NKO = RC ? NC_RO : 1, NKT = RC ? NC_RT : 1;
        const bool obj_on = !(flags & 1);
        const float dt = P.dt, inv_dt = P.inv_dt;
        PBRE_PROBE_DECL
        constexpr int PB = ROLE == 2 ? 24 : 16;      // probe slots: 16.. one-lane step / robot wave, 24.. object wave
        float q[ND], qd[ND];
        if (ROBOT) { PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = st[j]; qd[j] = st[16 + j]; } }
        V3 op = v3(0.f, 0.f, 0.f);
        Q4 oq; oq.x = 0.f; oq.y = 0.f; oq.z = 0.f; oq.w = 1.f;
        if (OBJECT) { op = v3(st[9], st[10], st[11]); oq.x = st[12]; oq.y = st[13]; oq.z = st[14]; oq.w = st[15]; }     // (the robot wave gets the NEW pose, from the object wave)
        M3 Ro = quat_R(oq);
        // per-env object parameters (domain randomisation, pbre_set_physics_per_env): X[12] mass, X[13] lateral friction,
        // X[15] 1 + linear damping; 0 = the batch value.  The inertia of the (cube) object scales with its mass.
        const float o_m = st[44] > 0.f ? st[44] : P.obj_m, o_mu = st[45] > 0.f ? st[45] : P.obj_mu, o_kl = st[47] > 0.f ? st[47] - 1.f : P.kl;
        // ... and the robot links' linear damping (change_physics_params' robot_damping, panda_push_gym_env.py:365-367): V[15] = 1 + damping
        const float r_kl = st[31] > 0.f ? st[31] - 1.f : P.kl;

        // ---- one forward sweep over the links: FK, joint axes, velocities, velocity-product accelerations,
        //      collision-sphere distances, per-link bias force and spatial inertia (world frame, about the world origin)
        V3 Sa[ND], Sl[ND];
        // bias torques tau_i = -S_i . (bias wrench of i's subtree): accumulated link by link in the forward sweep (link j's wrench projected
        // on the axes of j and of its ancestors: 44 pairs, ~160 more FMAs than summing subtree wrenches backwards) -- so that no per-link
        // wrench (54 floats) has to stay alive until the backward sweep
        float tau[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) tau[j] = 0.f;
        float Cm[ND]; V3 Ch[ND]; float CI[ND][6];
        Cand kO[NKO], kT[NKT];   // the NC_RO robot-object / NC_RT robot-table candidates closest to contact, best first
        auto none = [&](Cand& k) { k.dist = 3e38f; k.idx = 99; k.mu = 0.f; k.owner = 0; k.n = k.pA = k.pB = v3(0.f, 0.f, 0.f); };
        PBRE_UNROLL for (int i = 0; i < NKO; i++) none(kO[i]);
        PBRE_UNROLL for (int i = 0; i < NKT; i++) none(kT[i]);
        if (ROBOT) {
            M3 R[ND]; V3 p[ND]; V3 Va[ND], Vl[ND], Aa[ND], Al[ND];
            const V3 oh = v3(P.obj_h[0], P.obj_h[1], P.obj_h[2]);
            const V3 tc = v3(P.tab_c[0], P.tab_c[1], P.tab_c[2]), th = v3(P.tab_h[0], P.tab_h[1], P.tab_h[2]);
            M3 Id; PBRE_UNROLL for (int k = 0; k < 9; k++) Id.m[k] = (k % 4 == 0) ? 1.f : 0.f;
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                const int pj = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
                const bool root = Topo::parent(j) < 0;
                V3 ax = v3(T.axis[0][j], T.axis[1][j], T.axis[2][j]);
                M3 R0; PBRE_UNROLL for (int k = 0; k < 9; k++) R0.m[k] = T.R0[k][j];
                V3 p0 = v3(T.p0[0][j], T.p0[1][j], T.p0[2][j]);
                M3 Rl; V3 pl;
                if (Topo::jtype(j) == 1) {
                    float c, sn; sincos_(q[j], sn, c); const float C = 1.f - c;
                    M3 Rj;
                    Rj.m[0] = c + ax.x*ax.x*C;       Rj.m[1] = ax.x*ax.y*C - ax.z*sn; Rj.m[2] = ax.x*ax.z*C + ax.y*sn;
                    Rj.m[3] = ax.y*ax.x*C + ax.z*sn; Rj.m[4] = c + ax.y*ax.y*C;       Rj.m[5] = ax.y*ax.z*C - ax.x*sn;
                    Rj.m[6] = ax.z*ax.x*C - ax.y*sn; Rj.m[7] = ax.z*ax.y*C + ax.x*sn; Rj.m[8] = c + ax.z*ax.z*C;
                    Rl = mm(R0, Rj); pl = p0;
                } else {
                    Rl = R0; V3 d = mv(R0, ax); pl = v3(fmaf(d.x, q[j], p0.x), fmaf(d.y, q[j], p0.y), fmaf(d.z, q[j], p0.z));
                }
                if (root) { R[j] = Rl; p[j] = pl; } else { R[j] = mm(R[pj], Rl); p[j] = add(p[pj], mv(R[pj], pl)); }
                V3 aw = mv(R[j], ax);
                if (Topo::jtype(j) == 1) { Sa[j] = aw; Sl[j] = cross(p[j], aw); } else { Sa[j] = v3(0.f, 0.f, 0.f); Sl[j] = aw; }
                V3 sa = scl(Sa[j], qd[j]), sl = scl(Sl[j], qd[j]);
                if (root) { Va[j] = sa; Vl[j] = sl; } else { Va[j] = add(Va[pj], sa); Vl[j] = add(Vl[pj], sl); }
                V3 ca = cross(Va[j], sa), cl = add(cross(Va[j], sl), cross(Vl[j], sa));
                if (root) { Aa[j] = ca; Al[j] = v3(cl.x, cl.y, cl.z - P.gz); } else { Aa[j] = add(Aa[pj], ca); Al[j] = add(Al[pj], cl); }
                // robot collision spheres owned by this link vs object / table (the simple class has none in range)
                if (RC) for (int s = 0; s < T.nspheres; s++) {
                    if (T.s_owner[s] != j) continue;
                    V3 sc = add(p[j], mv(R[j], v3(T.s_c[0][s], T.s_c[1][s], T.s_c[2][s])));
                    {
                        Cand c; c.idx = s; c.owner = j;
                        if (obj_on) {
                            c.dist = sphere_box(sc, T.s_r[s], op, Ro, oh, c.n, c.pB); c.pA = add(c.pB, scl(c.n, c.dist));
                            c.mu = T.s_mu[s] * o_mu;
                            keepn(c, P.margin, kO);
                        }
                        c.dist = sphere_box(sc, T.s_r[s], tc, Id, th, c.n, c.pB); c.pA = add(c.pB, scl(c.n, c.dist));
                        c.mu = T.s_mu[s] * P.tab_mu;
                        keepn(c, P.margin, kT);
                    }
                }
                V3 Faj = v3(0.f, 0.f, 0.f), Flj = v3(0.f, 0.f, 0.f); Cm[j] = 0.f; Ch[j] = v3(0.f, 0.f, 0.f);
                PBRE_UNROLL for (int k = 0; k < 6; k++) CI[j][k] = 0.f;
                PBRE_UNROLL for (int b = 0; b < NSUB; b++) {
                    if (b >= Topo::nsub(j)) continue;
                    const float m = T.sb_m[b][j];
                    V3 c = add(p[j], mv(R[j], v3(T.sb_c[b][0][j], T.sb_c[b][1][j], T.sb_c[b][2][j])));
                    M3 Il; Il.m[0] = T.sb_I[b][0][j]; Il.m[1] = T.sb_I[b][3][j]; Il.m[2] = T.sb_I[b][4][j];
                    Il.m[3] = Il.m[1]; Il.m[4] = T.sb_I[b][1][j]; Il.m[5] = T.sb_I[b][5][j]; Il.m[6] = Il.m[2]; Il.m[7] = Il.m[5]; Il.m[8] = T.sb_I[b][2][j];
                    M3 RI = mm(R[j], Il), Iw;
                    PBRE_UNROLL for (int a = 0; a < 3; a++)
                        PBRE_UNROLL for (int bb = 0; bb < 3; bb++)
                            Iw.m[a*3+bb] = fmaf(RI.m[a*3], R[j].m[bb*3], fmaf(RI.m[a*3+1], R[j].m[bb*3+1], RI.m[a*3+2] * R[j].m[bb*3+2]));
                    V3 w = Va[j];
                    V3 vc = add(Vl[j], cross(w, c));
                    V3 ac = add(add(Al[j], cross(Aa[j], c)), cross(w, vc));
                    float sl_ = fmaf(r_kl, norm(vc), r_kl);
                    V3 f = scl(add(ac, scl(vc, sl_)), m);
                    V3 Iww = mv(Iw, w);
                    float sa_ = fmaf(P.ka, norm(w), P.ka);
                    V3 nc = add(add(mv(Iw, Aa[j]), cross(w, Iww)), scl(Iww, sa_));
                    Faj = add(Faj, add(nc, cross(c, f))); Flj = add(Flj, f);
                    Cm[j] += m; Ch[j] = add(Ch[j], scl(c, m));
                    const float cc = dot(c, c);
                    CI[j][0] += fmaf(m, cc - c.x*c.x, Iw.m[0]); CI[j][1] += fmaf(m, cc - c.y*c.y, Iw.m[4]); CI[j][2] += fmaf(m, cc - c.z*c.z, Iw.m[8]);
                    CI[j][3] += fmaf(-m, c.x*c.y, Iw.m[1]); CI[j][4] += fmaf(-m, c.x*c.z, Iw.m[2]); CI[j][5] += fmaf(-m, c.y*c.z, Iw.m[5]);
                }
                PBRE_UNROLL for (int i = 0; i < ND; i++)
                    if (Topo::is_anc(i, j)) tau[i] -= dot(Sa[i], Faj) + dot(Sl[i], Flj);
            }
        }

        PBRE_PROBE(PB + 0);      // forward sweep
        // ---- backward sweep: composite inertias -> mass matrix (CRBA)
        float Mi[ND * (ND + 1) / 2];       // symmetric storage, becomes M^-1
        if (ROBOT) PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) {
            M3 Io; Io.m[0] = CI[j][0]; Io.m[1] = CI[j][3]; Io.m[2] = CI[j][4]; Io.m[3] = CI[j][3]; Io.m[4] = CI[j][1]; Io.m[5] = CI[j][5];
            Io.m[6] = CI[j][4]; Io.m[7] = CI[j][5]; Io.m[8] = CI[j][2];
            V3 Ga = add(mv(Io, Sa[j]), cross(Ch[j], Sl[j]));
            V3 Gl = add(scl(Sl[j], Cm[j]), cross(Sa[j], Ch[j]));
            PBRE_UNROLL for (int i = 0; i < ND; i++) {
                if (i > j) { if (!Topo::is_anc(j, i)) Mi[sym(i, j)] = 0.f; continue; }   // unrelated branches (the two fingers)
                if (Topo::is_anc(i, j)) Mi[sym(j, i)] = dot(Sa[i], Ga) + dot(Sl[i], Gl);
                else Mi[sym(j, i)] = 0.f;
            }
            if (Topo::parent(j) >= 0) {
                const int pp = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
                Cm[pp] += Cm[j]; Ch[pp] = add(Ch[pp], Ch[j]);
                PBRE_UNROLL for (int k = 0; k < 6; k++) CI[pp][k] += CI[j][k];
            }
        }
        // (simple class: M itself is needed once more, for the impulse bound of the motor rows' closed form)
        float M0[ND * (ND + 1) / 2];      // (dead, and never materialised, in the complex class)
        if (!RC && ROBOT) { PBRE_UNROLL for (int i = 0; i < ND * (ND + 1) / 2; i++) M0[i] = Mi[i]; }
        // ---- M^-1 by the symmetric sweep operator (A -> -A^-1), Gauss-Jordan arithmetic on the triangle
        if (ROBOT) PBRE_UNROLL for (int k = 0; k < ND; k++) {
            const float pv = 1.f / Mi[sym(k, k)];
            float b[ND];
            PBRE_UNROLL for (int i = 0; i < ND; i++) b[i] = Mi[sym(i, k)];
            PBRE_UNROLL for (int i = 0; i < ND; i++) {
                if (i == k) continue;
                const float bp = b[i] * pv;
                PBRE_UNROLL for (int j = 0; j <= i; j++) { if (j == k) continue; Mi[sym(i, j)] = fmaf(-bp, b[j], Mi[sym(i, j)]); }
                Mi[sym(i, k)] = bp;
            }
            Mi[sym(k, k)] = -pv;
        }
        if (ROBOT) { PBRE_UNROLL for (int i = 0; i < ND * (ND + 1) / 2; i++) Mi[i] = -Mi[i]; }

        PBRE_PROBE(PB + 1);      // CRBA, M^-1
        // ---- unconstrained joint velocities w = v*; motor rows (btMultiBodyJointMotor) written against the running
        //      velocity w = v* + dv:  t = dinv*w - rhs2 with rhs2 = (kp (q_des - q)/dt + (1 - kd) v*) dinv
        const float vmax = P.vmax;
        WV w;
        PBRE_UNROLL for (int k = ND; k < 2 * ((ND + 1) / 2); k++) wset(w, k, 0.f);
        float m_dinv[ND], m_rhs[ND], m_app[ND], m_t[ND];
        // (simple class: joint angles / velocities are re-read from the state record here rather than kept in registers across the
        // kinematic sweeps and the inversion -- the barrier stops the compiler from reusing the earlier loads)
        if (!RC && ROBOT) {
            PBRE_REG_BARRIER();
            PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = st[j]; qd[j] = st[16 + j]; }
        }
        if (ROBOT) PBRE_UNROLL for (int j = 0; j < ND; j++) {
            float a = 0.f;
            PBRE_UNROLL for (int k = 0; k < ND; k++) a = fmaf(Mi[sym(j, k)], tau[k], a);
            const float wj = clampf(fmaf(dt, a, qd[j]), -vmax, vmax);
            wset(w, j, wj);
            float qdes = T.home[j], kp = T.kp_hold[j], kd = T.kd_hold[j];
            if (mode & M_TGT) qdes = tgt[j];          // IK mode: all joints track the IK solution with the hold gains (panda_env.py:276-282)
            if (mode & M_ACTION) {
                kp = T.kp_act[j]; kd = T.kd_act[j];
                if (j < T.n_act) qdes = clampf(fmaf(act[j], P.act_scale, q[j]), T.lower[j], T.upper[j]);
            }
            m_dinv[j] = 1.f / Mi[sym(j, j)];
            m_t[j] = kp * (qdes - q[j]) * inv_dt + (1.f - kd) * wj;      // the motor's target velocity (btMultiBodyJointMotor)
            m_rhs[j] = m_t[j] * m_dinv[j];
            m_app[j] = 0.f;
        }

        const float mlim = P.motor_imp;
        float lsr = 0.f;      // RT: the sweep's largest |delta impulse / jacDiagABInv| so far (a motor / limit row on joint j: / dinv = * (M^-1)_jj)
        // motor row in delta form: clamp(applied + delta) - applied = clamp(delta, lo - applied, hi - applied), so a row whose clamp
        // does not bind returns Bullet's delta = rhs' - dinv w_j bit for bit -- the same value the closed form reproduces
        auto motor = [&](int j) {
            const float nt = fmaf(-m_dinv[j], wget(w, j), m_rhs[j]);
            const float d = med3(nt, -mlim - m_app[j], mlim - m_app[j]);
            m_app[j] += d;
            if constexpr (RT) lsr = fmaxf(lsr, fabsf(d * Mi[sym(j, j)]));
            PBRE_UNROLL for (int k = 0; k < ND; k++) wset(w, k, fmaf(d, Mi[sym(k, j)], wget(w, k)));
        };
        // can a motor reach its impulse bound within the sweeps (see the solver section below)?  False: the motor rows are clamp-free.
        auto motor_may_clamp = [&](const float* e) -> bool {
            float en = 0.f, lam[ND];
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                float l = 0.f;
                PBRE_UNROLL for (int k = 0; k < ND; k++) l = fmaf(M0[sym(j, k)], e[k], l);
                lam[j] = l; en = fmaf(l, e[j], en);
            }
            bool ov_ = !(en >= 0.f);
            PBRE_UNROLL for (int j = 0; j < ND; j++) ov_ = ov_ || !(fabsf(lam[j]) + sqrtf(en * M0[sym(j, j)]) <= mlim);
            return ov_;
        };
        bool rt_over = true;      // RT, simple class: the same test, for the closed form of Bullet's residual exit in the solver section
        if (!RC && RT && ROBOT) {
            float e[ND];
            PBRE_UNROLL for (int j = 0; j < ND; j++) e[j] = wget(w, j) - m_t[j];
            rt_over = motor_may_clamp(e);
        }
        if (!RC && !RT && ROBOT) {
            // ---- simple class, motor block (see the solver section below for the why and the validity bound)
            const bool want_closed = !(flags & 32) && P.iters >= 4 && !(P.iters & 1);
            bool over = true;
            float wc[ND];
            if (want_closed) {
                float e[ND];
                PBRE_UNROLL for (int j = 0; j < ND; j++) e[j] = wget(w, j) - m_t[j];
                over = motor_may_clamp(e);
                motor_closed(Mi, m_dinv, e, P.iters >> 1);
                PBRE_UNROLL for (int j = 0; j < ND; j++) wc[j] = m_t[j] + e[j];
            }
            if (PBRE_ANY(over)) {
                for (int it = 0; it < P.iters; it += 2) {
                    PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) motor(j);
                    if (it + 1 >= P.iters) break;
                    PBRE_UNROLL for (int j = 0; j < ND; j++) motor(j);
                }
            }
            if (want_closed) { PBRE_UNROLL for (int j = 0; j < ND; j++) wset(w, j, over ? wget(w, j) : wc[j]); }
        }

        PBRE_PROBE(PB + 2);      // motor targets, motor block (closed form)
        // joint-limit rows (btMultiBodyJointLimitConstraint; complex class only): a row exists while the joint is at/over
        // the limit; J = dir e_j, positional rhs -pen*erp/dt, impulse in [0, limit_imp]
        float l_dir[ND], l_rhs[ND], l_app[ND];
        bool any_lim = false;
        PBRE_UNROLL for (int j = 0; j < ND; j++) { l_dir[j] = 0.f; l_rhs[j] = 0.f; l_app[j] = 0.f; }
        if (RC) {
            bool lim = false;
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                const float pl = q[j] - T.lower[j], pu = T.upper[j] - q[j];
                const bool lo_v = pl <= 0.f, up_v = !lo_v && pu <= 0.f;
                l_dir[j] = lo_v ? 1.f : (up_v ? -1.f : 0.f);
                const float pen = lo_v ? pl : pu;
                l_rhs[j] = (lo_v || up_v) ? (-pen * P.erp * inv_dt) * m_dinv[j] : 0.f;
                lim = lim || lo_v || up_v;
            }
            any_lim = PBRE_ANY(lim);
        }

        // ---- object: unconstrained velocity, object-table contacts (normal +z, friction directions -y and +x)
        V3 ov = v3(0.f, 0.f, 0.f), ow = v3(0.f, 0.f, 0.f);
        if (OBJECT) { ov = v3(st[25], st[26], st[27]); ow = v3(st[28], st[29], st[30]); }
        constexpr int NK = NC_OT;
        float c_rx[NK], c_ry[NK], c_rz[NK];
        bool c_act[NK];
        // per row: 1/(J M^-1 J^T) and the accumulated impulse; the normal row also carries its positional rhs.  The object's
        // inertia is isotropic (checked at create time, fast_eligible()), so M^-1 J^T = [dir/m ; (r x dir)/I] needs no storage.
        float r_dinv[NK][3], r_app[NK][3], r_rhs[NK];
        float r_den[RT ? NK : 1][3];      // RT: 1 / dinv of the row (in the scaled units the object rows work in)
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            c_act[c] = false; c_rx[c] = c_ry[c] = c_rz[c] = 0.f; r_rhs[c] = 0.f;
            PBRE_UNROLL for (int d = 0; d < 3; d++) { r_dinv[c][d] = r_app[c][d] = 0.f; if (RT) r_den[RT ? c : 0][d] = 0.f; }
        }
        // The object is solved in scaled coordinates in which it has unit mass and unit (isotropic) inertia: lever arms
        // r' = sk r and angular velocity u = omega / sk with sk = sqrt(m / I), object-table impulses in delta-v units (a = lambda / m).
        // Then J M^-1 J^T = (1 + |r' x dir|^2) / m and a row update is ov += da dir, u += da (r' x dir): no per-row
        // multiplications by 1/m and 1/I in the solver loop.
        const float inv_m = 1.f / o_m, inv_I = P.obj_m / (P.obj_I[0] * o_m);
        const float sk = sqrtf(inv_I / inv_m), inv_sk = 1.f / sk;
        const float mu = o_mu * P.tab_mu;
        // A box with unequal principal inertias (obj_name other than the cube; P.obj_iso == 0): the in-line object rows below assume
        // I_w^-1 = 1/I, so the object's half of the step -- it shares no unknown with the robot rows in this class -- is done by
        // ObjStep (pbre_objstep.hpp: same rows, any principal inertia) and the solver loop runs the robot rows alone.  The envs with
        // robot contacts of such a scene are stepped by the row kernel (launch_step), never by step_t<true>.
        const bool obj_inline = OBJECT && obj_on && (P.obj_iso != 0) && P.obj_shape == 0;      // (round objects -- sphere, cylinder -- are ObjStep's too)
        float o_tw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        ObjStep os;           // RT: the object's rows are swept inside the solver loop below, next to the robot's
        if (OBJECT && obj_on && !obj_inline) {
            const float pose[7] = {op.x, op.y, op.z, oq.x, oq.y, oq.z, oq.w};
            const float tw0[6] = {ov.x, ov.y, ov.z, ow.x, ow.y, ow.z};
            if constexpr (RT) os.setup(P, pose, tw0, o_m, o_mu, o_kl);
            else ObjStep::run_p(P, pose, tw0, o_tw, o_m, o_mu, o_kl);
        }
        if (obj_inline) {
            const float isc = o_m / P.obj_m;
            V3 oI = v3(P.obj_I[0] * isc, P.obj_I[1] * isc, P.obj_I[2] * isc);
            M3 Iinv;
            {
                M3 D; PBRE_UNROLL for (int i = 0; i < 3; i++) { D.m[i*3] = Ro.m[i*3] / oI.x; D.m[i*3+1] = Ro.m[i*3+1] / oI.y; D.m[i*3+2] = Ro.m[i*3+2] / oI.z; }
                PBRE_UNROLL for (int i = 0; i < 3; i++)
                    PBRE_UNROLL for (int j = 0; j < 3; j++)
                        Iinv.m[i*3+j] = fmaf(D.m[i*3], Ro.m[j*3], fmaf(D.m[i*3+1], Ro.m[j*3+1], D.m[i*3+2] * Ro.m[j*3+2]));
            }
            V3 wl = mtv(Ro, ow);
            V3 Lw = mv(Ro, v3(wl.x * oI.x, wl.y * oI.y, wl.z * oI.z));
            float sl_ = fmaf(o_kl, norm(ov), o_kl);
            V3 al = v3(-sl_ * ov.x, -sl_ * ov.y, P.gz - sl_ * ov.z);
            float sa_ = fmaf(P.ka, norm(ow), P.ka);
            V3 tq = sub(scl(cross(ow, Lw), -1.f), scl(Lw, sa_));
            V3 aa = mv(Iinv, tq);
            ov = v3(clampf(fmaf(dt, al.x, ov.x), -vmax, vmax), clampf(fmaf(dt, al.y, ov.y), -vmax, vmax), clampf(fmaf(dt, al.z, ov.z), -vmax, vmax));
            ow = v3(clampf(fmaf(dt, aa.x, ow.x), -vmax, vmax), clampf(fmaf(dt, aa.y, ow.y), -vmax, vmax), clampf(fmaf(dt, aa.z, ow.z), -vmax, vmax));
            // vertices vs support surface; keep the NC_OT smallest distances < margin, ordered by vertex index
            float vd[8]; V3 vr[8];
            const float top = P.tab_c[2] + P.tab_h[2], bot = P.tab_c[2] - P.tab_h[2];
            PBRE_UNROLL for (int v = 0; v < 8; v++) {
                V3 l = v3((v & 1) ? P.obj_h[0] : -P.obj_h[0], (v & 2) ? P.obj_h[1] : -P.obj_h[1], (v & 4) ? P.obj_h[2] : -P.obj_h[2]);
                vr[v] = mv(Ro, l);
                V3 x = add(op, vr[v]);
                const bool in = fabsf(x.x - P.tab_c[0]) <= P.tab_h[0] && fabsf(x.y - P.tab_c[1]) <= P.tab_h[1];
                const float hs = (in && x.z > bot) ? top : P.ground_z;
                vd[v] = x.z - hs;
            }
            int slot = 0;
            float c_dist[NK] = {0.f, 0.f, 0.f, 0.f};
            PBRE_UNROLL for (int v = 0; v < 8; v++) {
                int r = 0;
                PBRE_UNROLL for (int u = 0; u < 8; u++) {
                    if (u == v) continue;
                    const bool before = vd[u] < vd[v] || (vd[u] == vd[v] && u < v);
                    r += (vd[u] < P.margin && before) ? 1 : 0;
                }
                if (vd[v] < P.margin && r < NK) {
                    PBRE_UNROLL for (int c = 0; c < NK; c++) if (slot == c) {
                        c_act[c] = true; c_rx[c] = sk * vr[v].x; c_ry[c] = sk * vr[v].y; c_rz[c] = sk * vr[v].z; c_dist[c] = vd[v];
                    }
                    slot++;
                }
            }
            PBRE_UNROLL for (int c = 0; c < NK; c++) {
                const float rx = c_rx[c], ry = c_ry[c], rz = c_rz[c];
                const V3 Ja[3] = {v3(ry, -rx, 0.f), v3(rz, 0.f, -rx), v3(0.f, rz, -ry)};   // r x dir for dir = +z, -y, +x
                PBRE_UNROLL for (int d = 0; d < 3; d++) {
                    r_dinv[c][d] = c_act[c] ? 1.f / (1.f + dot(Ja[d], Ja[d])) : 0.f;
                    if (RT) r_den[RT ? c : 0][d] = c_act[c] ? 1.f + dot(Ja[d], Ja[d]) : 0.f;
                }
                // setupMultiBodyContactConstraint, restitution 0: only the positional part remains in the rhs because the
                // row is evaluated against the running velocity
                const float pen = c_dist[c] + P.slop;
                r_rhs[c] = c_act[c] ? (pen > 0.f ? -pen * inv_dt : -pen * P.erp * inv_dt) * r_dinv[c][0] : 0.f;
            }
        }

        // ---- robot contacts (RC): dense rows J_r (9 joint DoF) [+ object part for robot-object], B_r = M^-1 J_r^T
        float rc_J[NR][3][ND], rc_B[NR][3][ND], rc_dinv[NR][3], rc_app[NR][3], rc_rhs[NR], rc_mu[NR];
        float rc_den[RT ? NR : 1][3];
        V3 rc_dir[NC_RO][3], rc_rxd[NC_RO][3];
        bool rc_act[NR];
        PBRE_UNROLL for (int c = 0; c < NR; c++) rc_act[c] = false;
        if (RC) {
            PBRE_UNROLL for (int c = 0; c < NR; c++) {
                // slot order = sphere index order among the selected candidates (Bullet walks its manifolds in creation order; the row
                // kernel's select_k ranks them in lane order): slot k is the candidate with the k-th lowest sphere index
                Cand cc; none(cc);
                if (c < NC_RO) {
                    PBRE_UNROLL for (int i = 0; i < NKO; i++) {
                        int below = 0;
                        PBRE_UNROLL for (int j = 0; j < NKO; j++) below += (j != i && kO[j].idx < kO[i].idx) ? 1 : 0;       // (free slots carry idx 99, distinct real ones differ)
                        if (below == c && kO[i].dist < 3e38f) cc = kO[i];
                    }
                } else {
                    PBRE_UNROLL for (int i = 0; i < NKT; i++) {
                        int below = 0;
                        PBRE_UNROLL for (int j = 0; j < NKT; j++) below += (j != i && kT[j].idx < kT[i].idx) ? 1 : 0;
                        if (below == c - NC_RO && kT[i].dist < 3e38f) cc = kT[i];
                    }
                }
                rc_act[c] = cc.dist < 3e38f;
                rc_mu[c] = rc_act[c] ? cc.mu : 0.f;
                const V3 n = cc.n;
                V3 t1, t2;     // btPlaneSpace1
                if (fabsf(n.z) > 0.70710678118654752f) {
                    const float a = n.y*n.y + n.z*n.z, kk = 1.f / sqrtf(fmaxf(a, 1e-30f));
                    t1 = v3(0.f, -n.z * kk, n.y * kk); t2 = v3(a * kk, -n.x * t1.z, n.x * t1.y);
                } else {
                    const float a = n.x*n.x + n.y*n.y, kk = 1.f / sqrtf(fmaxf(a, 1e-30f));
                    t1 = v3(-n.y * kk, n.x * kk, 0.f); t2 = v3(-n.z * t1.y, n.z * t1.x, a * kk);
                }
                const V3 rB = sub(cc.pB, op);
                PBRE_UNROLL for (int d = 0; d < 3; d++) {
                    const V3 dir = d == 0 ? n : (d == 1 ? t1 : t2);
                    PBRE_UNROLL for (int j = 0; j < ND; j++) {
                        bool onchain = false;      // joint j moves the contact link (compile-time tree, per-lane owner)
                        PBRE_UNROLL for (int e = 0; e < ND; e++) if (Topo::is_anc(j, e) && cc.owner == e) onchain = true;
                        rc_J[c][d][j] = (rc_act[c] && onchain) ? dot(dir, add(Sl[j], cross(Sa[j], cc.pA))) : 0.f;
                    }
                    float denom = 0.f;
                    PBRE_UNROLL for (int kx = 0; kx < ND; kx++) {
                        float b = 0.f;
                        PBRE_UNROLL for (int j = 0; j < ND; j++) b = fmaf(Mi[sym(kx, j)], rc_J[c][d][j], b);
                        rc_B[c][d][kx] = b; denom = fmaf(rc_J[c][d][kx], b, denom);
                    }
                    if (c < NC_RO) {
                        const int co = c < NC_RO ? c : 0;
                        const V3 rxd = cross(rB, dir);
                        rc_dir[co][d] = rc_act[c] ? dir : v3(0.f, 0.f, 0.f); rc_rxd[co][d] = rc_act[c] ? scl(rxd, sk) : v3(0.f, 0.f, 0.f);
                        denom += fmaf(dot(rxd, rxd), inv_I, inv_m);
                    }
                    rc_dinv[c][d] = rc_act[c] ? 1.f / denom : 0.f;
                    if (RT) rc_den[RT ? c : 0][d] = rc_act[c] ? denom : 0.f;
                    rc_app[c][d] = 0.f;
                }
                const float pen = cc.dist + P.slop;
                rc_rhs[c] = rc_act[c] ? (pen > 0.f ? -pen * inv_dt : -pen * P.erp * inv_dt) * rc_dinv[c][0] : 0.f;
            }
        }

        PBRE_PROBE(PB + 3);      // object: unconstrained velocity, candidates, rows (+ robot-contact rows)
        // ---- projected Gauss-Seidel, Bullet order (motors reversed on even iterations, forward on odd; normals; frictions)
        ow = scl(ow, inv_sk);                 // scaled angular velocity u inside the solver loop
        const float llim = P.limit_imp;
        auto limit = [&](int j) {
            const float t = fmaf(m_dinv[j] * l_dir[j], wget(w, j), -l_rhs[j]);
            const float s = med3(l_app[j] - t, 0.f, llim);
            const float d = (s - l_app[j]) * l_dir[j]; l_app[j] = s;
            if constexpr (RT) lsr = fmaxf(lsr, fabsf(d * Mi[sym(j, j)]));
            PBRE_UNROLL for (int k = 0; k < ND; k++) wset(w, k, fmaf(d, Mi[sym(k, j)], wget(w, k)));
        };
        auto orow = [&](int c, int d) {
            const float rx = c_rx[c], ry = c_ry[c], rz = c_rz[c];
            float jv;
            if (d == 0) jv = ov.z + ry * ow.x - rx * ow.y;
            else if (d == 1) jv = -ov.y + rz * ow.x - rx * ow.z;
            else jv = ov.x + rz * ow.y - ry * ow.z;
            float s;
            if (d == 0) s = med3(r_app[c][0] - fmaf(jv, r_dinv[c][0], -r_rhs[c]), 0.f, 1e10f);
            else {
                const float hi = mu * r_app[c][0];
                s = med3(r_app[c][d] - jv * r_dinv[c][d], -hi, hi);
                s = hi > 0.f ? s : r_app[c][d];
            }
            const float dd = s - r_app[c][d]; r_app[c][d] = s;
            if constexpr (RT) lsr = fmaxf(lsr, fabsf(dd * r_den[RT ? c : 0][d]));
            if (d == 0) { ov.z += dd; ow.x = fmaf(dd, ry, ow.x); ow.y = fmaf(-dd, rx, ow.y); }
            else if (d == 1) { ov.y -= dd; ow.x = fmaf(dd, rz, ow.x); ow.z = fmaf(-dd, rx, ow.z); }
            else { ov.x += dd; ow.y = fmaf(dd, rz, ow.y); ow.z = fmaf(-dd, ry, ow.z); }
        };
        auto rrow = [&](int c, int d) {       // robot contact row (RC only)
            float jv = 0.f;
            PBRE_UNROLL for (int j = 0; j < ND; j++) jv = fmaf(rc_J[c][d][j], wget(w, j), jv);
            if (c < NC_RO) { const int co = c < NC_RO ? c : 0; jv -= dot(rc_dir[co][d], ov) + dot(rc_rxd[co][d], ow); }
            float s;
            if (d == 0) s = med3(rc_app[c][0] - fmaf(jv, rc_dinv[c][0], -rc_rhs[c]), 0.f, 1e10f);
            else {
                const float hi = rc_mu[c] * rc_app[c][0];
                s = med3(rc_app[c][d] - jv * rc_dinv[c][d], -hi, hi);
                s = hi > 0.f ? s : rc_app[c][d];
            }
            const float dd = s - rc_app[c][d]; rc_app[c][d] = s;
            if constexpr (RT) lsr = fmaxf(lsr, fabsf(dd * rc_den[RT ? c : 0][d]));
            PBRE_UNROLL for (int k = 0; k < ND; k++) wset(w, k, fmaf(dd, rc_B[c][d][k], wget(w, k)));
            if (c < NC_RO) {
                const int co = c < NC_RO ? c : 0;
                const float dm = -dd * inv_m;      // robot-contact impulses stay in impulse units (the robot side needs them)
                ov.x = fmaf(dm, rc_dir[co][d].x, ov.x); ov.y = fmaf(dm, rc_dir[co][d].y, ov.y); ov.z = fmaf(dm, rc_dir[co][d].z, ov.z);
                ow.x = fmaf(dm, rc_rxd[co][d].x, ow.x); ow.y = fmaf(dm, rc_rxd[co][d].y, ow.y); ow.z = fmaf(dm, rc_rxd[co][d].z, ow.z);
            }
        };
        bool any_c[NK], any_r[NR];
        PBRE_UNROLL for (int c = 0; c < NK; c++) any_c[c] = PBRE_ANY(c_act[c]);    // wave-uniform; inactive slots are exact no-ops
        PBRE_UNROLL for (int c = 0; c < NR; c++) any_r[c] = RC && PBRE_ANY(rc_act[c]);
        auto contacts = [&]() {      // Bullet: all normals (object-table, robot-object, robot-table), then all frictions
            PBRE_UNROLL for (int c = 0; c < NK; c++) if (any_c[c]) orow(c, 0);
            if (RC) { PBRE_UNROLL for (int c = 0; c < NR; c++) if (any_r[c]) rrow(c, 0); }
            PBRE_UNROLL for (int c = 0; c < NK; c++) if (any_c[c]) { orow(c, 1); orow(c, 2); }
            if (RC) { PBRE_UNROLL for (int c = 0; c < NR; c++) if (any_r[c]) { rrow(c, 1); rrow(c, 2); } }
        };
        int used = P.iters;
        if constexpr (RT) {
            const bool obj_sep = OBJECT && obj_on && !obj_inline;
            bool done = false;
            WV w_k = w; V3 ov_k = ov, ow_k = ow;
            float os_k[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // end of sweep `it`: Bullet's test (btSequentialImpulseConstraintSolver::solveGroupCacheFriendlyIterations: leastSquaresResidual
            // <= m_leastSquaresResidualThreshold, both squared there); true once every lane of the wave has left the loop
            auto sweep_end = [&](int it) -> bool {
                const bool newly = !done && lsr <= P.res_lim;
                if (PBRE_ANY(newly)) {
                    PBRE_UNROLL for (int k = 0; k < ND; k++) wset(w_k, k, newly ? wget(w, k) : wget(w_k, k));
                    ov_k = v3(newly ? ov.x : ov_k.x, newly ? ov.y : ov_k.y, newly ? ov.z : ov_k.z);
                    ow_k = v3(newly ? ow.x : ow_k.x, newly ? ow.y : ow_k.y, newly ? ow.z : ow_k.z);
                    if (obj_sep) {
                        os_k[0] = newly ? os.vx : os_k[0]; os_k[1] = newly ? os.vy : os_k[1]; os_k[2] = newly ? os.vz : os_k[2];
                        os_k[3] = newly ? os.wx : os_k[3]; os_k[4] = newly ? os.wy : os_k[4]; os_k[5] = newly ? os.wz : os_k[5];
                    }
                    used = newly ? it + 1 : used;
                    done = done || newly;
                }
                return !PBRE_ANY(!done);
            };
            // the sweeps, with the contact rows of a sweep as `rows`: the usual wave (simple class, cube resting on all four object-table slots
            // in every lane) gets straight-line rows, as in the loop without the exit test -- the per-slot "does any lane use it" branches made
            // the compiler duplicate the loop body per branch combination (16 k instructions, spills inside the loop)
            auto run = [&](auto&& rows) {
                for (int it = 0; it < P.iters; it += 2) {
                    lsr = 0.f;
                    if (ROBOT) { PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) motor(j); }
                    if (RC && any_lim) { PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) limit(j); }
                    rows();
                    if (sweep_end(it)) break;
                    if (it + 1 >= P.iters) break;
                    lsr = 0.f;
                    if (RC && any_lim) { PBRE_UNROLL for (int j = 0; j < ND; j++) limit(j); }
                    if (ROBOT) { PBRE_UNROLL for (int j = 0; j < ND; j++) motor(j); }
                    rows();
                    if (sweep_end(it + 1)) break;
                }
            };
            bool all_slots = true;
            PBRE_UNROLL for (int c = 0; c < NK; c++) all_slots = all_slots && any_c[c];
            // ---- Round 6: the exit sweep and the state at it WITHOUT running every sweep (simple class, cube resting on all four slots).
            // The blocks share no unknown, so sweep `it` passes Bullet's test iff the motor rows' largest change and the object rows' largest
            // change both are <= res_lim.  (a) Object: the OC_K explicit sweeps of the step without the exit test, with the test of each
            // recorded (bit `it` of opass); from OC_K on the rows are the clamp-free Kaczmarz passes of obj_closed, whose changes only shrink:
            // a lane qualifies when sweep OC_K - 1 passes and obj_closed's bound holds.  (b) Motors: motor_scan finds the first sweep that
            // passes in both blocks and the error there.  (c) Object at that sweep: the matrix power with a per-lane exponent, or -- a lane
            // that leaves within the explicit sweeps -- those sweeps again up to its exit.  A lane that does not qualify (a motor may reach
            // its impulse bound, the cube slides / rocks, a residual that does not decrease) takes the explicit rows below: the wave
            // runs them if any of its lanes needs them, a qualifying lane still takes its closed form, so what a lane computes does not depend
            // on its wave-mates.  PBRE_F_SEQ_MOTORS / PBRE_F_SEQ_OBJECT (either): explicit rows for every lane (validation, A/B).
            bool cl_ok = false;
            float cl_w[RC ? 1 : ND];
            V3 cl_ov = ov, cl_ow = ow;
            int cl_used = P.iters;
            if constexpr (!RC && ROLE == 0) {
#ifndef PBRE_OC_K
#define PBRE_OC_K 22
#endif
                constexpr int OC_K = PBRE_OC_K;
#ifndef PBRE_RT_CLOSED       // (build knob, tools/build_variant.sh: 0 = Bullet's residual exit by explicit rows only, as in round 5)
#define PBRE_RT_CLOSED 1
#endif
                const bool want = PBRE_RT_CLOSED && ROBOT && OBJECT && obj_on && obj_inline && all_slots && !(flags & (32 | 64)) && P.iters >= OC_K + 32 && !(P.iters & 1) && OC_K <= 32;
                if (want) {
                    auto osweep = [&]() {
                        PBRE_UNROLL for (int c = 0; c < NK; c++) orow(c, 0);
                        PBRE_UNROLL for (int c = 0; c < NK; c++) { orow(c, 1); orow(c, 2); }
                    };
                    const V3 ov0 = ov, ow0 = ow;
                    unsigned opass = 0u;
                    for (int it = 0; it < OC_K; it++) {
                        lsr = 0.f;
                        osweep();
                        opass |= (lsr <= P.res_lim) ? (1u << it) : 0u;
                    }
                    bool ok = ((opass >> (OC_K - 1)) & 1u) != 0u && !rt_over;
                    float e[ND];
                    PBRE_UNROLL for (int j = 0; j < ND; j++) e[j] = wget(w, j) - m_t[j];
                    bool mono = true;
                    cl_used = motor_scan(Mi, m_dinv, e, P.iters, P.res_lim, opass, OC_K, mono);
                    ok = ok && mono;
                    PBRE_UNROLL for (int j = 0; j < ND; j++) cl_w[j] = m_t[j] + e[j];
                    const int n_tail = cl_used - OC_K;
                    float xc[6];
                    const bool okc = obj_closed<true>(c_rx, c_ry, c_rz, r_dinv, r_rhs, r_app, mu, ov, ow, n_tail > 0 ? n_tail : 0, xc);
                    ok = ok && (okc || n_tail < 0);
                    cl_ov = v3(xc[0], xc[1], xc[2]); cl_ow = v3(xc[3], xc[4], xc[5]);
                    // the rows start over: for the lanes that left within the explicit sweeps (their twist after `cl_used` sweeps), and for
                    // the explicit path below
                    ov = ov0; ow = ow0;
                    PBRE_UNROLL for (int c = 0; c < NK; c++) PBRE_UNROLL for (int d = 0; d < 3; d++) r_app[c][d] = 0.f;
                    if (PBRE_ANY(ok && n_tail < 0)) {
                        for (int it = 0; it < OC_K - 1; it++) {
                            osweep();
                            const bool at = n_tail < 0 && it + 1 == cl_used;
                            cl_ov = v3(at ? ov.x : cl_ov.x, at ? ov.y : cl_ov.y, at ? ov.z : cl_ov.z);
                            cl_ow = v3(at ? ow.x : cl_ow.x, at ? ow.y : cl_ow.y, at ? ow.z : cl_ow.z);
                        }
                        ov = ov0; ow = ow0;
                        PBRE_UNROLL for (int c = 0; c < NK; c++) PBRE_UNROLL for (int d = 0; d < 3; d++) r_app[c][d] = 0.f;
                    }
                    cl_ok = ok;
                    PBRE_OC_PROBE(ok);
                    PBRE_RT_PROBE((((opass >> (OC_K - 1)) & 1u) != 0u ? 0 : 1) | (rt_over ? 2 : 0) | (mono ? 0 : 4) | ((okc || n_tail < 0) ? 0 : 8), st);
                }
            }
            // (RC || ...: in the complex-class kernel the test is constant -- and must be WRITTEN as one: with `if (PBRE_ANY(!cl_ok))` alone,
            // cl_ok never set, k_fast_rc<., RT> left the sweep loop early on the GPU, sweep counts 55 / 1 where the emulation and the oracle say
            // 69 / 48; found by tools/rt_rc_probe.py, variant builds B / C of round 6)
#ifndef PBRE_RT_NO_FALLBACK      // (diagnostic build knob: 1 = the explicit rows are never run beside the closed form -- timing A/B only)
#define PBRE_RT_NO_FALLBACK 0
#endif
            if (RC || ROLE != 0 || (!PBRE_RT_NO_FALLBACK && PBRE_ANY(!cl_ok))) {
            if (obj_sep) run([&]() { lsr = fmaxf(lsr, os.sweep_res()); });
            else if (!RC && all_slots) run([&]() {
                PBRE_UNROLL for (int c = 0; c < NK; c++) orow(c, 0);
                PBRE_UNROLL for (int c = 0; c < NK; c++) { orow(c, 1); orow(c, 2); }
            });
            else run([&]() { contacts(); });
            PBRE_UNROLL for (int k = 0; k < ND; k++) wset(w, k, done ? wget(w_k, k) : wget(w, k));
            ov = v3(done ? ov_k.x : ov.x, done ? ov_k.y : ov.y, done ? ov_k.z : ov.z);
            ow = v3(done ? ow_k.x : ow.x, done ? ow_k.y : ow.y, done ? ow_k.z : ow.z);
            if (obj_sep) {
                os.vx = done ? os_k[0] : os.vx; os.vy = done ? os_k[1] : os.vy; os.vz = done ? os_k[2] : os.vz;
                os.wx = done ? os_k[3] : os.wx; os.wy = done ? os_k[4] : os.wy; os.wz = done ? os_k[5] : os.wz;
                os.result(P, o_tw);
            }
            }
            if constexpr (!RC && ROLE == 0) {
                PBRE_UNROLL for (int k = 0; k < ND; k++) wset(w, k, cl_ok ? cl_w[k] : wget(w, k));
                ov = v3(cl_ok ? cl_ov.x : ov.x, cl_ok ? cl_ov.y : ov.y, cl_ok ? cl_ov.z : ov.z);
                ow = v3(cl_ok ? cl_ow.x : ow.x, cl_ok ? cl_ow.y : ow.y, cl_ok ? cl_ow.z : ow.z);
                used = cl_ok ? cl_used : used;
            }
            if (sw) *sw = used;
        } else
        if (!RC && !OBJECT) {
        } else if (!RC) {
            // ---- simple class.  The motor rows and the object rows share no unknown, so Bullet's interleaved sweeps give each block
            // exactly the iterates it would get alone: the two blocks are solved one after the other -- the motor block earlier, right
            // after the motors' targets were formed (motor_block above: its matrices are dead before the object's rows are built).
            // (1) Motor block.  While no motor reaches its impulse bound, a motor row is the linear map e <- (I - A_j e_j^T / A_jj) e on the
            // error e = w - t (A = M^-1, t = the motors' target velocities, the fixed point), so the `iters` sweeps are a power of one
            // matrix: w = t + G^(iters/2) (w0 - t) with G = one reversed + one forward sweep -- evaluated by repeated squaring
            // (motor_closed) instead of 150 x 9 dependent row updates.  Same numbers as the sequential sweeps up to rounding (the
            // correction term G^k e decays geometrically, so its rounding errors are far below those of 1350 sequential row updates);
            // PBRE_F_SEQ_MOTORS (flags bit 5) runs Bullet's sequential rows instead (validation, A/B).
            // Validity: the rows are clamp-free only while every applied impulse stays within +-motor_imp.  Every row update minimises the
            // energy 1/2 l^T A l - b^T l along one coordinate, so |l_k - l*|_A never grows: |l_k,j| <= |l*_j| + |l*|_A sqrt(M_jj) with
            // l* = M (t - w0), |l*|_A^2 = (t - w0)^T l*.  A lane that fails the bound (a NaN fails it too) takes the sequential clamping rows;
            // what a lane computes does not depend on the lanes it shares a wave with.
            // (2) Object block: 4 normal rows, then the friction pairs.  The usual wave has all four object-table slots in use (the
            // cube rests on the table in every env): that case gets its own copy of the loop without the per-slot "does any lane use
            // it" branches (rows of a lane without the contact are exact no-ops either way).
            bool all_slots = true, no_slot = true;
            PBRE_UNROLL for (int c = 0; c < NK; c++) { all_slots = all_slots && any_c[c]; no_slot = no_slot && !any_c[c]; }
            if (all_slots) {
                auto osweep = [&]() {
                    PBRE_UNROLL for (int c = 0; c < NK; c++) orow(c, 0);
                    PBRE_UNROLL for (int c = 0; c < NK; c++) { orow(c, 1); orow(c, 2); }
                };
                // (two sweeps per trip: a row's new applied impulse lands in a fresh register, with one sweep per trip every row pays a
                // register move at the back edge)
                // Closed form of the tail of the sweeps (PBRE_F_SEQ_OBJECT, flags bit 6, runs all of them row by row: validation, A/B).
                // In the scaled coordinates of this block (unit mass, unit inertia) an object-table row has J M^-1 J^T = |J|^2 and its
                // unclamped update  x <- x + (rhs' - (J.x) / |J|^2) J  is the ORTHOGONAL PROJECTION of the twist x onto the row's
                // hyperplane: a sweep is one pass of Kaczmarz's method, an affine map x <- S x + s whose every factor is non-expansive
                // in the Euclidean norm.  After OC_K explicit sweeps the remaining N = iters - OC_K are applied as (S, s)^N by binary
                // powering (~3 k FMAs instead of N x 12 dependent rows), PROVIDED no clamp can bind in any of them.  The test, per lane:
                // with y = x - x~ (x~: the closed form's own result, the system's solution up to rounding), a row maps y to
                // P_r y + eta_r J_r, eta_r = the row's delta at x~ (0 for a consistent system), so |y| grows by at most E = sum |eta_r| |J_r|
                // per sweep and contracts by sigma = |S^16|_F < 1 per 16 sweeps; a row's delta is eta_r - (J_r.y) / |J_r|^2, at most
                // |eta_r| + |y| in size (|J_r| >= 1: its linear part is a unit vector).  With rho = |x_K - x~|: at the 16-sweep block
                // boundaries |y| <= Y_m = sigma^m rho + 16 E / (1 - sigma), inside a block |y| <= Y_m + j E (j = 0..15), so the sum of |y|
                // over the N sweeps is at most 16 rho / (1 - sigma) + 16 N E / (1 - sigma) + 7.5 N E, and an applied impulse moves by at
                // most  mov_r = N |eta_r| + T,  T = 16 (rho + N E) / (1 - sigma) + 8 N E,
                // from its value after the explicit sweeps.  If every normal impulse stays positive (app_n - mov_n > 0) and every friction
                // impulse inside its cone (|app_f| + mov_f <= mu (app_n - mov_n)) under these bounds -- doubled, for the rounding of the
                // bound itself -- no clamp binds and the closed form IS the sequence of rows up to rounding; a lane that fails (cube
                // sliding, tipping, in flight; NaN) takes the explicit rows for the remaining sweeps.  What a lane computes does not
                // depend on the lanes it shares a wave with.
#ifndef PBRE_OC_K        // explicit sweeps before the closed form (even; A/B and the acceptance probe tools/oc_accept_probe.py)
#define PBRE_OC_K 22
#endif
                constexpr int OC_K = PBRE_OC_K;
                const bool want_oc = !(flags & 64) && P.iters >= OC_K + 32 && !(P.iters & 1);
                const int it_explicit = want_oc ? OC_K : P.iters;
                for (int it = 0; it < it_explicit; it += 2) {
                    osweep();
                    if (it + 1 >= it_explicit) break;
                    osweep();
                }
                if (want_oc) {
                    float xc[6];
                    const bool okc = obj_closed(c_rx, c_ry, c_rz, r_dinv, r_rhs, r_app, mu, ov, ow, P.iters - OC_K, xc);
                    PBRE_OC_PROBE(okc);
#ifndef PBRE_OC_NO_FALLBACK      // (diagnostic build knob: 1 = a lane that fails the bound keeps its 22-sweep state -- timing A/B only)
#define PBRE_OC_NO_FALLBACK 0
#endif
                    if (!PBRE_OC_NO_FALLBACK && PBRE_ANY(!okc)) {
                        for (int it = OC_K; it < P.iters; it += 2) { osweep(); osweep(); }
                    }
                    ov = v3(okc ? xc[0] : ov.x, okc ? xc[1] : ov.y, okc ? xc[2] : ov.z);
                    ow = v3(okc ? xc[3] : ow.x, okc ? xc[4] : ow.y, okc ? xc[5] : ow.z);
                }
            } else if (!no_slot) {
                for (int it = 0; it < P.iters; it++) contacts();
            }
        } else
        for (int it = 0; it < P.iters; it += 2) {
            PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) motor(j);
            if (RC && any_lim) { PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) limit(j); }
            contacts();
            if (it + 1 >= P.iters) break;
            if (RC && any_lim) { PBRE_UNROLL for (int j = 0; j < ND; j++) limit(j); }
            PBRE_UNROLL for (int j = 0; j < ND; j++) motor(j);
            contacts();
        }

        PBRE_PROBE(PB + 4);      // the sweeps
        ow = scl(ow, sk);
        if (OBJECT && obj_on && !obj_inline) { ov = v3(o_tw[0], o_tw[1], o_tw[2]); ow = v3(o_tw[3], o_tw[4], o_tw[5]); }
        // ---- integrate.  Positions are re-read from the state record (still the old values) rather than kept in
        //      registers across the solver loop; the barrier stops the compiler from reusing the earlier loads.
        PBRE_REG_BARRIER();
        // NaN / Inf guard.  fin_r / fin_o: 0 while every entry of the robot's / the object's part of the INCOMING state
        // (re-read here: still the old values) is finite, NaN otherwise (x * 0 is NaN for x = NaN or +-Inf).  They are ADDED to one
        // position of the new state, so a non-finite input -- which the solver's clamps (v_med3, v_max: they return the other
        // operand) would otherwise turn into finite garbage -- leaves a NaN that finish() finds, flags and, with PBRE_F_AUTO_RESET,
        // restarts the env from.
        float fin_r = 0.f, fin_o = 0.f;
        if (ROBOT) PBRE_UNROLL for (int j = 0; j < ND; j++) {
            const float v = clampf(wget(w, j), -vmax, vmax);
            const float q0 = st[j];
            if (PBRE_NAN_GUARD) { fin_r = fmaf(q0, 0.f, fin_r); fin_r = fmaf(st[16 + j], 0.f, fin_r); }
            qd[j] = v; q[j] = fmaf(dt, v, q0);
            st[16 + j] = v;
        }
        if (ROBOT) { q[0] += fin_r; PBRE_UNROLL for (int j = 0; j < ND; j++) st[j] = q[j]; }
        if (OBJECT && obj_on) {
            op = v3(st[9], st[10], st[11]);
            oq.x = st[12]; oq.y = st[13]; oq.z = st[14]; oq.w = st[15];
            if (PBRE_NAN_GUARD) {
                fin_o = fmaf(op.x, 0.f, fmaf(op.y, 0.f, fmaf(op.z, 0.f, fmaf(oq.x, 0.f, fmaf(oq.y, 0.f, fmaf(oq.z, 0.f, oq.w * 0.f))))));
                PBRE_UNROLL for (int k = 25; k < 31; k++) fin_o = fmaf(st[k], 0.f, fin_o);
            }
            ov = v3(clampf(ov.x, -vmax, vmax), clampf(ov.y, -vmax, vmax), clampf(ov.z, -vmax, vmax));
            ow = v3(clampf(ow.x, -vmax, vmax), clampf(ow.y, -vmax, vmax), clampf(ow.z, -vmax, vmax));
            op = v3(fmaf(dt, ov.x, op.x) + fin_o, fmaf(dt, ov.y, op.y), fmaf(dt, ov.z, op.z));
            float ang = norm(ow);
            if (ang * dt > 0.78539816339744831f) ang = 0.78539816339744831f * inv_dt;
            float sh, ch;
            sincos_(0.5f * ang * dt, sh, ch);
            const float sc_ = ang < 0.001f ? 0.5f * dt - dt * dt * dt * 0.020833333333f * ang * ang : sh / ang;
            Q4 dq; dq.x = ow.x * sc_; dq.y = ow.y * sc_; dq.z = ow.z * sc_; dq.w = ch;
            Q4 nq = qmul(dq, oq);
            const float in = 1.f / sqrtf(nq.x*nq.x + nq.y*nq.y + nq.z*nq.z + nq.w*nq.w);
            oq.x = nq.x * in; oq.y = nq.y * in; oq.z = nq.z * in; oq.w = nq.w * in;
            st[9] = op.x; st[10] = op.y; st[11] = op.z; st[12] = oq.x; st[13] = oq.y; st[14] = oq.z; st[15] = oq.w;
            st[25] = ov.x; st[26] = ov.y; st[27] = ov.z; st[28] = ow.x; st[29] = ow.y; st[30] = ow.z;
        }
        PBRE_PROBE(PB + 5);      // integration
        if (ROLE == 2) {         // the object's new pose for the robot wave's observation
            px->o[0][ln] = op.x; px->o[1][ln] = op.y; px->o[2][ln] = op.z;
            px->o[3][ln] = oq.x; px->o[4][ln] = oq.y; px->o[5][ln] = oq.z; px->o[6][ln] = oq.w;
            return 0;
        }
        // observation / reward / termination of the new state, and its class for the next step.  The model constants
        // are re-read after the solver loop instead of keeping ~130 of them live across it.
        const TT* T2 = &T;
        PBRE_LAUNDER(T2);
        const CTables* T4 = (const CTables*)T2;
        return finish<ROLE>(*T4, P, st, q, qd, op, oq, out, mode, flags, env_id, !RC, px, ln);
    }

    // Class of a state (same distance arithmetic as the contact candidates of step_t<true>):
    //   0 simple; 1 limit rows only; 2 one robot-table contact; 3 two robot-table contacts; 4 one robot-object contact;
    //   5 any other combination.
    // Classes 1..5 are all stepped by step_t<true>; they exist so that a wave of the compacted complex list is homogeneous
    // and the wave-uniform "any lane uses this row slot" tests skip the row slots nobody in the wave needs.
    // NCLASS = 3 (default since round 4): class 1 = complex without, class 2 = complex WITH a robot collision sphere within the margin of
    // the object.  The second kind (~5 % of the complex envs) solves the coupled system -- ~390 instead of ~150 instructions per sweep
    // in the row kernel -- and a row wave pays for the rows of its heaviest group: listed apart they get waves of their own
    // (k_row_list) instead of slowing three cheap wave-mates down, and their own wave sweeps only the row slots THEY use.
#ifndef PBRE_NCLASS
#define PBRE_NCLASS 3        // 2: all complex envs share one list; 3: see above; 6: one list per class of the table above (pays off only when
#endif                       //    k_fast_rc is throughput-bound)
    static constexpr int NCLASS = PBRE_NCLASS;
    static constexpr int BAD_BIT = 256;      // step() / finish() return value: class | BAD_BIT when the NaN / Inf guard fired (the caller counts)
    static_assert(NCLASS == 2 || NCLASS == 3 || NCLASS == 6, "supported class layouts");
    struct Tail { int cls; M3 Re; V3 pe, Va, Vl; int nT; bool lim; };   // class + end-effector owner frame and spatial velocity (nT, lim: what the class was made of)
    static PBRE_HD int cls_of(int nO, int nT, bool lim) {
        if (nO == 0 && nT == 0) return lim ? 1 : 0;
        if (NCLASS == 2) return 1;
        if (NCLASS == 3) return nO > 0 ? 2 : 1;
        if (!lim && nO == 0) return nT == 1 ? 2 : 3;
        if (!lim && nT == 0 && nO == 1) return 4;
        return 5;
    }

    // One streaming sweep over the links (a link's frame is dropped as soon as its children are done): kinematics, the
    // class of the state, and -- when qd is given -- the frame and spatial velocity of the end-effector's owner link.
    // `bounds`: try cheap wave-wide lower bounds before the exact sphere-box distances (the simple-env kernel, where every lane is
    // normally far from any contact; on the few waves of the complex-env kernels the extra tests would only add latency)
    // ROLE 1 (robot wave of the pair kernel): the new object pose is not known yet -- the sphere centres are parked in LDS and tested
    // against the object by sweep_object() once the object wave has delivered it; cls then only counts the table and the limits.
    template <int ROLE = 0, class TT = Tables>
    static PBRE_HD Tail sweep(const TT& T, const Params& P, const float* q, const float* qd, V3 op, Q4 oq, int flags, bool bounds = false,
                              PairX* px = nullptr, int ln = 0) {
        const bool obj_on = !(flags & 1);
        Tail t;
        bool lim = false;
        PBRE_UNROLL for (int j = 0; j < ND; j++) lim = lim || (q[j] - T.lower[j] <= 0.f) || (T.upper[j] - q[j] <= 0.f);
        const M3 Ro = quat_R(oq);
        const V3 oh = v3(P.obj_h[0], P.obj_h[1], P.obj_h[2]);
        const V3 tc = v3(P.tab_c[0], P.tab_c[1], P.tab_c[2]), th = v3(P.tab_h[0], P.tab_h[1], P.tab_h[2]);
        M3 Id; PBRE_UNROLL for (int k = 0; k < 9; k++) Id.m[k] = (k % 4 == 0) ? 1.f : 0.f;
        const float orad = sqrtf(dot(oh, oh)), ztop = tc.z + th.z;
        int nO = 0, nT = 0;
        const int eo = T.ee_owner;
        t.Va = v3(0.f, 0.f, 0.f); t.Vl = v3(0.f, 0.f, 0.f); t.pe = v3(0.f, 0.f, 0.f);
        PBRE_UNROLL for (int k = 0; k < 9; k++) t.Re.m[k] = 0.f;
        M3 R[ND]; V3 p[ND];
        V3 pk0 = v3(0.f, 0.f, 0.f), pk1 = pk0;      // ROLE 3: this lane's sphere(s) of the new state (<= 32 spheres: s and s + 16)
        float pr0 = 0.f, pr1 = 0.f;
        PBRE_UNROLL for (int j = 0; j < ND; j++) {
            const int pj = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
            V3 ax = v3(T.axis[0][j], T.axis[1][j], T.axis[2][j]);
            M3 R0; PBRE_UNROLL for (int k = 0; k < 9; k++) R0.m[k] = T.R0[k][j];
            V3 p0 = v3(T.p0[0][j], T.p0[1][j], T.p0[2][j]);
            M3 Rl; V3 pl;
            if (Topo::jtype(j) == 1) {
                float c, sn; sincos_(q[j], sn, c); const float C = 1.f - c;
                M3 Rj;
                Rj.m[0] = c + ax.x*ax.x*C;       Rj.m[1] = ax.x*ax.y*C - ax.z*sn; Rj.m[2] = ax.x*ax.z*C + ax.y*sn;
                Rj.m[3] = ax.y*ax.x*C + ax.z*sn; Rj.m[4] = c + ax.y*ax.y*C;       Rj.m[5] = ax.y*ax.z*C - ax.x*sn;
                Rj.m[6] = ax.z*ax.x*C - ax.y*sn; Rj.m[7] = ax.z*ax.y*C + ax.x*sn; Rj.m[8] = c + ax.z*ax.z*C;
                Rl = mm(R0, Rj); pl = p0;
            } else {
                Rl = R0; V3 d = mv(R0, ax); pl = v3(fmaf(d.x, q[j], p0.x), fmaf(d.y, q[j], p0.y), fmaf(d.z, q[j], p0.z));
            }
            if (Topo::parent(j) < 0) { R[j] = Rl; p[j] = pl; } else { R[j] = mm(R[pj], Rl); p[j] = add(p[pj], mv(R[pj], pl)); }
            for (int s = 0; s < T.nspheres; s++) {
                if (T.s_owner[s] != j) continue;
                V3 sc = add(p[j], mv(R[j], v3(T.s_c[0][s], T.s_c[1][s], T.s_c[2][s])));
                // cheap lower bounds on the two distances first (bounding sphere of the object; height above the table top); the
                // exact sphere-box tests (a square root each) only run if some lane of the wave is not clearly far
                const float sr = T.s_r[s];
                if (ROLE == 1) { px->sc[3 * s][ln] = sc.x; px->sc[3 * s + 1][ln] = sc.y; px->sc[3 * s + 2][ln] = sc.z; }
                else if (ROLE == 3) {      // row wave: lane s mod 16 of the group keeps sphere s, the tests follow the loop -- one pass for all spheres
                    const bool mine = (s & 15) == ln;
                    if (s < 16) { if (mine) { pk0 = sc; pr0 = sr; } } else if (mine) { pk1 = sc; pr1 = sr; }
                    continue;
                }
                else if (obj_on) {
                    const V3 dd = sub(sc, op);
                    const float reach = sr + P.margin + orad;
                    if ((!bounds || PBRE_ANY(!(dot(dd, dd) >= reach * reach))) && sphere_obj_dist(P, sc, sr, op, Ro, oh) < P.margin) nO++;
                }
                if ((!bounds || PBRE_ANY(!(sc.z - sr - ztop >= P.margin))) && sphere_box_dist(sc, sr, tc, Id, th) < P.margin) nT++;
            }
            if (qd) {
                bool anc = false;      // is j an ancestor-or-self of the EE owner?  (compile-time tree, uniform runtime owner)
                PBRE_UNROLL for (int e = 0; e < ND; e++) if (Topo::is_anc(j, e) && eo == e) anc = true;
                if (anc) {
                    V3 aw = mv(R[j], ax);
                    if (Topo::jtype(j) == 1) { t.Va = add(t.Va, scl(aw, qd[j])); t.Vl = add(t.Vl, scl(cross(p[j], aw), qd[j])); }
                    else t.Vl = add(t.Vl, scl(aw, qd[j]));
                }
                if (eo == j) { t.Re = R[j]; t.pe = p[j]; }
            }
        }
        if (ROLE == 3) {
            // the 16 lanes of the group ran the same kinematics on the same inputs; each now tests the sphere(s) it kept -- the same two tests on
            // the same operands as lane-per-env, one pass instead of `nspheres` -- and the counts are summed over the group: the same class
            PBRE_UNROLL for (int k = 0; k < 2; k++) {
                const int s = ln + 16 * k;
                if (s < T.nspheres) {
                    const V3 sc = k ? pk1 : pk0; const float sr = k ? pr1 : pr0;
                    if (obj_on && sphere_obj_dist(P, sc, sr, op, Ro, oh) < P.margin) nO++;
                    if (sphere_box_dist(sc, sr, tc, Id, th) < P.margin) nT++;
                }
            }
            nO = PBRE_ROW_SUM_I(nO); nT = PBRE_ROW_SUM_I(nT);
        }
        t.cls = cls_of(nO, nT, lim); t.nT = nT; t.lim = lim;
        return t;
    }
    // the object half of sweep<1>'s classification: the parked sphere centres against the object pose (same tests, same operands)
    template <class TT>
    static PBRE_HD int sweep_object(const TT& T, const Params& P, V3 op, Q4 oq, bool bounds, const PairX* px, int ln) {
        const M3 Ro = quat_R(oq);
        const V3 oh = v3(P.obj_h[0], P.obj_h[1], P.obj_h[2]);
        const float orad = sqrtf(dot(oh, oh));
        int nO = 0;
        for (int s = 0; s < T.nspheres; s++) {
            const V3 sc = v3(px->sc[3 * s][ln], px->sc[3 * s + 1][ln], px->sc[3 * s + 2][ln]);
            const float sr = T.s_r[s];
            const V3 dd = sub(sc, op);
            const float reach = sr + P.margin + orad;
            if ((!bounds || PBRE_ANY(!(dot(dd, dd) >= reach * reach))) && sphere_obj_dist(P, sc, sr, op, Ro, oh) < P.margin) nO++;
        }
        return nO;
    }
    // class of the state stored in `st` (after reset / set_state)
    static PBRE_HD int classify_state(const Tables& T, const Params& P, const float* st, int flags) {
        float q[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) q[j] = st[j];
        Q4 oq; oq.x = st[12]; oq.y = st[13]; oq.z = st[14]; oq.w = st[15];
        return rt_class(P, st, flags, sweep(T, P, q, nullptr, v3(st[9], st[10], st[11]), oq, flags).cls);
    }
    // Residual exit (round 6): in the simple class the exit sweep comes from closed forms whose object part needs a cube AT REST on the table.  A
    // cube that moves -- sliding after a push, rocking, in flight -- fails that bound, and on the GPU the lane's whole wave then runs the 150
    // explicit sweeps: one such lane per ~150 k env-steps (tools/rt_why_probe.py) = about one wave per step at 131072 envs, which is the step's
    // tail (0.237 -> 0.20 ms without it, profiles/r06z_rt_ab.txt).  With the exit test on, a state whose object moves is therefore a complex
    // one (class 1: the row kernel's 16 lanes sweep its rows explicitly) -- a function of the env's own state, like every class.  The fallback
    // inside the simple-class kernel stays for what the speed test does not foresee.
    static PBRE_HD int rt_class(const Params& P, const float* st, int flags, int cls) {
        if (!(P.res_lim > 0.f) || cls != 0 || (flags & 1)) return cls;
        const float v2 = fmaf(st[25], st[25], fmaf(st[26], st[26], st[27] * st[27])), w2 = fmaf(st[28], st[28], fmaf(st[29], st[29], st[30] * st[30]));
        return (v2 > 1e-6f || w2 > 1e-4f) ? 1 : 0;       // |v| > 1 mm/s or |w| > 0.01 rad/s (a resting cube: < 1e-5 either)
    }

    // ---------------------------------------------------------------- inverse kinematics (use_IK = 1)
    // Damped-least-squares IK of the end effector from the current joint angles (replaces p.calculateInverseKinematics,
    // reference panda_env.py:269-272: maxNumIterations=100, residualThreshold=1e-3); same algorithm as oracle/orc_ik:
    // e = [p_t - p_ee ; rotvec(R_t R_ee^T)], dq = J^T (J J^T + lambda^2 I)^-1 e on the joints of the EE chain, until |e_pos| < res.
    static PBRE_HD void ik_solve(const Tables& T, const Params& P, const float* q0, V3 pos, V3 eul, float* q) {
        const M3 Rt = quat_R(euler_quat(eul));
        PBRE_UNROLL for (int j = 0; j < ND; j++) q[j] = q0[j];
        const int eo = T.ee_owner;
        M3 Eo; PBRE_UNROLL for (int k = 0; k < 9; k++) Eo.m[k] = T.ee_R[k];
        bool done = false;
        for (int it = 0; it < P.ik_iters; it++) {
            if (!PBRE_ANY(!done)) break;
            M3 R[ND]; V3 p[ND], aw[ND];
            M3 Re = Eo; V3 po = v3(0.f, 0.f, 0.f);
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                const int pj = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
                V3 ax = v3(T.axis[0][j], T.axis[1][j], T.axis[2][j]);
                M3 R0; PBRE_UNROLL for (int k = 0; k < 9; k++) R0.m[k] = T.R0[k][j];
                V3 p0 = v3(T.p0[0][j], T.p0[1][j], T.p0[2][j]);
                M3 Rl; V3 pl;
                if (Topo::jtype(j) == 1) {
                    float c, sn; sincos_(q[j], sn, c); const float C = 1.f - c;
                    M3 Rj;
                    Rj.m[0] = c + ax.x*ax.x*C;       Rj.m[1] = ax.x*ax.y*C - ax.z*sn; Rj.m[2] = ax.x*ax.z*C + ax.y*sn;
                    Rj.m[3] = ax.y*ax.x*C + ax.z*sn; Rj.m[4] = c + ax.y*ax.y*C;       Rj.m[5] = ax.y*ax.z*C - ax.x*sn;
                    Rj.m[6] = ax.z*ax.x*C - ax.y*sn; Rj.m[7] = ax.z*ax.y*C + ax.x*sn; Rj.m[8] = c + ax.z*ax.z*C;
                    Rl = mm(R0, Rj); pl = p0;
                } else {
                    Rl = R0; V3 d = mv(R0, ax); pl = v3(fmaf(d.x, q[j], p0.x), fmaf(d.y, q[j], p0.y), fmaf(d.z, q[j], p0.z));
                }
                if (Topo::parent(j) < 0) { R[j] = Rl; p[j] = pl; } else { R[j] = mm(R[pj], Rl); p[j] = add(p[pj], mv(R[pj], pl)); }
                aw[j] = mv(R[j], ax);
                if (eo == j) { Re = R[j]; po = p[j]; }
            }
            const V3 pe = add(po, mv(Re, v3(T.ee_p[0], T.ee_p[1], T.ee_p[2])));
            const M3 Ree = mm(Re, Eo);
            float e[6];
            e[0] = pos.x - pe.x; e[1] = pos.y - pe.y; e[2] = pos.z - pe.z;
            done = done || sqrtf(e[0]*e[0] + e[1]*e[1] + e[2]*e[2]) < P.ik_res;
            // rotation vector of R_t R_ee^T
            M3 Rerr;
            PBRE_UNROLL for (int a = 0; a < 3; a++)
                PBRE_UNROLL for (int b = 0; b < 3; b++)
                    Rerr.m[a*3+b] = fmaf(Rt.m[a*3], Ree.m[b*3], fmaf(Rt.m[a*3+1], Ree.m[b*3+1], Rt.m[a*3+2] * Ree.m[b*3+2]));
            const float sx = Rerr.m[7] - Rerr.m[5], sy = Rerr.m[2] - Rerr.m[6], sz = Rerr.m[3] - Rerr.m[1];
            const float s2 = 0.5f * sqrtf(sx*sx + sy*sy + sz*sz), c2 = 0.5f * (Rerr.m[0] + Rerr.m[4] + Rerr.m[8] - 1.f);
            const float f = s2 > 1e-9f ? atan2f(s2, c2) / (2.f * s2) : 0.5f;
            e[3] = f * sx; e[4] = f * sy; e[5] = f * sz;
            // Jacobian columns (joints on the EE chain), A = J J^T + lambda^2 I
            float J[6][ND];
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                bool anc = false;
                PBRE_UNROLL for (int k = 0; k < ND; k++) if (Topo::is_anc(j, k) && eo == k) anc = true;
                V3 jl, ja;
                if (Topo::jtype(j) == 1) { jl = cross(aw[j], sub(pe, p[j])); ja = aw[j]; } else { jl = aw[j]; ja = v3(0.f, 0.f, 0.f); }
                J[0][j] = anc ? jl.x : 0.f; J[1][j] = anc ? jl.y : 0.f; J[2][j] = anc ? jl.z : 0.f;
                J[3][j] = anc ? ja.x : 0.f; J[4][j] = anc ? ja.y : 0.f; J[5][j] = anc ? ja.z : 0.f;
            }
            float A[6][6];
            PBRE_UNROLL for (int a = 0; a < 6; a++)
                PBRE_UNROLL for (int b = 0; b <= a; b++) {
                    float sum = a == b ? P.ik_l2 : 0.f;
                    PBRE_UNROLL for (int j = 0; j < ND; j++) sum = fmaf(J[a][j], J[b][j], sum);
                    A[a][b] = sum;
                }
            // in-place Cholesky (lower), forward/back substitution
            float y[6];
            PBRE_UNROLL for (int a = 0; a < 6; a++)
                PBRE_UNROLL for (int b = 0; b <= a; b++) {
                    float sum = A[a][b];
                    PBRE_UNROLL for (int k = 0; k < b; k++) sum = fmaf(-A[a][k], A[b][k], sum);
                    A[a][b] = a == b ? sqrtf(sum) : sum / A[b][b];
                }
            PBRE_UNROLL for (int a = 0; a < 6; a++) { float sum = e[a]; PBRE_UNROLL for (int k = 0; k < a; k++) sum = fmaf(-A[a][k], y[k], sum); y[a] = sum / A[a][a]; }
            PBRE_UNROLL for (int a = 5; a >= 0; a--) { float sum = y[a]; PBRE_UNROLL for (int k = a + 1; k < 6; k++) sum = fmaf(-A[k][a], y[k], sum); y[a] = sum / A[a][a]; }
            bool moved = false;
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                float dq = 0.f;
                PBRE_UNROLL for (int a = 0; a < 6; a++) dq = fmaf(J[a][j], y[a], dq);
                const float qn = done ? q[j] : q[j] + dq;
                moved = moved || qn != q[j];
                q[j] = qn;
            }
            // no joint angle of any env of the wave changed (targets out of reach: the damped step is below half an ulp): every further
            // iteration would repeat this one, leaving gives the same targets bit for bit
            if (!PBRE_ANY(moved)) break;
        }
    }
    // apply_action, IK branch (panda_push_gym_env.py:197-222 + panda_env.py:229-291): accumulate the scaled Cartesian action into
    // the hand pose, clip it to the rotation limits and the robot workspace, solve IK, store the joint targets.
    // reset = true: pandaEnv.reset with use_IK (panda_env.py:83-91): targets of the home hand pose.
    static PBRE_HD void ik_targets(const Tables& T, const Params& P, float* st, const float* act, float* tgt, bool reset) {
        float hp[6];
        if (reset) { PBRE_UNROLL for (int k = 0; k < 6; k++) hp[k] = P.home_hand[k]; }
        else {
            if (st[46] != 0.f) return;       // action_repeat > 1: this env already left the apply_action loop of this env.step()
            PBRE_UNROLL for (int k = 0; k < 3; k++) hp[k] = clampf(fmaf(act[k], P.ik_ps, st[38 + k]), P.rws[k][0], P.rws[k][1]);
            PBRE_UNROLL for (int k = 3; k < 6; k++) hp[k] = clampf(fmaf(act[k], P.ik_rs, st[38 + k]), P.eu_lim[k - 3][0], P.eu_lim[k - 3][1]);
        }
        PBRE_UNROLL for (int k = 0; k < 6; k++) st[38 + k] = hp[k];
        float q0[ND], q[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) q0[j] = st[j];
        ik_solve(T, P, q0, v3(hp[0], hp[1], clampf(hp[2], P.rws[2][0], P.rws[2][1])), v3(hp[3], hp[4], hp[5]), q);
        PBRE_UNROLL for (int j = 0; j < ND; j++) tgt[j] = q[j];
    }

    // counter-based sampling shared with pbre_core.hpp / the oracle (Philox4x32-10 keyed by seed, counter = global env id,
    // episode, stream)
    static PBRE_HD void philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned o[4]) {
        for (int r = 0; r < 10; r++) {
            unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
            unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
    }
    static PBRE_HD float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

    // Observation / reward / termination of the new state and its class.  With PBRE_F_AUTO_RESET a finished env is
    // re-initialised right here (snapshot reset): the transition's reward and done flag are returned
    // together with the first observation of the next episode.
    // TT: `Tables`, or the same struct in the constant address space (CTables).  The second half of a step re-reads the model constants
    // (they are not kept live across the solver loop); through a plain pointer those re-reads come after the step's stores and behind an
    // opaque pointer, so the compiler issued them as per-lane FLAT loads of a uniform address -- ~200 vector loads with a full memory
    // round trip each, 45 k of the robot wave's 138 k cycles on a lone wave (tools/phase_probe.py) -- and even the `owner == link` scan
    // over the collision spheres ran on vector compares.  From the constant address space they are scalar loads again.
    template <int ROLE = 0, class TT = Tables>
    static PBRE_HD int finish(const TT& T, const Params& P, float* st, float* q, float* qd, V3 op, Q4 oq,
                              float* out, int mode, int flags, unsigned long long env_id, bool bounds = false, PairX* px = nullptr, int ln = 0) {
        const bool want_obs = (mode & (M_OBS | M_TASK)) != 0;
        PBRE_PROBE_DECL
        float reward = 0.f, done = 0.f;
        // end-effector pose / velocity of the state (q, qd) and its class, by the streaming sweep
        V3 ee, eul, vee;
        int cls;
        bool first = true;
        auto kin = [&]() {
            Tail tl;
            if (ROLE == 1 && first) {
                // robot wave of the pair kernel: everything of the sweep that does not involve the object, then the block barrier behind
                // which the object wave's new pose is in LDS, then the sphere-object tests
                tl = sweep<1>(T, P, q, want_obs ? qd : nullptr, op, oq, flags, bounds, px, ln);
                if (ROLE == 1) PBRE_PROBE(22);      // robot wave: kinematics of the new state
                PBRE_PAIR_SYNC(px, ln);
                if (ROLE == 1) PBRE_PROBE(23);      // robot wave: waiting for the object wave
                if (!(flags & 1)) {
                    op = v3(px->o[0][ln], px->o[1][ln], px->o[2][ln]);
                    oq.x = px->o[3][ln]; oq.y = px->o[4][ln]; oq.z = px->o[5][ln]; oq.w = px->o[6][ln];
                    tl.cls = cls_of(sweep_object(T, P, op, oq, bounds, px, ln), tl.nT, tl.lim);
                }
            } else if (ROLE == 3) tl = sweep<3>(T, P, q, want_obs ? qd : nullptr, op, oq, flags, false, nullptr, ln);      // (all 16 lanes of a row wave's group)
            else tl = sweep<0>(T, P, q, want_obs ? qd : nullptr, op, oq, flags, bounds);
            first = false;
            cls = tl.cls;
            if (!want_obs) return;
            M3 Eo; PBRE_UNROLL for (int k = 0; k < 9; k++) Eo.m[k] = T.ee_R[k];
            const M3 Ree = mm(tl.Re, Eo);
            ee = add(tl.pe, mv(tl.Re, v3(T.ee_p[0], T.ee_p[1], T.ee_p[2])));
            vee = add(tl.Vl, cross(tl.Va, ee));
            eul = quat_euler(R_quat(Ree));
        };
        kin();
        // NaN / Inf guard: the new state's positions (a non-finite input left its mark there, see step_t; a velocity that diverged to
        // Inf / NaN reaches them through the integration).  Such an env-step is counted (pbre_kernel_info[12]); its transition is
        // returned as reward 0, done 1 and, with PBRE_F_AUTO_RESET, the env restarts from the settled snapshot in this same step --
        // a single diverged env can neither poison a whole rollout silently nor stay dead.
        float fin = 0.f;
        PBRE_UNROLL for (int j = 0; j < ND; j++) fin = fmaf(q[j], 0.f, fin);
        if (!(flags & 1)) fin = fmaf(op.x, 0.f, fmaf(op.y, 0.f, fmaf(op.z, 0.f, fmaf(oq.x, 0.f, fmaf(oq.y, 0.f, fmaf(oq.z, 0.f, fmaf(oq.w, 0.f, fin)))))));
        // (the count itself is the CALLER's: finish() returns the class with BAD_BIT set -- an atomic in the middle of this function made
        // the compiler duplicate the rest of it, +57 % static instructions and +14 % executed VALU in k_fast)
        const bool bad = PBRE_NAN_GUARD && !(fin == 0.f);
        if (!want_obs) return cls | (bad ? BAD_BIT : 0);
        V3 tg = v3(st[32], st[33], st[34]);
        bool again = false;
        if (mode & M_TASK) {
            const float d1 = norm(sub(ee, op)), d2 = norm(sub(op, tg));
            const float dsucc = P.task >= 1 ? d2 : d1;
            const bool succ = dsucc <= P.dist_min;
            float cnt = st[35], term = st[36];
            const float mx = (float)P.max_steps;
            bool left;        // `if self._termination(): break` fired in this iteration of the apply_action loop
            if (P.task == 2) {
                left = cnt > mx;
                cnt = cnt > mx ? cnt : cnt + 1.f;
                done = (succ || cnt > mx) ? 1.f : 0.f;
                reward = succ ? 0.f : -1.f;
            } else {
                const bool d0 = succ || term != 0.f || cnt > mx;
                left = d0;
                cnt = d0 ? cnt : cnt + 1.f;
                term = succ ? 1.f : term;
                done = (succ || term != 0.f || cnt > mx) ? 1.f : 0.f;
                const float base = P.task == 1 ? -d1 - d2 : -d1;
                reward = succ ? 1000.f + (100.f - dsucc * 80.f) : base;
            }
            if (bad) { reward = 0.f; done = 1.f; }
            st[35] = cnt; st[36] = term;
            st[46] = ((mode & M_INNER) && left) ? 1.f : 0.f;      // consumed by the remaining iterations of this env.step(), cleared by its last one
            again = (flags & 2) && !(mode & M_INNER) && done != 0.f;
        }
        if (PBRE_ANY(again)) {
            if (again) {
            // ---- snapshot reset: the settled state of reset_simulation (panda_push_gym_env.py:117-148) is invariant under
            // the sampled object x, y, yaw (flat table, vertical drop), so the next episode starts from the settled robot
            // state and object height recorded at the last full reset, with freshly sampled pose and target
            // (WorldEnv._sample_pose, world_env.py:145-176; sample_tg_pose, panda_push_gym_env.py:333-360).
            const unsigned ep = (unsigned)(int)st[37] + 1u;
            PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = P.rst_q[j]; qd[j] = 0.f; st[j] = q[j]; st[16 + j] = 0.f; }
            const float x_min = P.ws[0][0] + 0.05f, x_max = P.ws[0][1] - 0.1f;
            const float y_min = P.ws[1][0] + 0.05f, y_max = P.ws[1][1] - 0.05f;
            float px = x_min + 0.5f * (x_max - x_min), py = y_min + 0.5f * (y_max - y_min), yaw = 0.78539816339744831f;
            unsigned r[4];
            if (P.obj_std > 0.f) {
                philox((unsigned)env_id, (unsigned)(env_id >> 32), ep, 0u, P.seed_lo, P.seed_hi, r);
                px += -P.obj_std + 2.f * P.obj_std * u01(r[0]);
                py += -P.obj_std + 2.f * P.obj_std * u01(r[1]);
                yaw = -0.78539816339744831f + 1.57079632679489662f * u01(r[2]);
            }
            op = v3(clampf(px, x_min, x_max), clampf(py, y_min, y_max), P.rst_objz);
            oq.x = 0.f; oq.y = 0.f; sincos_(0.5f * yaw, oq.z, oq.w);
            st[9] = op.x; st[10] = op.y; st[11] = op.z; st[12] = oq.x; st[13] = oq.y; st[14] = oq.z; st[15] = oq.w;
            PBRE_UNROLL for (int k = 25; k < 31; k++) st[k] = 0.f;
            if (P.task >= 1) {
                const float tx_min = P.ws[0][0] + 0.07f, tx_max = P.ws[0][1] - 0.07f;
                float tx = op.x + 0.05f, ty = op.y + 0.05f;
                if (P.tg_std > 0.f) {
                    philox((unsigned)env_id, (unsigned)(env_id >> 32), ep, 1u, P.seed_lo, P.seed_hi, r);
                    const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u01(r[1]);
                    const float rad = sqrtf(-2.f * logf(u1)) * P.tg_std;
                    float su, cu;
                    sincos_(6.28318530717958648f * u2, su, cu);
                    tx = op.x + rad * cu;
                    ty = op.y + rad * su;
                }
                st[32] = clampf(tx, tx_min, tx_max); st[33] = clampf(ty, P.ws[1][0], P.ws[1][1]); st[34] = op.z;
                tg = v3(st[32], st[33], st[34]);
            }
            st[35] = 0.f; st[36] = 0.f; st[37] = (float)(int)ep;
            if (P.use_ik) { PBRE_UNROLL for (int k = 0; k < 6; k++) st[38 + k] = P.home_hand[k]; }
            }
            // first observation of the new episode.  The settled robot pose is the same in every env, so its end-effector pose was
            // recorded with the snapshot (P.rst_ee) and the robot is at rest: no second kinematic sweep -- the whole wave would pay
            // for it whenever one of its 64 envs finishes (measured: 12 % of the step at 131 resets per step).  P.rst_ok = 0 (no
            // snapshot yet, or the settled pose is not a simple-class state): the sweep is run again.
            if (P.rst_ok) {
                if (again) { ee = v3(P.rst_ee[0], P.rst_ee[1], P.rst_ee[2]); eul = v3(P.rst_ee[3], P.rst_ee[4], P.rst_ee[5]); vee = v3(0.f, 0.f, 0.f); cls = 0; }
            } else {
                const V3 ee0 = ee, eul0 = eul, vee0 = vee; const int cls0 = cls;
                kin();
                if (!again) { ee = ee0; eul = eul0; vee = vee0; cls = cls0; }
            }
        }
        if (out) {
            const V3 oe = quat_euler(oq);
            const Q4 qh = euler_quat(eul), qo = euler_quat(oe);
            const V3 rel = mtv(quat_R(qh), sub(op, ee));
            Q4 qhi; qhi.x = -qh.x; qhi.y = -qh.y; qhi.z = -qh.z; qhi.w = qh.w;
            const V3 er = quat_euler(qmul(qhi, qo));
            int o = 0;
            out[o++] = ee.x; out[o++] = ee.y; out[o++] = ee.z; out[o++] = eul.x; out[o++] = eul.y; out[o++] = eul.z;
            out[o++] = vee.x / 0.04f; out[o++] = (vee.y - 0.01f) / 0.07f; out[o++] = vee.z / 0.03f;
            PBRE_UNROLL for (int j = 0; j < ND; j++) out[o++] = q[j];
            out[o++] = op.x; out[o++] = op.y; out[o++] = op.z; out[o++] = oe.x; out[o++] = oe.y; out[o++] = oe.z;
            out[o++] = rel.x; out[o++] = rel.y; out[o++] = rel.z; out[o++] = er.x; out[o++] = er.y; out[o++] = er.z;
            if (P.task >= 1) { out[o++] = tg.x; out[o++] = tg.y; out[o++] = tg.z; }
            out[o++] = reward; out[o++] = done;
        }
        if (ROLE == 1) PBRE_PROBE(20);      // robot wave: object tests, observation, reward, row
        if constexpr (ROLE != 1) cls = rt_class(P, st, flags, cls);
        return cls | (bad ? BAD_BIT : 0);
    }
};

}  // namespace pbre
