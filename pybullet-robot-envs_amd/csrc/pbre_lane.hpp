// pbre_lane.hpp -- the iCub's step with one env per lane / per quad of lanes (20 DoF, Shape32 state records), against the lane-group
// kernel of pbre_core.hpp (one env per half-wave: every scalar of a row replicated over 32 lanes, every row ending in a cross-lane
// broadcast).
//
// This header holds the per-env pieces, plain C++ (device: pbre_lane.hip; host: tests/host_emu):
//   dynamics()    kinematics + dynamics of a state: bias torques, joint-space inertia M (CRBA), robot-table contact slots with their
//                 Jacobian rows -- chain by chain over a compile-time tree (TopoICub), everything in registers;
//   finish()      observation / reward / termination / auto-reset of the new state and its class;
//   ik_targets()  Cartesian control: damped-least-squares IK over the (static) chain to the hand;
//   step()        the whole step of one env in one piece: dynamics, M^-1 (sweep operator, in a caller-provided store), motor / limit /
//                 robot-table contact rows, the object's rows (ObjStep) inside the same 150 sweeps, integration, finish.  This is the
//                 form the CPU emulation runs and the reference the device pipeline is checked against.  As a device kernel (M^-1 in
//                 wave-private LDS, 40 KB per wave, one wave per SIMD) it was measured SLOWER than the lane-group kernel -- 1.05 ms
//                 against 0.74 ms per 32768-env step: a lone wave per SIMD exposes every latency -- and is no longer built.
// The device splits the step where the data layouts want to differ (pbre_lane.hip): dynamics() with one thread per env, the solve with
// FOUR lanes per env (each lane owns five DoF = five rows of M^-1, 100 registers, no LDS; a PGS row is one fma, a quad broadcast and
// five fmacs), finish() with one thread per env again.
//
// "Simple" class of this path: motors, joint-limit rows (frequent on the iCub under Cartesian control, where the IK targets push
// joints into their stops), robot-table contacts (with random actions a third to a half of the envs has a hand or forearm on the
// table) and the object's own rows against the table.  An env with a robot collision sphere within the contact margin of the OBJECT
// ("complex", ~1 % of a random-action batch) needs the coupled system: on the device the quad solver with the object's twist and table
// rows in the same sweeps (kw_quad_rc; dynamics() supplies the robot-object contact slots), in the CPU emulation the lane-group core
// (Core::step) followed by finish().
//
// Same mathematics and reference call sites as pbre_core.hpp / pbre_fast.hpp: world-frame RNEA + CRBA + explicit M^-1, Bullet's
// row order, iCubReachGymEnv / iCubPushGymEnv / iCubPushGymGoalEnv .step (icub_reach_gym_env.py:182-259,
// icub_push_gym_env.py:208-282), iCubEnv.apply_action / get_observation (icub_env.py:202-361).
#pragma once
#include <math.h>
#include "pbre_fast.hpp"
#include "pbre_objstep.hpp"
#ifndef PBRE_IK_STORE        // Lane::ik_targets writes target j of its env (device pipeline with the per-env hand-over, pbre_lane.hip: also as a
#define PBRE_IK_STORE(done, seq, j, p, v) (*(p) = (v))      // (value, sequence number) pair into the env's hand-over box `done`; host: the plain store)
#endif

#ifndef PBRE_LANE_MSTRIDE      // floats between consecutive M^-1 entries of one env (device: 64 = [entry][lane]; host: 1)
#define PBRE_LANE_MSTRIDE 1
#endif
#ifndef PBRE_COUNT_BAD       // ++*p from any number of lanes (device: atomicAdd)
#define PBRE_COUNT_BAD(p) (++*(p))
#endif
#ifndef PBRE_OPAQUE_I          // device: the optimiser must assume the (per-lane) int changed (asm volatile("" : "+v"(x))); host: nothing
#define PBRE_OPAQUE_I(x) do {} while (0)
#endif
#ifndef PBRE_OPAQUE_F          // the same for a float: starts a new live range (a value the register allocator spilled during the setup
#define PBRE_OPAQUE_F(x) do {} while (0)   // phase is reloaded once here instead of at every use inside the solver loop)
#endif
#ifndef PBRE_NOUNROLL          // device: _Pragma("nounroll")
#define PBRE_NOUNROLL
#endif
#ifndef PBRE_LANE_MREG         // M^-1 entries (highest triangle indices) kept in registers instead of LDS
#define PBRE_LANE_MREG 0
#endif

PBRE_FP_CONTRACT_FAST      // (pbre_math.hpp: contraction is stated per header)
namespace pbre {

// The iCub as flattened by build_tables() from model/table.py: icub_table (legs pruned): torso 0-1-2, left arm 3..9, head 10..12,
// right arm 13..19, the three branches attached to torso lane 2; lanes 5 and 15 carry two rigid sub-bodies.
struct TopoICub {
    static constexpr int ND = 20;
    static constexpr int parent(int j) { return j == 0 ? -1 : ((j == 10 || j == 13) ? 2 : j - 1); }
    static constexpr int jtype(int) { return 1; }
    static constexpr int nsub(int j) { return (j == 5 || j == 15) ? 2 : 1; }
    static constexpr bool is_anc(int i, int j) {   // i ancestor-or-self of j
        return i == j || (j >= 0 && parent(j) >= 0 && is_anc(i, parent(j)));
    }
    // the tree as chains of consecutive lanes: chain 0 is the trunk (from the root), every other chain hangs on a trunk link
    static constexpr int NCHAIN = 4;
    static constexpr int chain_first(int c) { return c == 0 ? 0 : (c == 1 ? 3 : (c == 2 ? 10 : 13)); }
    static constexpr int chain_last(int c) { return c == 0 ? 2 : (c == 1 ? 9 : (c == 2 ? 12 : 19)); }
    // lanes that can own the end effector (left / right hand): the Cartesian-control code is compiled per candidate, with a static chain
    static constexpr int ee0 = 9, ee1 = 19;
};

template <class Topo, class S>
inline bool lane_topo_matches(const TablesT<S>& T) {
    if (T.ndof != Topo::ND || Topo::ND > S::NJ) return false;
    for (int j = 0; j < Topo::ND; j++) {
        if (T.anc[0][j] != Topo::parent(j) || T.jtype[j] != Topo::jtype(j)) return false;
        for (int b = 0; b < S::NSUB; b++) if ((T.sb_m[b][j] != 0.f || T.sb_I[b][0][j] != 0.f) != (b < Topo::nsub(j))) return false;
    }
    if (T.ee_owner != Topo::ee0 && T.ee_owner != Topo::ee1) return false;
    return true;
}

template <class Topo, class S>
struct Lane {
    using FX = Fast<Topo>;
    using V3 = typename FX::V3; using M3 = typename FX::M3; using Q4 = typename FX::Q4;
    using Tab = TablesT<S>;
    static constexpr int ND = Topo::ND, W = S::W, LC = S::LC, XO = 2 * S::W, NSUB = S::NSUB;
    static constexpr int NM = ND * (ND + 1) / 2;
    static constexpr int MREG = PBRE_LANE_MREG, MLDS = NM - MREG;      // entries [0, MLDS) in LDS, [MLDS, NM) in registers
    static constexpr int MS = PBRE_LANE_MSTRIDE;
    static_assert(!S::MREC && S::NTIP == 0, "task-env shapes only");
    enum { M_ACTION = FX::M_ACTION, M_OBS = FX::M_OBS, M_TASK = FX::M_TASK, M_TGT = FX::M_TGT, M_INNER = FX::M_INNER };

    static PBRE_HD V3 v3(float x, float y, float z) { return FX::v3(x, y, z); }
    static PBRE_HD V3 add(V3 a, V3 b) { return FX::add(a, b); }
    static PBRE_HD V3 sub(V3 a, V3 b) { return FX::sub(a, b); }
    static PBRE_HD V3 scl(V3 a, float s) { return FX::scl(a, s); }
    static PBRE_HD float dot(V3 a, V3 b) { return FX::dot(a, b); }
    static PBRE_HD V3 cross(V3 a, V3 b) { return FX::cross(a, b); }
    static PBRE_HD float norm(V3 a) { return FX::norm(a); }
    static PBRE_HD V3 mv(const M3& A, V3 v) { return FX::mv(A, v); }
    static PBRE_HD V3 mtv(const M3& A, V3 v) { return FX::mtv(A, v); }
    static PBRE_HD M3 mm(const M3& A, const M3& B) { return FX::mm(A, B); }
    static PBRE_HD float clampf(float x, float lo, float hi) { return FX::clampf(x, lo, hi); }
    static PBRE_HD float med3(float x, float lo, float hi) { return FX::med3(x, lo, hi); }
    static constexpr int sym(int i, int j) { return FX::sym(i, j); }

    // M^-1 (during setup: M, then the sweep operator's intermediate) behind one accessor: entry k of this env
    struct Mat {
        float* lds;                                   // this lane's column of the wave's LDS region
        float reg[MREG > 0 ? MREG : 1];
        PBRE_HD float get(int k) const { return k < MLDS ? lds[k * MS] : reg[k < MLDS ? 0 : k - MLDS]; }
        // solver loop: entry k read through an offset `o` (always 0) the optimiser cannot see through -- refreshed per row, so that it
        // neither hoists the loads out of the loop nor merges them with the other use of the entry in the same sweep (either would
        // keep all 210 entries live in registers, which is what the LDS copy is there to avoid)
        PBRE_HD float geto(int k, int o) const { return k < MLDS ? lds[k * MS + o] : reg[k < MLDS ? 0 : k - MLDS]; }
        PBRE_HD void set(int k, float v) { if (k < MLDS) lds[k * MS] = v; else reg[k < MLDS ? 0 : k - MLDS] = v; }
        float rtJ[2][3][ND];                          // robot-table contact Jacobian rows (dynamics() sink)
        PBRE_HD void rt_j(int c, int d, int j, float v) { rtJ[c][d][j] = v; }
        PBRE_HD void rt_none(int c) { PBRE_UNROLL for (int d = 0; d < 3; d++) PBRE_UNROLL for (int j = 0; j < ND; j++) rtJ[c][d][j] = 0.f; }
        PBRE_HD void ro_j(int, int, int, float) {}     // (the one-piece step has no robot-object rows)
        PBRE_HD void put(int j, int i, float v) { set(sym(j, i), v); }      // dynamics() sink
        PBRE_HD void zero(int j, int i) { set(sym(j, i), 0.f); }
    };

    // joint frame of link j in its parent's frame at angle qj
    static PBRE_HD void joint_xf(const Tab& T, int j, float qj, M3& Rl, V3& pl, V3& ax) {
        ax = v3(T.axis[0][j], T.axis[1][j], T.axis[2][j]);
        M3 R0; PBRE_UNROLL for (int k = 0; k < 9; k++) R0.m[k] = T.R0[k][j];
        const V3 p0 = v3(T.p0[0][j], T.p0[1][j], T.p0[2][j]);
        if (Topo::jtype(j) == 1) {
            float c, sn;
            FX::sincos_(qj, sn, c);                   // shared-reduction sin / cos (pbre_fast.hpp: ~28 instructions, no slow path)
            const float C = 1.f - c;
            M3 Rj;
            Rj.m[0] = c + ax.x*ax.x*C;       Rj.m[1] = ax.x*ax.y*C - ax.z*sn; Rj.m[2] = ax.x*ax.z*C + ax.y*sn;
            Rj.m[3] = ax.y*ax.x*C + ax.z*sn; Rj.m[4] = c + ax.y*ax.y*C;       Rj.m[5] = ax.y*ax.z*C - ax.x*sn;
            Rj.m[6] = ax.z*ax.x*C - ax.y*sn; Rj.m[7] = ax.z*ax.y*C + ax.x*sn; Rj.m[8] = c + ax.z*ax.z*C;
            Rl = mm(R0, Rj); pl = p0;
        } else {
            Rl = R0; const V3 d = mv(R0, ax); pl = v3(fmaf(d.x, qj, p0.x), fmaf(d.y, qj, p0.y), fmaf(d.z, qj, p0.z));
        }
    }

    // ------------------------------------------------------------------------------------------------ dynamics
    // Kinematics + dynamics of the state (q, qd): bias torques tau (gravity, velocity products, Bullet's link damping, explicit joint
    // damping) and the joint-space inertia M (CRBA).  sink.put(j, i, v): entry M[j][i] = M[i][j], j >= i, i an ancestor-or-self of j;
    // sink.zero(j, i): a pair on unrelated branches.
    // rt: the (at most NRT) robot collision spheres closest to the table within the contact margin -- contact slots in sphere order, as
    // the lane-group kernel selects them -- with Bullet's contact frame (normal, btPlaneSpace1 tangents), friction coefficient and the
    // Jacobian rows J[c][d][j] = dir_d . (S_l,j + S_a,j x pA) over the joints that move the sphere's link.
    static constexpr int NRT = S::NC_RT;
    static_assert(NRT == 2, "keep2() selects two candidates");
    struct RtC { bool act[NRT]; float dist[NRT], mu[NRT]; };          // (the Jacobian rows go to the sink: sink.rt_j(c, d, j, value))
    // ro (optional; envs of the complex class): the same for the (at most NRO) spheres closest to the OBJECT (box at op, oq): contact
    // frame, lever arm rB = point on the box - op, the robot side's Jacobian rows.
    static constexpr int NRO = 2;
    struct RoC { bool act[NRO]; float dist[NRO], mu[NRO]; V3 dir[NRO][3], rB[NRO]; };      // (rows: sink.ro_j(c, d, j, value))
    template <class Sink>
    static PBRE_HD void dynamics(const Tab& T, const Params& P, const float* q, const float* qd, Sink& sink, float* tau, RtC& rt) {
        dynamics(T, P, q, qd, sink, tau, rt, (RoC*)nullptr, v3(0.f, 0.f, 0.f), Q4{0.f, 0.f, 0.f, 1.f});
    }
    template <class Sink>
    static PBRE_HD void dynamics(const Tab& T, const Params& P, const float* q, const float* qd, Sink& sink, float* tau, RtC& rt,
                                 RoC* ro, V3 op, Q4 oq) {
        typename FX::Cand o1, o2;        // best and second-best robot-object candidate
        o1.dist = o2.dist = 3e38f; o1.idx = o2.idx = 99; o1.mu = o2.mu = 0.f; o1.owner = o2.owner = 0;
        o1.n = o2.n = o1.pA = o2.pA = o1.pB = o2.pB = v3(0.f, 0.f, 0.f);
        const M3 Ro = FX::quat_R(oq);
        const V3 oh = v3(P.obj_h[0], P.obj_h[1], P.obj_h[2]);
        const float orad = sqrtf(dot(oh, oh));
        typename FX::Cand k1, k2;        // best and second-best robot-table candidate
        k1.dist = k2.dist = 3e38f; k1.idx = k2.idx = 99; k1.mu = k2.mu = 0.f; k1.owner = k2.owner = 0;
        k1.n = k2.n = k1.pA = k2.pA = k1.pB = k2.pB = v3(0.f, 0.f, 0.f);
        const V3 tc = v3(P.tab_c[0], P.tab_c[1], P.tab_c[2]), th = v3(P.tab_h[0], P.tab_h[1], P.tab_h[2]);
        M3 Id; PBRE_UNROLL for (int k = 0; k < 9; k++) Id.m[k] = (k % 4 == 0) ? 1.f : 0.f;
        const float ztop = tc.z + th.z;
        // ---- kinematics + dynamics, chain by chain (Topo::chain_*: the trunk, then every branch): forward over the chain's links
        //      (FK, joint axes, velocities, velocity-product accelerations, per-link bias force and spatial inertia; world frame, about
        //      the world origin), then backward over them (subtree forces -> bias torques, composite inertias -> rows of M, CRBA).  A
        //      branch hands its composite to the trunk link it hangs on; the trunk's backward pass runs last.  Only the trunk's and the
        //      current branch's per-link data are live at any time.
        V3 Sa[ND], Sl[ND];                    // joint axes: a link needs those of its ancestors (its chain and the trunk)
        V3 Fa[ND], Fl[ND];
        float Cm[ND]; V3 Ch[ND]; float CI[ND][6];
        M3 R[ND]; V3 p[ND], Va[ND], Vl[ND], Aa[ND], Al[ND];
        auto forward = [&](int j) {
            const int pj = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
            const bool root = Topo::parent(j) < 0;
            M3 Rl; V3 pl, ax;
            joint_xf(T, j, q[j], Rl, pl, ax);
            if (root) { R[j] = Rl; p[j] = pl; } else { R[j] = mm(R[pj], Rl); p[j] = add(p[pj], mv(R[pj], pl)); }
            const V3 aw = mv(R[j], ax);
            if (Topo::jtype(j) == 1) { Sa[j] = aw; Sl[j] = cross(p[j], aw); } else { Sa[j] = v3(0.f, 0.f, 0.f); Sl[j] = aw; }
            for (int sp = 0; sp < T.nspheres; sp++) {          // robot collision spheres of this link vs the table
                if (T.s_owner[sp] != j) continue;
                const V3 sc = add(p[j], mv(R[j], v3(T.s_c[0][sp], T.s_c[1][sp], T.s_c[2][sp])));
                const float sr = T.s_r[sp];
                if (ro) {
                    const V3 dd = sub(sc, op);
                    const float reach = sr + P.margin + orad;
                    if (PBRE_ANY(!(dot(dd, dd) >= reach * reach))) {
                        typename FX::Cand c; c.idx = sp; c.owner = j;
                        c.dist = FX::sphere_obj(P, sc, sr, op, Ro, oh, c.n, c.pB); c.pA = add(c.pB, scl(c.n, c.dist));
                        c.mu = T.s_mu[sp] * P.obj_mu;
                        FX::keep2(c, P.margin, o1, o2);
                    }
                }
                if (!PBRE_ANY(!(sc.z - sr - ztop >= P.margin))) continue;       // cheap wave-wide lower bound first
                typename FX::Cand c; c.idx = sp; c.owner = j;
                c.dist = FX::sphere_box(sc, sr, tc, Id, th, c.n, c.pB); c.pA = add(c.pB, scl(c.n, c.dist));
                c.mu = T.s_mu[sp] * P.tab_mu;
                FX::keep2(c, P.margin, k1, k2);
            }
            const V3 sa = scl(Sa[j], qd[j]), sl = scl(Sl[j], qd[j]);
            if (root) { Va[j] = sa; Vl[j] = sl; } else { Va[j] = add(Va[pj], sa); Vl[j] = add(Vl[pj], sl); }
            const V3 ca = cross(Va[j], sa), cl = add(cross(Va[j], sl), cross(Vl[j], sa));
            if (root) { Aa[j] = ca; Al[j] = v3(cl.x, cl.y, cl.z - P.gz); } else { Aa[j] = add(Aa[pj], ca); Al[j] = add(Al[pj], cl); }
            Fa[j] = v3(0.f, 0.f, 0.f); Fl[j] = v3(0.f, 0.f, 0.f); Cm[j] = 0.f; Ch[j] = v3(0.f, 0.f, 0.f);
            PBRE_UNROLL for (int k = 0; k < 6; k++) CI[j][k] = 0.f;
            PBRE_UNROLL for (int b = 0; b < NSUB; b++) {
                if (b >= Topo::nsub(j)) continue;
                const float m = T.sb_m[b][j];
                const V3 c = add(p[j], mv(R[j], v3(T.sb_c[b][0][j], T.sb_c[b][1][j], T.sb_c[b][2][j])));
                M3 Il; Il.m[0] = T.sb_I[b][0][j]; Il.m[1] = T.sb_I[b][3][j]; Il.m[2] = T.sb_I[b][4][j];
                Il.m[3] = Il.m[1]; Il.m[4] = T.sb_I[b][1][j]; Il.m[5] = T.sb_I[b][5][j]; Il.m[6] = Il.m[2]; Il.m[7] = Il.m[5]; Il.m[8] = T.sb_I[b][2][j];
                const M3 RI = mm(R[j], Il); M3 Iw;
                PBRE_UNROLL for (int a = 0; a < 3; a++)
                    PBRE_UNROLL for (int bb = 0; bb < 3; bb++)
                        Iw.m[a*3+bb] = fmaf(RI.m[a*3], R[j].m[bb*3], fmaf(RI.m[a*3+1], R[j].m[bb*3+1], RI.m[a*3+2] * R[j].m[bb*3+2]));
                const V3 w = Va[j];
                const V3 vc = add(Vl[j], cross(w, c));
                const V3 ac = add(add(Al[j], cross(Aa[j], c)), cross(w, vc));
                const float sl_ = fmaf(P.kl, norm(vc), P.kl);            // Bullet velocity damping K1 + K2 |v|
                const V3 f = scl(add(ac, scl(vc, sl_)), m);
                const V3 Iww = mv(Iw, w);
                const float sa_ = fmaf(P.ka, norm(w), P.ka);
                const V3 nc = add(add(mv(Iw, Aa[j]), cross(w, Iww)), scl(Iww, sa_));
                Fa[j] = add(Fa[j], add(nc, cross(c, f))); Fl[j] = add(Fl[j], f);
                Cm[j] += m; Ch[j] = add(Ch[j], scl(c, m));
                const float cc = dot(c, c);
                CI[j][0] += fmaf(m, cc - c.x*c.x, Iw.m[0]); CI[j][1] += fmaf(m, cc - c.y*c.y, Iw.m[4]); CI[j][2] += fmaf(m, cc - c.z*c.z, Iw.m[8]);
                CI[j][3] += fmaf(-m, c.x*c.y, Iw.m[1]); CI[j][4] += fmaf(-m, c.x*c.z, Iw.m[2]); CI[j][5] += fmaf(-m, c.y*c.z, Iw.m[5]);
            }
        };
        auto backward = [&](int j) {
            tau[j] = -(dot(Sa[j], Fa[j]) + dot(Sl[j], Fl[j])) - T.jdamp[j] * qd[j];      // bias torque + explicit joint damping
            M3 Io; Io.m[0] = CI[j][0]; Io.m[1] = CI[j][3]; Io.m[2] = CI[j][4]; Io.m[3] = CI[j][3]; Io.m[4] = CI[j][1]; Io.m[5] = CI[j][5];
            Io.m[6] = CI[j][4]; Io.m[7] = CI[j][5]; Io.m[8] = CI[j][2];
            const V3 Ga = add(mv(Io, Sa[j]), cross(Ch[j], Sl[j]));
            const V3 Gl = add(scl(Sl[j], Cm[j]), cross(Sa[j], Ch[j]));
            PBRE_UNROLL for (int i = 0; i < ND; i++) {
                if (i > j) { if (!Topo::is_anc(j, i)) sink.zero(i, j); continue; }      // unrelated branches
                if (Topo::is_anc(i, j)) sink.put(j, i, dot(Sa[i], Ga) + dot(Sl[i], Gl)); else sink.zero(j, i);
            }
            if (Topo::parent(j) >= 0) {
                const int pp = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
                Fa[pp] = add(Fa[pp], Fa[j]); Fl[pp] = add(Fl[pp], Fl[j]);
                Cm[pp] += Cm[j]; Ch[pp] = add(Ch[pp], Ch[j]);
                PBRE_UNROLL for (int k = 0; k < 6; k++) CI[pp][k] += CI[j][k];
            }
        };
        PBRE_UNROLL for (int j = Topo::chain_first(0); j <= Topo::chain_last(0); j++) forward(j);
        PBRE_UNROLL for (int c = 1; c < Topo::NCHAIN; c++) {
            PBRE_UNROLL for (int j = Topo::chain_first(c); j <= Topo::chain_last(c); j++) forward(j);
            PBRE_UNROLL for (int j = Topo::chain_last(c); j >= Topo::chain_first(c); j--) backward(j);
        }
        PBRE_UNROLL for (int j = Topo::chain_last(0); j >= Topo::chain_first(0); j--) backward(j);
        // ---- robot-table contact slots
        const bool two = k2.dist < 3e38f;
        const bool swap = two && k2.idx < k1.idx;                      // slot order = sphere index order
        const bool any_rt = PBRE_ANY(k1.dist < 3e38f);
        PBRE_UNROLL for (int c = 0; c < NRT; c++) {
            const typename FX::Cand cc = (c == 0) ? (swap ? k2 : k1) : (swap ? k1 : k2);
            rt.act[c] = cc.dist < 3e38f;
            rt.dist[c] = cc.dist; rt.mu[c] = rt.act[c] ? cc.mu : 0.f;
            if (!any_rt) { sink.rt_none(c); continue; }
            const V3 n = cc.n;
            V3 t1, t2;     // btPlaneSpace1
            if (fabsf(n.z) > 0.70710678118654752f) {
                const float a = n.y*n.y + n.z*n.z, kk = 1.f / sqrtf(fmaxf(a, 1e-30f));
                t1 = v3(0.f, -n.z * kk, n.y * kk); t2 = v3(a * kk, -n.x * t1.z, n.x * t1.y);
            } else {
                const float a = n.x*n.x + n.y*n.y, kk = 1.f / sqrtf(fmaxf(a, 1e-30f));
                t1 = v3(-n.y * kk, n.x * kk, 0.f); t2 = v3(-n.z * t1.y, n.z * t1.x, a * kk);
            }
            PBRE_UNROLL for (int d = 0; d < 3; d++) {
                const V3 dir = d == 0 ? n : (d == 1 ? t1 : t2);
                PBRE_UNROLL for (int j = 0; j < ND; j++) {
                    bool onchain = false;      // joint j moves the contact link (compile-time tree, per-lane owner)
                    PBRE_UNROLL for (int e = 0; e < ND; e++) if (Topo::is_anc(j, e) && cc.owner == e) onchain = true;
                    sink.rt_j(c, d, j, (rt.act[c] && onchain) ? dot(dir, add(Sl[j], cross(Sa[j], cc.pA))) : 0.f);
                }
            }
        }
        // ---- robot-object contact slots (complex class)
        if (ro) {
            const bool two_o = o2.dist < 3e38f;
            const bool swap_o = two_o && o2.idx < o1.idx;
            PBRE_UNROLL for (int c = 0; c < NRO; c++) {
                const typename FX::Cand cc = (c == 0) ? (swap_o ? o2 : o1) : (swap_o ? o1 : o2);
                ro->act[c] = cc.dist < 3e38f;
                ro->dist[c] = cc.dist; ro->mu[c] = ro->act[c] ? cc.mu : 0.f;
                const V3 n = cc.n;
                V3 t1, t2;     // btPlaneSpace1
                if (fabsf(n.z) > 0.70710678118654752f) {
                    const float a = n.y*n.y + n.z*n.z, kk = 1.f / sqrtf(fmaxf(a, 1e-30f));
                    t1 = v3(0.f, -n.z * kk, n.y * kk); t2 = v3(a * kk, -n.x * t1.z, n.x * t1.y);
                } else {
                    const float a = n.x*n.x + n.y*n.y, kk = 1.f / sqrtf(fmaxf(a, 1e-30f));
                    t1 = v3(-n.y * kk, n.x * kk, 0.f); t2 = v3(-n.z * t1.y, n.z * t1.x, a * kk);
                }
                ro->rB[c] = sub(cc.pB, op);
                PBRE_UNROLL for (int d = 0; d < 3; d++) {
                    const V3 dir = d == 0 ? n : (d == 1 ? t1 : t2);
                    ro->dir[c][d] = ro->act[c] ? dir : v3(0.f, 0.f, 0.f);
                    PBRE_UNROLL for (int j = 0; j < ND; j++) {
                        bool onchain = false;
                        PBRE_UNROLL for (int e = 0; e < ND; e++) if (Topo::is_anc(j, e) && cc.owner == e) onchain = true;
                        sink.ro_j(c, d, j, (ro->act[c] && onchain) ? dot(dir, add(Sl[j], cross(Sa[j], cc.pA))) : 0.f);
                    }
                }
            }
        }
    }

    // ------------------------------------------------------------------------------------------------ the step
    // st: the env's Q[W] | V[W] | X[16] record.  mi: this lane's slice of the wave's LDS region (MLDS floats, stride MS).
    // Returns the class of the state it produced (0 simple, 1 complex).
    static PBRE_HD int step(const Tab& T, const Params& P, float* st, const float* act, float* out, int mode, int flags,
                            unsigned long long env_id, const float* tgt, float* mi) {
        const bool obj_on = !(flags & 1);
        float q[ND], qd[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = st[j]; qd[j] = st[W + j]; }
        // action_repeat > 1: the env left the apply_action loop in an earlier iteration of this env.step() (X[14]): it does not simulate
        // (its lane runs along and stores nothing), only the evaluation of the state it is in
        const bool skip = st[XO + 14] != 0.f;
        const float dt = P.dt, inv_dt = P.inv_dt, vmax = P.vmax;
        // NaN / Inf guard (see Fast::step_t): 0 while the incoming state is finite, NaN otherwise; added to one position of the new state
        float fin_in = 0.f;
        PBRE_UNROLL for (int j = 0; j < ND; j++) { fin_in = fmaf(q[j], 0.f, fin_in); fin_in = fmaf(qd[j], 0.f, fin_in); }
        if (obj_on) { PBRE_UNROLL for (int k = 0; k < 7; k++) fin_in = fmaf(st[LC + k], 0.f, fin_in); PBRE_UNROLL for (int k = 0; k < 6; k++) fin_in = fmaf(st[W + LC + k], 0.f, fin_in); }

        // ---- kinematics + dynamics -> bias torques tau and the joint-space inertia M (into the M^-1 store)
        Mat Mi; Mi.lds = mi;
        float tau[ND];
        RtC rt;
        dynamics(T, P, q, qd, Mi, tau, rt);
        if (P.jd_dt != 0.f) { PBRE_UNROLL for (int j = 0; j < ND; j++) Mi.set(sym(j, j), fmaf(P.jd_dt, T.jdamp[j], Mi.get(sym(j, j)))); }   // implicit joint damping: M + dt C
        // ---- M^-1 by the symmetric sweep operator (A -> -A^-1), Gauss-Jordan arithmetic on the triangle, in place
        PBRE_UNROLL for (int k = 0; k < ND; k++) {
            float b[ND];
            PBRE_UNROLL for (int i = 0; i < ND; i++) b[i] = Mi.get(sym(i, k));
            const float pv = 1.f / b[k];
            PBRE_UNROLL for (int i = 0; i < ND; i++) {
                if (i == k) continue;
                const float bp = b[i] * pv;
                PBRE_UNROLL for (int j = 0; j <= i; j++) { if (j == k) continue; Mi.set(sym(i, j), fmaf(-bp, b[j], Mi.get(sym(i, j)))); }
                Mi.set(sym(i, k), bp);
            }
            Mi.set(sym(k, k), -pv);
        }
        PBRE_UNROLL for (int i = 0; i < NM; i++) Mi.set(i, -Mi.get(i));

        // ---- unconstrained joint velocities w = v*, motor rows against the running velocity (see Fast::step_t), limit rows
        float w[ND], w0[ND], m_dinv[ND], m_rhs[ND], m_app[ND];
        float l_dir[ND], l_rhs[ND], l_app[ND];
        bool lim_any[ND], has_limit = false;
        {
            float acc[ND];
            PBRE_UNROLL for (int j = 0; j < ND; j++) acc[j] = 0.f;
            PBRE_UNROLL for (int j = 0; j < ND; j++)
                PBRE_UNROLL for (int k = 0; k <= j; k++) {
                    const float e = Mi.get(sym(j, k));
                    acc[j] = fmaf(e, tau[k], acc[j]);
                    if (k != j) acc[k] = fmaf(e, tau[j], acc[k]);
                }
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                const float wj = clampf(fmaf(dt, acc[j], qd[j]), -vmax, vmax);
                w[j] = wj; w0[j] = wj;
                float qdes = T.home[j], kp = T.kp_hold[j], kd = T.kd_hold[j];
                if (mode & M_TGT) qdes = tgt[j];                     // Cartesian control: every joint tracks the IK solution with the hold gains
                if (mode & M_ACTION) {                               // joint control (icub_env.py:341-361 through the task env's scaling)
                    kp = T.kp_act[j]; kd = T.kd_act[j];
                    const int ai = T.act_idx[j];
                    if (ai >= 0) qdes = clampf(fmaf(act[ai], P.act_scale, q[j]), T.lower[j], T.upper[j]);
                }
                m_dinv[j] = 1.f / Mi.get(sym(j, j));
                m_rhs[j] = (kp * (qdes - q[j]) * inv_dt + (1.f - kd) * wj) * m_dinv[j];
                m_app[j] = 0.f;
                // joint-limit row (btMultiBodyJointLimitConstraint): exists while the joint is at / beyond the limit
                const float pl = q[j] - T.lower[j], pu = T.upper[j] - q[j];
                const bool lo_v = pl <= 0.f, up_v = !lo_v && pu <= 0.f;
                l_dir[j] = lo_v ? 1.f : (up_v ? -1.f : 0.f);
                const float pen = lo_v ? pl : pu;
                l_rhs[j] = (lo_v || up_v) ? (-pen * P.erp * inv_dt) * m_dinv[j] : 0.f;
                l_app[j] = 0.f;
                lim_any[j] = PBRE_ANY(lo_v || up_v);                 // wave-uniform: the row of a joint nobody has at a limit is skipped
                has_limit = has_limit || lim_any[j];
            }
        }

        // ---- robot-table contact rows (setupMultiBodyContactConstraint, restitution 0): B = M^-1 J^T, 1 / (J B), positional rhs; rows
        //      are evaluated against the running velocity
        float rc_B[NRT][3][ND], rc_dinv[NRT][3], rc_app[NRT][3], rc_rhs[NRT];
        bool rt_on[NRT];
        PBRE_UNROLL for (int c = 0; c < NRT; c++) {
            rt_on[c] = PBRE_ANY(rt.act[c]);                        // wave-uniform; the rows of a lane without the contact are exact no-ops
            PBRE_UNROLL for (int d = 0; d < 3; d++) {
                float denom = 0.f;
                PBRE_UNROLL for (int k = 0; k < ND; k++) {
                    float b = 0.f;
                    if (rt_on[c]) { PBRE_UNROLL for (int j = 0; j < ND; j++) b = fmaf(Mi.get(sym(k, j)), Mi.rtJ[c][d][j], b); }
                    rc_B[c][d][k] = b; denom = fmaf(Mi.rtJ[c][d][k], b, denom);
                }
                rc_dinv[c][d] = rt.act[c] ? 1.f / denom : 0.f;
                rc_app[c][d] = 0.f;
            }
            const float pen = rt.dist[c] + P.slop;
            rc_rhs[c] = rt.act[c] ? (pen > 0.f ? -pen * inv_dt : -pen * P.erp * inv_dt) * rc_dinv[c][0] : 0.f;
        }
        auto rrow = [&](int c, int d) {
            float jv = 0.f;
            PBRE_UNROLL for (int j = 0; j < ND; j++) jv = fmaf(Mi.rtJ[c][d][j], w[j], jv);
            float sn;
            if (d == 0) sn = med3(rc_app[c][0] - fmaf(jv, rc_dinv[c][0], -rc_rhs[c]), 0.f, 1e10f);
            else {
                const float hi = rt.mu[c] * rc_app[c][0];
                sn = med3(rc_app[c][d] - jv * rc_dinv[c][d], -hi, hi);
                sn = hi > 0.f ? sn : rc_app[c][d];
            }
            const float dd = sn - rc_app[c][d]; rc_app[c][d] = sn;
            PBRE_UNROLL for (int k = 0; k < ND; k++) w[k] = fmaf(dd, rc_B[c][d][k], w[k]);
        };
        auto contacts = [&]() {      // Bullet: normals, then frictions (the object's own rows are ObjStep's: no unknown shared with these)
            PBRE_UNROLL for (int c = 0; c < NRT; c++) if (rt_on[c]) rrow(c, 0);
            PBRE_UNROLL for (int c = 0; c < NRT; c++) if (rt_on[c]) { rrow(c, 1); rrow(c, 2); }
        };

        // ---- the object's half of the step: rows against the table only (this class has no robot-object contact)
        ObjStep ob;
        float pose[7], tw0[6];
        if (obj_on) {
            PBRE_UNROLL for (int k = 0; k < 7; k++) pose[k] = st[LC + k];
            PBRE_UNROLL for (int k = 0; k < 6; k++) tw0[k] = st[W + LC + k];
            ob.setup(P, pose, tw0, P.obj_m, P.obj_mu, P.kl);
        }

        // ---- projected Gauss-Seidel, Bullet order: motors and limits reversed on even sweeps, forward on odd ones; contacts
        const float mlim = P.motor_imp, llim = P.limit_imp;
        // Column j of M^-1 is fetched from LDS one row ahead of its use (`cn`, loaded before the arithmetic of the current row is
        // written down, so the reads are in flight while the current row's 20 FMAs issue).
        float cc[ND], cn[ND];
        auto fetch = [&](int j, float* c) {
            int o = 0;
            PBRE_OPAQUE_I(o);
            PBRE_UNROLL for (int k = 0; k < ND; k++) c[k] = Mi.geto(sym(k, j), o);
        };
        auto axpy = [&](const float* c, float d) { PBRE_UNROLL for (int k = 0; k < ND; k++) w[k] = fmaf(d, c[k], w[k]); };
        auto motor = [&](int j, const float* c) {        // delta form, see Fast::step_t
            const float nt = fmaf(-m_dinv[j], w[j], m_rhs[j]);
            const float d = med3(nt, -mlim - m_app[j], mlim - m_app[j]);
            m_app[j] += d;
            axpy(c, d);
        };
        float m_peak = 0.f;              // clamp-free rows: the largest |applied impulse| any motor ever had
        auto motor_free = [&](int j, const float* c) {
            const float d = fmaf(-m_dinv[j], w[j], m_rhs[j]);
            m_app[j] += d;
            m_peak = fmaxf(m_peak, fabsf(m_app[j]));
            axpy(c, d);
        };
        auto limit = [&](int j) {
            float c[ND];
            fetch(j, c);
            const float t = fmaf(m_dinv[j] * l_dir[j], w[j], -l_rhs[j]);
            const float s = med3(l_app[j] - t, 0.f, llim);
            const float d = (s - l_app[j]) * l_dir[j]; l_app[j] = s;
            axpy(c, d);
        };
        auto motors = [&](auto&& mrow, bool rev) {          // (two buffers used alternately: after unrolling every index is static, no copies)
            fetch(rev ? ND - 1 : 0, cc);
            PBRE_UNROLL for (int t = 0; t < ND; t++) {
                const int j = rev ? ND - 1 - t : t;
                float* cur = (t & 1) ? cn : cc;
                float* nxt = (t & 1) ? cc : cn;
                if (t + 1 < ND) fetch(rev ? j - 1 : j + 1, nxt);
                mrow(j, cur);
            }
        };
        auto solve = [&](auto&& mrow) {
            for (int it = 0; it < P.iters; it += 2) {
                motors(mrow, true);
                if (has_limit) { PBRE_UNROLL for (int j = ND - 1; j >= 0; j--) if (lim_any[j]) limit(j); }
                contacts();
                if (obj_on) ob.sweep();
                if (it + 1 >= P.iters) break;
                if (has_limit) { PBRE_UNROLL for (int j = 0; j < ND; j++) if (lim_any[j]) limit(j); }
                motors(mrow, false);
                contacts();
                if (obj_on) ob.sweep();
            }
        };
        // Clamp-free motor rows first (no motor comes near PyBullet's default force bound; the largest |applied impulse| is tracked off
        // the row-to-row chain); one test after the loop decides, a wave in which it fails starts over with the clamping rows (same
        // deltas bit for bit wherever a clamp does not bind).
        PBRE_REG_BARRIER();
        // everything the loop reads starts a fresh live range here
        PBRE_UNROLL for (int k = 0; k < (MREG > 0 ? MREG : 1); k++) PBRE_OPAQUE_F(Mi.reg[k]);
        PBRE_UNROLL for (int j = 0; j < ND; j++) { PBRE_OPAQUE_F(m_dinv[j]); PBRE_OPAQUE_F(m_rhs[j]); PBRE_OPAQUE_F(w[j]); }
        if (obj_on) {
            PBRE_UNROLL for (int c = 0; c < ObjStep::NK; c++) {
                PBRE_OPAQUE_F(ob.c_rx[c]); PBRE_OPAQUE_F(ob.c_ry[c]); PBRE_OPAQUE_F(ob.c_rz[c]); PBRE_OPAQUE_F(ob.r_rhs[c]);
                PBRE_UNROLL for (int d = 0; d < 3; d++) { PBRE_OPAQUE_F(ob.r_dinv[c][d]); PBRE_UNROLL for (int e = 0; e < 3; e++) PBRE_OPAQUE_F(ob.g[c][d][e]); }
            }
        }
        solve(motor_free);
        {
            const bool over = !(m_peak <= mlim);      // (a NaN fails the test as well)
#ifdef PBRE_LANE_NOFALLBACK
            if (false) {
#else
            if (PBRE_ANY(over)) {
#endif
                PBRE_UNROLL for (int j = 0; j < ND; j++) { w[j] = w0[j]; m_app[j] = 0.f; l_app[j] = 0.f; }
                PBRE_UNROLL for (int c = 0; c < NRT; c++) PBRE_UNROLL for (int d = 0; d < 3; d++) rc_app[c][d] = 0.f;
                if (obj_on) ob.setup(P, pose, tw0, P.obj_m, P.obj_mu, P.kl);
                solve(motor);
            }
        }

        // ---- integrate (semi-implicit Euler; quaternion exponential map for the object)
        PBRE_REG_BARRIER();
        PBRE_UNROLL for (int j = 0; j < ND; j++) {
            const float v = clampf(w[j], -vmax, vmax);
            qd[j] = skip ? st[W + j] : v; q[j] = skip ? st[j] : fmaf(dt, v, st[j]);
            if (j == 0) q[j] += fin_in;
            st[j] = q[j]; st[W + j] = qd[j];
        }
        V3 op = v3(st[LC], st[LC + 1], st[LC + 2]);
        Q4 oq; oq.x = st[LC + 3]; oq.y = st[LC + 4]; oq.z = st[LC + 5]; oq.w = st[LC + 6];
        if (obj_on && !skip) {
            float o[6];
            ob.result(P, o);
            const V3 ov = v3(o[0], o[1], o[2]), ow = v3(o[3], o[4], o[5]);
            op = v3(fmaf(dt, ov.x, op.x), fmaf(dt, ov.y, op.y), fmaf(dt, ov.z, op.z));
            float ang = norm(ow);
            if (ang * dt > 0.78539816339744831f) ang = 0.78539816339744831f * inv_dt;
            float sh_, ch_;
            FX::sincos_(0.5f * ang * dt, sh_, ch_);
            const float sc_ = ang < 0.001f ? 0.5f * dt - dt * dt * dt * 0.020833333333f * ang * ang : sh_ / ang;
            Q4 dq; dq.x = ow.x * sc_; dq.y = ow.y * sc_; dq.z = ow.z * sc_; dq.w = ch_;
            const Q4 nq = FX::qmul(dq, oq);
            const float in = 1.f / sqrtf(nq.x*nq.x + nq.y*nq.y + nq.z*nq.z + nq.w*nq.w);
            oq.x = nq.x * in; oq.y = nq.y * in; oq.z = nq.z * in; oq.w = nq.w * in;
            st[LC] = op.x; st[LC + 1] = op.y; st[LC + 2] = op.z; st[LC + 3] = oq.x; st[LC + 4] = oq.y; st[LC + 5] = oq.z; st[LC + 6] = oq.w;
            PBRE_UNROLL for (int k = 0; k < 6; k++) st[W + LC + k] = o[k];
        }
        const Tab* T2 = &T;
        PBRE_LAUNDER(T2);
        return finish(*T2, P, st, q, qd, op, oq, out, mode, flags, env_id);
    }

    // ------------------------------------------------------------------------------------------------ class + end effector
    struct Tail { int cls; M3 Re; V3 pe, Va, Vl; };
    // One streaming sweep over the links: class of the state (any robot collision sphere within the contact margin of the object
    // or the table -> 1) and, when qd is given, the frame and spatial velocity of the end effector's owner link.
    static PBRE_HD Tail sweep(const Tab& T, const Params& P, const float* q, const float* qd, V3 op, Q4 oq, int flags) {
        const bool obj_on = !(flags & 1);
        Tail t;
        const M3 Ro = FX::quat_R(oq);
        const V3 oh = v3(P.obj_h[0], P.obj_h[1], P.obj_h[2]);
        const V3 tc = v3(P.tab_c[0], P.tab_c[1], P.tab_c[2]), th = v3(P.tab_h[0], P.tab_h[1], P.tab_h[2]);
        M3 Id; PBRE_UNROLL for (int k = 0; k < 9; k++) Id.m[k] = (k % 4 == 0) ? 1.f : 0.f;
        const float orad = sqrtf(dot(oh, oh)), ztop = tc.z + th.z;
        int nO = 0;
        const int eo = T.ee_owner;
        t.Va = v3(0.f, 0.f, 0.f); t.Vl = v3(0.f, 0.f, 0.f); t.pe = v3(0.f, 0.f, 0.f);
        PBRE_UNROLL for (int k = 0; k < 9; k++) t.Re.m[k] = 0.f;
        M3 R[ND]; V3 p[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) {
            const int pj = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
            M3 Rl; V3 pl, ax;
            joint_xf(T, j, q[j], Rl, pl, ax);
            if (Topo::parent(j) < 0) { R[j] = Rl; p[j] = pl; } else { R[j] = mm(R[pj], Rl); p[j] = add(p[pj], mv(R[pj], pl)); }
            for (int s = 0; s < T.nspheres; s++) {
                if (T.s_owner[s] != j) continue;
                const V3 sc = add(p[j], mv(R[j], v3(T.s_c[0][s], T.s_c[1][s], T.s_c[2][s])));
                // cheap wave-wide lower bounds first (bounding sphere of the object; height above the table top)
                const float sr = T.s_r[s];
                if (obj_on) {
                    const V3 dd = sub(sc, op);
                    const float reach = sr + P.margin + orad;
                    if (PBRE_ANY(!(dot(dd, dd) >= reach * reach)) && FX::sphere_obj_dist(P, sc, sr, op, Ro, oh) < P.margin) nO++;
                }
            }
            if (qd) {
                bool anc = false;      // is j an ancestor-or-self of the EE owner?  (compile-time tree, uniform runtime owner)
                PBRE_UNROLL for (int e = 0; e < ND; e++) if (Topo::is_anc(j, e) && eo == e) anc = true;
                if (anc) {
                    const V3 aw = mv(R[j], ax);
                    if (Topo::jtype(j) == 1) { t.Va = add(t.Va, scl(aw, qd[j])); t.Vl = add(t.Vl, scl(cross(p[j], aw), qd[j])); }
                    else t.Vl = add(t.Vl, scl(aw, qd[j]));
                }
                if (eo == j) { t.Re = R[j]; t.pe = p[j]; }
            }
        }
        t.cls = nO != 0 ? 1 : 0;      // complex: a robot-object contact (robot-table contacts are rows of the lane-per-env pipeline)
        return t;
    }
    static PBRE_HD int classify_state(const Tab& T, const Params& P, const float* st, int flags) {
        float q[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) q[j] = st[j];
        Q4 oq; oq.x = st[LC + 3]; oq.y = st[LC + 4]; oq.z = st[LC + 5]; oq.w = st[LC + 6];
        return sweep(T, P, q, nullptr, v3(st[LC], st[LC + 1], st[LC + 2]), oq, flags).cls;
    }

    // ------------------------------------------------------------------------------------------------ observation / task
    // Observation, reward, termination of the state (q, qd, op, oq; already stored in st) and its class.  Same logic as Core::observe
    // (iCub branches: icub_reach_gym_env.py:262-330, icub_push_gym_env.py:284-373, icub_push_gym_goal_env.py:89-139); with
    // PBRE_F_AUTO_RESET a finished env restarts from the settled snapshot right here.
    static PBRE_HD int finish(const Tab& T, const Params& P, float* st, float* q, float* qd, V3 op, Q4 oq,
                              float* out, int mode, int flags, unsigned long long env_id) {
        const bool want_obs = (mode & (M_OBS | M_TASK)) != 0;
        float reward = 0.f, done = 0.f;
        V3 ee, eul, vee;
        int cls;
        // The kinematic sweep is needed for the state the step produced and, when some env of the wave finishes its episode under
        // PBRE_F_AUTO_RESET, again for the first state of the next one: a two-pass loop around ONE copy of the code.
        float* X = st + XO;
        V3 tg = v3(X[0], X[1], X[2]);
        V3 ee0 = v3(0.f, 0.f, 0.f), eul0 = ee0, vee0 = ee0, op0 = op, tg0 = tg; Q4 oq0 = oq; int cls0 = 0;
        bool again = false, bad = false;
        PBRE_NOUNROLL for (int pass = 0; pass < 2; pass++) {
            const Tail tl = sweep(T, P, q, want_obs ? qd : nullptr, op, oq, flags);
            cls = tl.cls;
            if (pass == 0) {
                // NaN / Inf guard (see Fast::finish): a non-finite position of the new state is counted, returned as reward 0 / done 1 and
                // restarted with PBRE_F_AUTO_RESET
                float fin = 0.f;
                PBRE_UNROLL for (int j = 0; j < ND; j++) fin = fmaf(q[j], 0.f, fin);
                if (!(flags & 1)) fin = fmaf(op.x, 0.f, fmaf(op.y, 0.f, fmaf(op.z, 0.f, fmaf(oq.x, 0.f, fmaf(oq.y, 0.f, fmaf(oq.z, 0.f, fmaf(oq.w, 0.f, fin)))))));
                bad = !(fin == 0.f);
                if (PBRE_ANY(bad)) { if (bad && P.bad_count) PBRE_COUNT_BAD(P.bad_count); }
            }
            if (!want_obs) return cls;
            {
                M3 Eo; PBRE_UNROLL for (int k = 0; k < 9; k++) Eo.m[k] = T.ee_R[k];
                const M3 Ree = mm(tl.Re, Eo);
                ee = add(tl.pe, mv(tl.Re, v3(T.ee_p[0], T.ee_p[1], T.ee_p[2])));
                vee = add(tl.Vl, cross(tl.Va, ee));
                eul = FX::quat_euler(FX::R_quat(Ree));
            }
            if (pass == 1) {
                if (again) {
                    if (P.robot >= 1 && P.task >= 1) { X[12] = norm(sub(ee, op)); X[13] = norm(sub(op, tg)); }     // icub_push_gym_env.py:124-127
                } else { ee = ee0; eul = eul0; vee = vee0; op = op0; oq = oq0; tg = tg0; cls = cls0; }
                break;
            }
            if (mode & M_TASK) {
                const float d1 = norm(sub(ee, op)), d2 = norm(sub(op, tg));
                const float dsucc = P.task >= 1 ? d2 : d1;
                const bool succ = dsucc <= P.dist_min;
                float cnt = X[3], term = X[4];
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
                    float base = P.task == 1 ? -d1 - d2 : -d1;
                    if (P.robot >= 1) {
                        // iCub (icub_reach_gym_env.py:318-330: the bonus is added; icub_push_gym_env.py:346-373: reward types 0 / 1)
                        if (P.task == 0) reward = base + (succ ? 1000.f + (100.f - d1 * 80.f) : 0.f);
                        else {
                            if (P.reward_type != 0) {
                                const float r1 = 0.125f * (1.f - d1 / X[12]);
                                const float r2 = 0.25f * (1.f - d2 / X[13]);
                                base = r1 + (d1 > 0.1f ? 0.f : r2);
                            }
                            reward = base + (succ ? 1000.f : 0.f);
                        }
                    } else
                        reward = succ ? 1000.f + (100.f - dsucc * 80.f) : base;
                }
                if (bad) { reward = 0.f; done = 1.f; }
                X[3] = cnt; X[4] = term;
                X[14] = ((mode & M_INNER) && left) ? 1.f : 0.f;
                again = (flags & 2) && !(mode & M_INNER) && done != 0.f;
            }
            if (!PBRE_ANY(again)) break;
            if (P.rst_ok) {
                // The settled robot pose is the same in every env, so its end-effector pose was recorded with the snapshot (P.rst_ee) and the
                // robot is at rest: the first observation of the next episode needs no second kinematic sweep -- which the whole wave would
                // otherwise run whenever one of its 64 envs finishes (every step of a 32768-env batch: kw_fin 0.10 instead of 0.05 ms).
                // As in Fast::finish; P.rst_ok = 0 (no snapshot yet, or its pose is not a simple-class state) takes the second pass below.
                if (again) {
                    snapshot_reset(T, P, env_id, st);
                    PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = st[j]; qd[j] = 0.f; }
                    op = v3(st[LC], st[LC + 1], st[LC + 2]);
                    oq.x = st[LC + 3]; oq.y = st[LC + 4]; oq.z = st[LC + 5]; oq.w = st[LC + 6];
                    tg = v3(X[0], X[1], X[2]);
                    ee = v3(P.rst_ee[0], P.rst_ee[1], P.rst_ee[2]); eul = v3(P.rst_ee[3], P.rst_ee[4], P.rst_ee[5]); vee = v3(0.f, 0.f, 0.f);
                    cls = 0;
                    if (P.robot >= 1 && P.task >= 1) { X[12] = norm(sub(ee, op)); X[13] = norm(sub(op, tg)); }     // icub_push_gym_env.py:124-127
                }
                break;
            }
            // snapshot reset of the finished envs (settled robot pose and object height of the last full reset, freshly sampled object
            // pose and target), then the first observation of the new episode in the second pass
            ee0 = ee; eul0 = eul; vee0 = vee; op0 = op; tg0 = tg; oq0 = oq; cls0 = cls;
            if (again) {
                snapshot_reset(T, P, env_id, st);
                PBRE_UNROLL for (int j = 0; j < ND; j++) { q[j] = st[j]; qd[j] = 0.f; }
                op = v3(st[LC], st[LC + 1], st[LC + 2]);
                oq.x = st[LC + 3]; oq.y = st[LC + 4]; oq.z = st[LC + 5]; oq.w = st[LC + 6];
                tg = v3(X[0], X[1], X[2]);
            }
        }
        if (out) {
            const V3 oe = FX::quat_euler(oq);
            const Q4 qh = FX::euler_quat(eul), qo = FX::euler_quat(oe);
            const V3 rel = mtv(FX::quat_R(qh), sub(op, ee));
            Q4 qhi; qhi.x = -qh.x; qhi.y = -qh.y; qhi.z = -qh.z; qhi.w = qh.w;
            const V3 er = FX::quat_euler(FX::qmul(qhi, qo));
            // Panda: normalised EE velocity (panda_env.py:174-178); iCub: raw (icub_env.py:233-236)
            const V3 vn = P.robot >= 1 ? vee : v3(vee.x / 0.04f, (vee.y - 0.01f) / 0.07f, vee.z / 0.03f);
            out[0] = ee.x; out[1] = ee.y; out[2] = ee.z; out[3] = eul.x; out[4] = eul.y; out[5] = eul.z;
            out[6] = vn.x; out[7] = vn.y; out[8] = vn.z;
            PBRE_UNROLL for (int j = 0; j < ND; j++) { const int oi = T.obs_idx[j]; if (oi >= 0) out[9 + oi] = q[j]; }
            float* o2 = out + 9 + T.n_obs_j;
            o2[0] = op.x; o2[1] = op.y; o2[2] = op.z; o2[3] = oe.x; o2[4] = oe.y; o2[5] = oe.z;
            o2[6] = rel.x; o2[7] = rel.y; o2[8] = rel.z; o2[9] = er.x; o2[10] = er.y; o2[11] = er.z;
            int o = 12;
            if (P.task >= 1) { o2[12] = tg.x; o2[13] = tg.y; o2[14] = tg.z; o = 15; }
            o2[o] = reward; o2[o + 1] = done;
        }
        return cls;
    }

    // ------------------------------------------------------------------------------------------------ reset helpers (scalar, as Core's)
    static PBRE_HD float clamps(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
    static PBRE_HD void snapshot_reset(const Tab& T, const Params& P, unsigned long long env_id, float* st) {
        const unsigned ep = (unsigned)(int)st[XO + 5] + 1u;
        float* ob = st + LC;
        float* X = st + XO;
        for (int k = 0; k < S::STATE; k++) st[k] = 0.f;
        for (int k = 0; k < ND; k++) st[k] = T.rst_q[k];                // settled robot pose
        const float x_min = P.ws[0][0] + 0.05f, x_max = P.ws[0][1] - 0.1f;
        const float y_min = P.ws[1][0] + 0.05f, y_max = P.ws[1][1] - 0.05f;
        float px = x_min + 0.5f * (x_max - x_min), py = y_min + 0.5f * (y_max - y_min);
        float yaw = 0.78539816339744831f;
        unsigned r[4];
        if (P.obj_std > 0.f) {
            FX::philox((unsigned)env_id, (unsigned)(env_id >> 32), ep, 0u, P.seed_lo, P.seed_hi, r);
            px += -P.obj_std + 2.f * P.obj_std * FX::u01(r[0]);
            py += -P.obj_std + 2.f * P.obj_std * FX::u01(r[1]);
            yaw = -0.78539816339744831f + 1.57079632679489662f * FX::u01(r[2]);
        }
        ob[0] = clamps(px, x_min, x_max); ob[1] = clamps(py, y_min, y_max); ob[2] = P.rst_objz;      // settled object height
        ob[3] = 0.f; ob[4] = 0.f; ob[5] = sinf(0.5f * yaw); ob[6] = cosf(0.5f * yaw);
        X[5] = (float)(int)ep;
        if (P.use_ik) for (int k = 0; k < 6; k++) X[6 + k] = P.home_hand[k];
        if (P.task >= 1) {                                               // sample_tg_pose on the settled object position
            const float tx_min = P.ws[0][0] + 0.07f, tx_max = P.ws[0][1] - 0.07f;
            float tx = ob[0] + 0.05f, ty = ob[1] + 0.05f;
            if (P.tg_std > 0.f) {
                FX::philox((unsigned)env_id, (unsigned)(env_id >> 32), ep, 1u, P.seed_lo, P.seed_hi, r);
                const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = FX::u01(r[1]);
                const float rad = sqrtf(-2.f * logf(u1)) * P.tg_std;
                tx = ob[0] + rad * cosf(6.28318530717958648f * u2);
                ty = ob[1] + rad * sinf(6.28318530717958648f * u2);
            }
            X[0] = clamps(tx, tx_min, tx_max); X[1] = clamps(ty, P.ws[1][0], P.ws[1][1]); X[2] = ob[2];
        }
    }

    // ------------------------------------------------------------------------------------------------ Cartesian control
    // apply_action, IK branch (icub_reach_gym_env.py:204-230 + icub_env.py:262-330): accumulate the scaled action on the commanded
    // hand pose (X[6..11]), clip rotation and workspace, damped-least-squares IK from the current joint angles over the joints of
    // the chain to the end effector; same algorithm and stopping rule as Core::ik_targets / oracle orc_ik.  Writes tgt[0..ND) and
    // X[6..11].  (The reset-time targets of the home hand pose are the lane-group kernel's.)
    // done / seq (device pipeline, pbre_lane.hip; null on the host): the env's hand-over box.  A lane writes its targets -- each with the
    // launch's sequence number beside it -- in the iteration in which ITS env converges, not when the slowest env of its wave leaves the
    // loop: the solve kernels wait per env (quad_step), so the 0.09 % of the envs that iterate to the cap no longer hold up everybody else.
    static PBRE_HD void ik_targets(const Tab& T, const Params& P, float* st, const float* act, float* tgt, int* done = nullptr, int seq = 0) {
        if (T.ee_owner == Topo::ee0) ik_targets_t<Topo::ee0>(T, P, st, act, tgt, done, seq); else ik_targets_t<Topo::ee1>(T, P, st, act, tgt, done, seq);
    }
    // EO: the lane that owns the end effector; the chain base -> EO is static (T.on_chain agrees with it, lane_topo_matches)
    template <int EO>
    static PBRE_HD void ik_targets_t(const Tab& T, const Params& P, float* st, const float* act, float* tgt, int* done = nullptr, int seq = 0) {
        float* X = st + XO;
        V3 pos = v3(fmaf(act[0], P.ik_ps, X[6]), fmaf(act[1], P.ik_ps, X[7]), fmaf(act[2], P.ik_ps, X[8]));
        V3 eul = v3(X[9], X[10], X[11]);
        if (P.ctrl_ori) {
            eul.x = clampf(fmaf(act[3], P.ik_rs, eul.x), P.eu_lim[0][0], P.eu_lim[0][1]);
            eul.y = clampf(fmaf(act[4], P.ik_rs, eul.y), P.eu_lim[1][0], P.eu_lim[1][1]);
            eul.z = clampf(fmaf(act[5], P.ik_rs, eul.z), P.eu_lim[2][0], P.eu_lim[2][1]);
        }
        pos.x = clampf(pos.x, P.rws[0][0], P.rws[0][1]); pos.y = clampf(pos.y, P.rws[1][0], P.rws[1][1]); pos.z = clampf(pos.z, P.rws[2][0], P.rws[2][1]);
        if (X[14] == 0.f) { X[6] = pos.x; X[7] = pos.y; X[8] = pos.z; X[9] = eul.x; X[10] = eul.y; X[11] = eul.z; }   // (an env that left the apply_action loop keeps its pose)
        const M3 Rt = FX::quat_R(FX::euler_quat(eul));
        const V3 tp = add(pos, mv(Rt, v3(P.ik_off[0], P.ik_off[1], P.ik_off[2])));
        constexpr int eo = EO;
        M3 Eo; PBRE_UNROLL for (int k = 0; k < 9; k++) Eo.m[k] = T.ee_R[k];
        float q[ND];
        PBRE_UNROLL for (int j = 0; j < ND; j++) q[j] = st[j];
        // joints off the chain: the iCub sends those it does not control to their rest pose (icub_env.py:316-317), the others keep
        // their current angle
        bool published = false;
        auto publish = [&]() {
            PBRE_UNROLL for (int j = 0; j < ND; j++) PBRE_IK_STORE(done, seq, j, tgt + j, Topo::is_anc(j, EO) ? q[j] : (T.blocked[j] ? T.home[j] : st[j]));
            published = true;
        };
#ifdef PBRE_IK_PROBE      // host emulation only (tools/ik_cycle_probe.py): at which iteration does this env's iteration become periodic?
        float q_m2[ND]; int probe_at = -1, probe_kind = 3;      // kind 0 converged (residual), 1 fixed point, 2 two-cycle, 3 neither within the cap
        PBRE_UNROLL for (int j = 0; j < ND; j++) q_m2[j] = 3.0e38f;
#endif
        for (int it = 0; it < P.ik_iters; it++) {
#ifdef PBRE_IK_PROBE
            float q_m1[ND];
            PBRE_UNROLL for (int j = 0; j < ND; j++) q_m1[j] = q[j];
#endif
            // FK of the chain links only (a link off the chain has no chain link below it)
            M3 R[ND]; V3 p[ND], aw[ND];
            M3 Re = Eo; V3 po = v3(0.f, 0.f, 0.f);
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                if (!Topo::is_anc(j, EO)) continue;
                const int pj = Topo::parent(j) < 0 ? 0 : Topo::parent(j);
                M3 Rl; V3 pl, ax;
                joint_xf(T, j, q[j], Rl, pl, ax);
                if (Topo::parent(j) < 0) { R[j] = Rl; p[j] = pl; } else { R[j] = mm(R[pj], Rl); p[j] = add(p[pj], mv(R[pj], pl)); }
                aw[j] = mv(R[j], ax);
                if (eo == j) { Re = R[j]; po = p[j]; }
            }
            const V3 pe = add(po, mv(Re, v3(T.ee_lp[0], T.ee_lp[1], T.ee_lp[2])));
            float e[6];
            e[0] = tp.x - pe.x; e[1] = tp.y - pe.y; e[2] = tp.z - pe.z;
            const bool go = sqrtf(fmaf(e[0], e[0], fmaf(e[1], e[1], e[2] * e[2]))) >= P.ik_res;
#ifdef PBRE_IK_PROBE
            if (!go && probe_at < 0) { probe_at = it; probe_kind = 0; }
#endif
            if (done && !go && !published) publish();      // (a converged env's angles no longer change: these are its final targets)
            if (!PBRE_ANY(go)) break;
            {   // orientation error as a world-frame rotation vector: axis-angle of Rt (Re Eo)^T
                const M3 Ree = mm(Re, Eo);
                M3 Rr;
                PBRE_UNROLL for (int a = 0; a < 3; a++)
                    PBRE_UNROLL for (int b = 0; b < 3; b++)
                        Rr.m[a*3+b] = fmaf(Rt.m[a*3], Ree.m[b*3], fmaf(Rt.m[a*3+1], Ree.m[b*3+1], Rt.m[a*3+2] * Ree.m[b*3+2]));
                const float sx = Rr.m[7] - Rr.m[5], sy = Rr.m[2] - Rr.m[6], sz = Rr.m[3] - Rr.m[1];
                const float s2 = sqrtf(fmaf(sx, sx, fmaf(sy, sy, sz * sz))) * 0.5f;
                const float c2 = (Rr.m[0] + Rr.m[4] + Rr.m[8] - 1.f) * 0.5f;
                const float ang = atan2f(s2, c2);
                const float f = s2 > 1e-9f ? ang / (2.f * fmaxf(s2, 1e-30f)) : 0.5f;
                e[3] = f * sx; e[4] = f * sy; e[5] = f * sz;
            }
            // A = J J^T + lambda^2 I over the chain joints (column j: [S_l + S_a x pe ; S_a]), Cholesky, y = A^-1 e, dq = J^T y
            float A[6][6];
            PBRE_UNROLL for (int a = 0; a < 6; a++) PBRE_UNROLL for (int b = 0; b <= a; b++) A[a][b] = a == b ? P.ik_l2 : 0.f;
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                if (!Topo::is_anc(j, EO)) continue;
                const V3 jl = Topo::jtype(j) == 1 ? cross(aw[j], sub(pe, p[j])) : aw[j];
                const V3 ja = Topo::jtype(j) == 1 ? aw[j] : v3(0.f, 0.f, 0.f);
                const float J[6] = {jl.x, jl.y, jl.z, ja.x, ja.y, ja.z};
                PBRE_UNROLL for (int a = 0; a < 6; a++) PBRE_UNROLL for (int b = 0; b <= a; b++) A[a][b] = fmaf(J[a], J[b], A[a][b]);
            }
            // (every division by a diagonal entry of the factor is a multiplication by its reciprocal, taken once: 6 reciprocals instead
            // of 27 divisions on this kernel's critical chain -- a lone wave per SIMD iterating up to 100 times)
            bool moved = false;
            float y[6], inv[6];
            PBRE_UNROLL for (int a = 0; a < 6; a++)
                PBRE_UNROLL for (int b = 0; b <= a; b++) {
                    float sum = A[a][b];
                    PBRE_UNROLL for (int k = 0; k < b; k++) sum = fmaf(-A[a][k], A[b][k], sum);
                    if (a == b) { A[a][a] = sqrtf(sum); inv[a] = 1.f / A[a][a]; } else A[a][b] = sum * inv[b];
                }
            PBRE_UNROLL for (int a = 0; a < 6; a++) { float sum = e[a]; PBRE_UNROLL for (int k = 0; k < a; k++) sum = fmaf(-A[a][k], y[k], sum); y[a] = sum * inv[a]; }
            PBRE_UNROLL for (int a = 5; a >= 0; a--) { float sum = y[a]; PBRE_UNROLL for (int k = a + 1; k < 6; k++) sum = fmaf(-A[k][a], y[k], sum); y[a] = sum * inv[a]; }
            PBRE_UNROLL for (int j = 0; j < ND; j++) {
                if (!Topo::is_anc(j, EO)) continue;
                const V3 jl = Topo::jtype(j) == 1 ? cross(aw[j], sub(pe, p[j])) : aw[j];
                const V3 ja = Topo::jtype(j) == 1 ? aw[j] : v3(0.f, 0.f, 0.f);
                const float dq = fmaf(jl.x, y[0], fmaf(jl.y, y[1], fmaf(jl.z, y[2], fmaf(ja.x, y[3], fmaf(ja.y, y[4], ja.z * y[5])))));
                const float qn = go ? q[j] + dq : q[j];
                moved = moved || qn != q[j];
                q[j] = qn;
            }
            // an iteration that changes no joint angle of any env of the wave (targets out of reach: the damped step has shrunk below half
            // an ulp of every angle) would be repeated unchanged until the iteration cap: leaving here gives the same targets bit for bit
#ifdef PBRE_IK_PROBE
            {
                bool two = true;
                PBRE_UNROLL for (int j = 0; j < ND; j++) two = two && q[j] == q_m2[j];
                if (probe_at < 0 && !moved) { probe_at = it; probe_kind = 1; }
                if (probe_at < 0 && two) { probe_at = it; probe_kind = 2; }
                PBRE_UNROLL for (int j = 0; j < ND; j++) q_m2[j] = q_m1[j];
            }
#else
#ifndef PBRE_IK_NO_FIXPOINT_EXIT
            if (!PBRE_ANY(moved)) break;
#endif
#endif
        }
#ifdef PBRE_IK_PROBE
        pbre_ik_probe_record(probe_kind, probe_at);
#endif
        if (!published) publish();
    }
};

}  // namespace pbre
