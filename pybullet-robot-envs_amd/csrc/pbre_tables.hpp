// pbre_tables.hpp -- host side: RobotTable (include/pbre.h) -> per-lane tables of the
// env group (DESIGN.md "Lane layout"): 16 lanes for robots with <= 9 DoF (Panda), a whole
// 64-lane wave for robots with <= 32 DoF (iCub).  Plain C++17, no HIP, so the same code
// feeds the device kernels (pbre_capi.hip) and the host lane emulation used by the CPU
// tests (tests/host_emu).
//
// What PyBullet does at `loadURDF` (reference panda_env.py:53-56) -- build a multibody,
// keep fixed joints as 0-DoF links -- is flattened here: every movable link becomes a
// lane; links behind fixed joints become "sub-bodies" of the nearest movable ancestor
// (mass, COM and inertia are kept separately per sub-body because Bullet's velocity
// damping is applied per link and is not additive).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace pbre {

// Lane layout of one env group.  Lanes 0..NJ-1 own the robot DoF, lanes LC..LC+2 / LC+3..LC+5 the object's linear /
// angular velocity (Q record: LC..LC+2 position, LC+3..LC+6 quaternion), lane L1 is the constant-one lane that carries
// -rhs of a contact row through the row dot product.  State record = Q[W] | V[W] | X[16] floats.
template <int W_, int NJ_, int NSUB_, int NLEV_, int NC_RO_ = 2, int NTIP_ = 0, bool MREC_ = (NTIP_ > 0)>
struct ShapeT {
    static constexpr int W = W_;          // lanes per env
    static constexpr int NJ = NJ_;        // robot DoF lanes
    static constexpr int LC = NJ_;        // first object lane
    static constexpr int L1 = NJ_ + 6;    // constant-one lane
    static constexpr int NSUB = NSUB_;    // rigid sub-bodies per lane
    static constexpr int NLEV = NLEV_;    // pointer-jumping levels (chains up to 2^NLEV deep)
    static constexpr int STATE = 2 * W_ + 16;
    static constexpr int NMW = (NJ_ + 31) / 32;            // 32-bit words of an ancestor / subtree mask
    static constexpr int NC_OT = 4, NC_RO = NC_RO_, NC_RT = 2, NC = NC_OT + NC_RO + NC_RT;   // contact slots: object-table, robot-object, robot-table
    static constexpr int NTIP = NTIP_;    // fingertips whose contact force is reported (iCub hands); such a shape also keeps a
    static constexpr bool MREC = MREC_;       // per-env motor record target[W] | kp[W] | force scale[W] | max velocity[W] (PyBullet's persistent motors)
    static constexpr int TGT = MREC ? 4 * W_ : NJ_;        // floats per env of the motor-target buffer
    static constexpr int TIP0 = NJ_ + 7;                   // Q lanes TIP0..TIP0+NTIP+1: tip forces, tips in contact, other robot-object contacts
    static_assert(NJ_ + 7 + (NTIP_ ? NTIP_ + 2 : 0) <= W_ && NJ_ <= 64, "lane budget");
};
#ifndef PBRE_PANDA_NC_RO
#define PBRE_PANDA_NC_RO 4                // robot-object contact slots of the Panda task envs ("<= 4 cube-robot points")
#endif
using Shape16 = ShapeT<16, 9, 3, 4, PBRE_PANDA_NC_RO>;      // Panda (<= 9 DoF): one env per 16-lane DPP row, 4 envs per wave
using Shape32 = ShapeT<32, 20, 2, 4>;     // iCub as simulated (legs pruned, 20 DoF): one env per half-wave, 2 envs per wave
using Shape64 = ShapeT<64, 32, 2, 4>;     // <= 32 DoF: one env per wave
using Shape128 = ShapeT<128, 60, 2, 4, 6, 5>;   // iCub with hands (legs pruned, 60 DoF): one env per wave, two virtual lanes per physical lane
using ShapePA = ShapeT<32, 9, 3, 4, 4, 5>;      // Panda, robot-level interface (pandaEnv alone: persistent motors, fingertip statistics -- 2 of the
                                                // 5 fingertip slots are used; 4 robot-object contact slots: both spheres of both fingers): one env
                                                // per half-wave, object at Q[9..15], statistics at Q[16..22]
using ShapeIA = ShapeT<32, 20, 2, 4, 2, 0, true>;   // iCub without hands, robot-level interface (iCubEnv alone: persistent motors; the model has no
                                                    // fingertips, so no statistics slots): one env per half-wave like Shape32

// the Panda shape's constants at namespace scope (lane-per-env kernels, C-ABI of the 48-float record)
constexpr int W = Shape16::W, NJ = Shape16::NJ, LC = Shape16::LC, L1 = Shape16::L1, NSUB = Shape16::NSUB, NLEV = Shape16::NLEV;
constexpr int NC_OT = Shape16::NC_OT, NC_RO = Shape16::NC_RO, NC_RT = Shape16::NC_RT, NC = NC_OT + NC_RO + NC_RT;
constexpr int STATE = Shape16::STATE;
constexpr int MAXJ = 64;     // DoF bound of any shape

template <class S>
struct TablesT {
    static constexpr int W = S::W, NJ = S::NJ, NSUB = S::NSUB, NLEV = S::NLEV;
    int   anc[NLEV][W];          // 2^l-th movable ancestor lane, -1 past the root
    static constexpr int NMW = S::NMW;
    int   amask[NMW][W];         // bit i (word i / 32): lane i is an ancestor-or-self of this lane
    int   dmask[NMW][W];         // bit i: lane i is in the subtree of this lane (incl. self)
    int   jtype[W];              // 0 none, 1 revolute, 2 prismatic
    float axis[3][W];
    float R0[9][W], p0[3][W];    // joint frame w.r.t. parent movable link frame at q = 0 (root: world)
    float sb_m[NSUB][W], sb_c[NSUB][3][W], sb_I[NSUB][6][W];   // xx yy zz xy xz yz, link axes, about sub-body COM
    float lower[W], upper[W], jdamp[W], home[W];
    float rst_q[W];              // settled joint positions recorded at the last full reset (snapshot auto-reset); home until then
    float kp_hold[W], kd_hold[W], kp_act[W], kd_act[W];
    int   s_owner[W], s_valid[W];
    float s_c[3][W], s_r[W], s_mu[W];
    int   act_idx[W];            // index of this lane's joint in the action vector (joint control), -1 if not driven
    int   obs_idx[W];            // slot of this lane's joint among the observed joint positions, -1 if not observed
    int   ee_owner;
    float ee_R[9], ee_p[3];      // EE (link COM frame) in the owner lane's link frame
    float ee_lp[3];              // EE link frame origin in the owner lane's link frame (IK target frame)
    int   on_chain[W];           // 1: this lane's joint is on the chain base -> end effector (joints the IK moves)
    int   blocked[W];            // 1: a joint the robot env does not control: the IK branch sends it to its rest pose (icub_env.py:316-317)
    int   tip_of[W];             // fingertip slot (0..NTIP-1) of the link this lane owns, -1 otherwise
    float mforce[W];             // > 0: force bound (N) of this lane's hold motor instead of the default -- the virtual joints of a soft-pinned floating base
                                 //      carry the base constraint's maxForce (link record [35] of a joint with [37] set; PyBullet's createConstraint default: 500 N)
    int   ndof, n_act, n_obs_j, nspheres;
};
using Tables = TablesT<Shape16>;

struct Params {                  // float copies of pbre_physics + task constants used on device
    float dt, inv_dt, gz;
    int   iters;
    float erp, slop, margin, kl, ka, vmax, motor_imp, limit_imp;
    float jd_dt;                  // dt when the joint damping is integrated implicitly (M + dt C), else 0
    float tab_c[3], tab_h[3], tab_mu, ground_z;
    float obj_h[3], obj_m, obj_I[3], obj_mu;
    int   obj_iso;                // the object's principal inertias are equal (a cube): the lane-per-env kernels' in-line object rows apply
    int   obj_shape;              // PBRE_SHAPE_*: 0 box (half extents obj_h), 1 sphere (radius obj_h[0]), 2 cylinder about local z (radius obj_h[0], half height obj_h[2])
    int   task, max_steps, flags;
    float dist_min, act_scale;
    float obj_std, tg_std, ws[3][2], h_table;
    unsigned seed_lo, seed_hi;
    unsigned long long env_id_base;
    float rst_q[MAXJ], rst_objz;  // settled robot pose / object height recorded at the last full reset (snapshot auto-reset)
    float rst_ee[6]; int rst_ok;  // ... and the end-effector position / Euler angles of that pose (lane-per-env kernels: the first
                                  // observation of a restarted episode needs no kinematic sweep); rst_ok: recorded and a simple-class state
    int   use_ik, ik_iters;       // Cartesian control (use_IK=1): damped-least-squares IK
    float ik_l2, ik_res, home_hand[6], rws[3][2];   // lambda^2, position residual, home hand pose, robot workspace
    int   robot, reward_type, ctrl_ori;             // PBRE_ROBOT_*; iCub push reward variant; IK mode: orientation part of the action
    float ik_ps, ik_rs, eu_lim[3][2], ik_off[3];    // action scales, Euler limits, hand COM frame -> link frame offset
    int   ik_abs;                                   // IK actions are absolute hand poses (robot-level apply_action, icub_env.py:262-330) instead of scaled increments
    float cmd_vmax, cmd_kp;                         // robot-level apply_action(max_vel): maxVelocity of the commanded motors (0: none) and their
    int   cmd_nj;                                   // positionGain (0: the hold gain); cmd_nj > 0: only the first cmd_nj DoF are commanded (pandaEnv, panda_env.py:284-290)
    int*  bad_count;                                // device counter of env-steps that met a non-finite state (NaN / Inf guard); may be null
    float res_lim;                                  // sqrt(pbre_physics.solver_residual_threshold): an env leaves the sweep loop after the first sweep whose largest
                                                    // velocity-level row change |delta impulse / jacDiagABInv| is <= res_lim (Bullet compares the squares); 0: never
    int*  sweeps;                                   // res_lim > 0: per-env count of the sweeps run in the step ([num_envs], this ctx's local env index); may be null
    // PBRE_SHAPE_HULL (obj_shape 3; pbre_set_object_hull): device table [HULL_V0 .. ) = hull_nv vertices x (x, y, z, 0), [HULL_T0 .. ) = hull_nf
    // triangles x (a[3], ab[3], ac[3], unit outward normal[3]) in the object's frame; hull_rb = the largest vertex distance from the origin
    const float* hull;
    int   hull_nv, hull_nf;
    float hull_rb;
    int   objv_seq;                                 // Core::step's `objv` side record: 0 = complete behind the block barrier (its producer is a sibling wave); else complete
                                                    // once its first word holds this number (its producer is a wave of another block: pbre_capi.hip k_fused)
};

constexpr int HULL_MAXV = 32, HULL_MAXF = 64, HULL_V0 = 0, HULL_T0 = 4 * HULL_MAXV, HULL_FLOATS = 4 * HULL_MAXV + 12 * HULL_MAXF;
struct HullTable { int nv = 0, nf = 0; float rb = 0.f; double half[3] = {0, 0, 0}; float data[HULL_FLOATS]; };

namespace detail {
struct Xf { double R[9]; double p[3]; };
inline Xf ident() { Xf x{}; x.R[0] = x.R[4] = x.R[8] = 1; return x; }
inline Xf mul(const Xf& a, const Xf& b) {
    Xf o{};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += a.R[i*3+k] * b.R[k*3+j]; o.R[i*3+j] = s; }
        o.p[i] = a.p[i] + a.R[i*3] * b.p[0] + a.R[i*3+1] * b.p[1] + a.R[i*3+2] * b.p[2];
    }
    return o;
}
}  // namespace detail

// Returns "" on success, else an error message.
// act_dof: DoF index of controlled joint k (k < n_ctrl); n_act of them are driven by the action in joint mode;
// observe_all: the observation reports every DoF (Panda) instead of the controlled joints (iCub).
template <class S>
inline std::string build_tables(const double* t, size_t n, const double* home, const double* gains /*kp_act,kd_act,kp_hold,kd_hold*/,
                                int n_act, const int* act_dof, int n_ctrl, bool observe_all, TablesT<S>& T) {
    using namespace detail;
    constexpr int W = S::W, NJ = S::NJ, NSUB = S::NSUB, NLEV = S::NLEV;
    std::memset(&T, 0, sizeof T);
    if (!t || n < 24 || t[0] != 1346523717.0 || t[1] != 1.0) return "robot_table: bad magic/version";
    const int nl = (int)t[2], ndof = (int)t[3], ee = (int)t[4], ns = (int)t[5];
    if (n < (size_t)(24 + nl * 40 + ns * 8)) return "robot_table: truncated";
    if (ndof > NJ) return "robot_table: more DoF than robot lanes of this kernel shape";
    if (ns > (W >= 64 ? 64 : 16)) return "robot_table: too many collision spheres";
    if ((int)t[18] != 1) return "robot_table: floating-base robots are not supported by this kernel";
    if (ee < 0 || ee >= nl) return "robot_table: ee_link out of range";
    auto L = [&](int i) { return t + 24 + i * 40; };
    std::vector<int> lane_of(nl, -1), owner(nl, -1);
    std::vector<Xf> to_owner(nl);          // link frame expressed in its owner's (movable link) frame; owner == -1: in world
    Xf base = ident();
    for (int k = 0; k < 3; k++) base.p[k] = t[6 + k];
    for (int k = 0; k < 9; k++) base.R[k] = t[9 + k];
    int nsub[W] = {0};
    for (int l = 0; l < W; l++) { for (int v = 0; v < NLEV; v++) T.anc[v][l] = -1; T.s_owner[l] = 0; }
    for (int i = 0; i < nl; i++) {
        const double* r = L(i);
        const int par = (int)r[0], jt = (int)r[1], dof = (int)r[33];
        if (par >= i) return "robot_table: links must be in topological order";
        Xf X = ident();
        for (int k = 0; k < 3; k++) X.p[k] = r[5 + k];
        for (int k = 0; k < 9; k++) X.R[k] = r[8 + k];
        // frame of the parent link expressed in the parent's owner frame (or world)
        Xf P = par < 0 ? base : to_owner[par];
        int pown = par < 0 ? -1 : owner[par];
        Xf J = mul(P, X);                   // this joint frame in pown's frame (q = 0)
        if (jt != 0) {
            if (dof < 0 || dof >= NJ) return "robot_table: bad dof index";
            const int ln = dof;
            lane_of[i] = ln; owner[i] = i; to_owner[i] = ident();
            T.jtype[ln] = jt;
            for (int k = 0; k < 3; k++) { T.axis[k][ln] = (float)r[2 + k]; T.p0[k][ln] = (float)J.p[k]; }
            for (int k = 0; k < 9; k++) T.R0[k][ln] = (float)J.R[k];
            T.anc[0][ln] = pown < 0 ? -1 : lane_of[pown];
            T.lower[ln] = (float)r[30]; T.upper[ln] = (float)r[31]; T.jdamp[ln] = (float)r[32];
        } else {
            owner[i] = pown; to_owner[i] = J;
        }
        // rigid sub-body
        const double mass = r[17];
        bool has_inertia = false;
        for (int k = 0; k < 9; k++) has_inertia = has_inertia || r[21 + k] != 0.0;
        if (owner[i] >= 0 && (mass > 0 || has_inertia)) {
            const int ln = lane_of[owner[i]];
            // A massless link keeps its rotational inertia (PyBullet keeps <inertia> with URDF_USE_INERTIA_FROM_FILE,
            // reference panda_env.py:53; e.g. panda_link8).  It moves rigidly with the lane, and every term it enters
            // is linear in the tensor, so it is folded exactly into the lane's first sub-body.
            const bool fold = mass <= 0 && nsub[ln] > 0;
            if (!fold && nsub[ln] >= NSUB) return "robot_table: too many fixed-attached bodies on one movable link";
            const int b = fold ? 0 : nsub[ln]++;
            const Xf& F = to_owner[i];
            double c[3], I[9], RI[9], Io[9];
            for (int a = 0; a < 3; a++) c[a] = F.p[a] + F.R[a*3] * r[18] + F.R[a*3+1] * r[19] + F.R[a*3+2] * r[20];
            for (int k = 0; k < 9; k++) I[k] = r[21 + k];
            for (int a = 0; a < 3; a++) for (int bb = 0; bb < 3; bb++) { double s = 0; for (int k = 0; k < 3; k++) s += F.R[a*3+k] * I[k*3+bb]; RI[a*3+bb] = s; }
            for (int a = 0; a < 3; a++) for (int bb = 0; bb < 3; bb++) { double s = 0; for (int k = 0; k < 3; k++) s += RI[a*3+k] * F.R[bb*3+k]; Io[a*3+bb] = s; }
            if (!fold) {
                T.sb_m[b][ln] = (float)mass;
                for (int a = 0; a < 3; a++) T.sb_c[b][a][ln] = (float)c[a];
            }
            T.sb_I[b][0][ln] += (float)Io[0]; T.sb_I[b][1][ln] += (float)Io[4]; T.sb_I[b][2][ln] += (float)Io[8];
            T.sb_I[b][3][ln] += (float)Io[1]; T.sb_I[b][4][ln] += (float)Io[2]; T.sb_I[b][5][ln] += (float)Io[5];
        }
    }
    // ancestor tables
    for (int v = 1; v < NLEV; v++) for (int l = 0; l < NJ; l++) { int a = T.anc[v-1][l]; T.anc[v][l] = a < 0 ? -1 : T.anc[v-1][a]; }
    for (int l = 0; l < NJ; l++) { if (!T.jtype[l]) continue; for (int a = l; a >= 0; a = T.anc[0][a]) T.amask[a >> 5][l] |= (int)(1u << (a & 31)); }
    for (int l = 0; l < NJ; l++) for (int i = 0; i < NJ; i++) if (T.jtype[i] && ((unsigned)T.amask[l >> 5][i] >> (l & 31) & 1u)) T.dmask[i >> 5][l] |= (int)(1u << (i & 31));
    for (int l = 0; l < NJ; l++) {           // NLEV doubling steps accumulate over self + (2^NLEV - 1) ancestors
        int depth = 0;
        for (int a = T.anc[0][l]; a >= 0; a = T.anc[0][a]) depth++;
        if (T.jtype[l] && depth > (1 << NLEV) - 1) return "robot_table: chain deeper than the pointer-jumping levels cover";
    }
    for (int l = 0; l < W; l++) { T.act_idx[l] = -1; T.obs_idx[l] = -1; T.tip_of[l] = -1; }
    if (n_act > n_ctrl) return "more action joints than controlled joints";
    for (int k = 0; k < n_ctrl; k++) {
        const int d = act_dof[k];
        if (d < 0 || d >= ndof || T.obs_idx[d] >= 0) return "act_dof: bad or repeated DoF index";
        if (k < n_act) T.act_idx[d] = k;
        T.obs_idx[d] = k;
    }
    T.n_obs_j = n_ctrl;
    for (int l = 0; l < ndof; l++) T.blocked[l] = !observe_all && T.obs_idx[l] < 0;
    if (observe_all) { for (int l = 0; l < ndof; l++) T.obs_idx[l] = l; T.n_obs_j = ndof; }
    for (int l = 0; l < ndof; l++) {
        T.home[l] = (float)home[l]; T.rst_q[l] = (float)home[l];
        const bool act = T.act_idx[l] >= 0;
        T.kp_act[l] = (float)(act ? gains[0] : gains[2]); T.kd_act[l] = (float)(act ? gains[1] : gains[3]);
        T.kp_hold[l] = (float)gains[2]; T.kd_hold[l] = (float)gains[3];
    }
    // end effector: getLinkState()[0] is the link COM frame (reference panda_env.py:147,155)
    {
        const double* r = L(ee);
        if (owner[ee] < 0) return "robot_table: end effector is fixed to the base";
        T.ee_owner = lane_of[owner[ee]];
        const Xf& F = to_owner[ee];
        for (int k = 0; k < 9; k++) T.ee_R[k] = (float)F.R[k];
        for (int a = 0; a < 3; a++) T.ee_p[a] = (float)(F.p[a] + F.R[a*3] * r[18] + F.R[a*3+1] * r[19] + F.R[a*3+2] * r[20]);
        for (int a = 0; a < 3; a++) T.ee_lp[a] = (float)F.p[a];
        for (int l = 0; l < NJ; l++) T.on_chain[l] = (int)((unsigned)T.amask[l >> 5][T.ee_owner] >> (l & 31) & 1u);
        // (link record [37]: a joint the inverse kinematics must not move -- the virtual joints of a soft-pinned floating base)
        for (int i = 0; i < nl; i++) if (lane_of[i] >= 0 && owner[i] == i && (int)L(i)[37] != 0) T.on_chain[lane_of[i]] = 0;
        for (int i = 0; i < nl; i++) if (lane_of[i] >= 0 && owner[i] == i && (int)L(i)[37] != 0 && L(i)[35] > 0) T.mforce[lane_of[i]] = (float)L(i)[35];
    }
    for (int s = 0; s < ns; s++) {
        const double* r = t + 24 + nl * 40 + s * 8;
        const int li = (int)r[0];
        if (li < 0 || li >= nl || owner[li] < 0) return "robot_table: sphere on a base-fixed link";
        const Xf& F = to_owner[li];
        T.s_owner[s] = lane_of[owner[li]]; T.s_valid[s] = 1;
        for (int a = 0; a < 3; a++) T.s_c[a][s] = (float)(F.p[a] + F.R[a*3] * r[1] + F.R[a*3+1] * r[2] + F.R[a*3+2] * r[3]);
        T.s_r[s] = (float)r[4]; T.s_mu[s] = (float)r[5];
        if ((int)r[6] > 0) {               // fingertip sphere: slot + 1
            if ((int)r[6] > S::NTIP) return "robot_table: more fingertips than this kernel shape reports";
            T.tip_of[T.s_owner[s]] = (int)r[6] - 1;
        }
    }
    T.ndof = ndof; T.n_act = n_act; T.nspheres = ns;
    return "";
}

}  // namespace pbre
