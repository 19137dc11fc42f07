// pbre_host.hpp -- host-side pieces of the C-ABI that do not depend on HIP: default
// configuration, pbre_config -> Tables/Params, observation limits.  Shared by
// pbre_capi.hip (product) and tests/host_emu (lane emulation, tests only).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include <cstring>
#include <string>
#include "../../include/pbre.h"
#include "pbre_tables.hpp"

namespace pbre {

inline void default_physics(pbre_physics& p) {
    std::memset(&p, 0, sizeof p);
    p.dt = 1.0 / 240.0;                 // reference panda_push_gym_env.py:39
    p.gravity_z = -9.8;                 // :126
    p.solver_iters = 150;               // :122
    p.erp = 0.2; p.linear_slop = 1e-5; p.contact_margin = 1e-3;   // Bullet defaults [EXT-UNVERIFIED], DESIGN.md
    p.lin_damping = 0.04; p.ang_damping = 0.04; p.max_coord_vel = 100.0;
    p.max_motor_impulse = 100000.0 / 240.0;   // pybullet default force 1e5 N * dt
    p.limit_max_impulse = 100.0;
    p.table_c[0] = 0.85; p.table_c[1] = 0.0; p.table_c[2] = 0.6;   // table.urdf top slab at world_env.py:65 base pos
    p.table_h[0] = 0.75; p.table_h[1] = 0.5; p.table_h[2] = 0.025;
    p.table_mu = 0.5; p.ground_z = 0.0;
    p.obj_h[0] = p.obj_h[1] = p.obj_h[2] = 0.025;                 // cube_small.urdf
    p.obj_mass = 0.1;
    p.obj_inertia[0] = p.obj_inertia[1] = p.obj_inertia[2] = 0.1 * (0.05 * 0.05 + 0.05 * 0.05) / 12.0;
    p.obj_mu = 1.0;
}

inline int default_config(pbre_config* c, int robot, int task) {
    const bool panda_arm = robot == PBRE_ROBOT_PANDA_ARM;
    if (panda_arm) robot = PBRE_ROBOT_PANDA;
    if (!c || (robot != PBRE_ROBOT_PANDA && robot != PBRE_ROBOT_ICUB && robot != PBRE_ROBOT_ICUB_HANDS) || (task != PBRE_TASK_REACH && task != PBRE_TASK_PUSH && task != PBRE_TASK_PUSH_GOAL)) return PBRE_E_ARG;
    std::memset(c, 0, sizeof *c);
    c->robot = robot; c->task = task; c->num_envs = 1; c->device_id = 0; c->seed = 1234;
    c->use_ik = 0; c->num_controlled_joints = 7; c->action_repeat = 1; c->max_steps = 1000;
    c->target_dist_min = task == PBRE_TASK_REACH ? 0.03 : 0.1;   // panda_push_gym_env.py:52, panda_reach_gym_env.py:47
    c->act_scale = 0.05;                                        // panda_push_gym_env.py:225
    c->kp_act = 0.5; c->kd_act = 1.0;                           // panda_env.py:308
    c->kp_hold = 0.2; c->kd_hold = 1.0;                         // panda_env.py:76
    c->h_table = 0.625;                                         // world_env.py:68-69
    c->ws_lim[0][0] = 0.3; c->ws_lim[0][1] = 0.65;              // panda_env.py:37 (x,y); world_env.py:72 (z)
    c->ws_lim[1][0] = -0.3; c->ws_lim[1][1] = 0.3;
    c->ws_lim[2][0] = 0.625; c->ws_lim[2][1] = 0.925;
    const double home[9] = {0.0, -0.54, 0.0, -2.6, -0.30, 2.0, 1.0, 0.02, 0.02};   // panda_env.py:19-23
    for (int k = 0; k < 9; k++) c->home[k] = home[k];
    default_physics(c->phys);
    c->ik_damping = 0.1; c->ik_residual = 1e-3; c->ik_max_iters = 100;              // panda_env.py:269-272
    const double hh[6] = {0.2, 0.0, 0.8, 3.14159265358979323846, 0.0, 0.0};        // panda_env.py:85-88
    for (int k = 0; k < 6; k++) c->home_hand_pose[k] = hh[k];
    c->robot_ws[0][0] = 0.3; c->robot_ws[0][1] = 0.65; c->robot_ws[1][0] = -0.3; c->robot_ws[1][1] = 0.3;   // panda_env.py:37
    c->robot_ws[2][0] = task == PBRE_TASK_REACH ? c->h_table : c->h_table - 0.2; c->robot_ws[2][1] = 1.5;   // panda_reach_gym_env.py:69 / panda_push_gym_env.py:74
    const double PI = 3.14159265358979323846;
    c->control_orientation = 1; c->reward_type = 1; c->num_joints_ctrl = 7;
    for (int k = 0; k < 64; k++) c->act_dof[k] = k < 7 ? k : -1;
    c->ik_pos_scale = 0.005; c->ik_rot_scale = 0.01;                                // panda_push_gym_env.py:200-203
    for (int k = 0; k < 3; k++) { c->eu_lim[k][0] = -PI; c->eu_lim[k][1] = PI; }     // panda_env.py:38
    if (panda_arm) {
        // pandaEnv used alone (panda_env.py:25-91, 195-365) in the scene of examples/helloworlds/helloworld_panda.py:72-85: table.urdf at
        // (1, 0, 0), a lego brick dropped at (0.5, 0, 0.8) -- stand-in [pybullet_data/lego is not available]: a 3.2 x 2.4 x 5 cm box of
        // 0.1 kg, tall enough for the demo's grasp height (hand at z = 0.67, 4.5 cm above the table top) to close the fingers on it
        c->task = PBRE_TASK_REACH; c->robot_level = 1; c->ik_absolute = 1;
        c->max_steps = 1 << 30; c->target_dist_min = -1.0;                          // no episode logic at the robot level
        c->num_controlled_joints = 9; c->num_joints_ctrl = 9;                       // joint_action_space = 9 (panda_env.py:26)
        for (int k = 0; k < 64; k++) c->act_dof[k] = k < 9 ? k : -1;
        c->ws_lim[0][0] = 0.35; c->ws_lim[0][1] = 0.70;                             // object dropped at x = 0.5, y = 0
        c->robot_ws[2][0] = 0.65; c->robot_ws[2][1] = 1.5;                          // panda_env.py:37
        c->ik_pos_scale = 1.0; c->ik_rot_scale = 1.0;
        c->phys.table_c[0] = 1.0;
        c->phys.obj_h[0] = 0.016; c->phys.obj_h[1] = 0.012; c->phys.obj_h[2] = 0.025;
        c->phys.obj_mass = 0.1;
        for (int k = 0; k < 3; k++) {
            const double a = 2 * c->phys.obj_h[(k + 1) % 3], b = 2 * c->phys.obj_h[(k + 2) % 3];
            c->phys.obj_inertia[k] = c->phys.obj_mass * (a * a + b * b) / 12.0;
        }
    }
    if (robot == PBRE_ROBOT_ICUB_HANDS) {
        c->robot_level = 1;
        // iCubHandsEnv defaults, left arm (icub_env_with_hands.py:51-83): robot-level interface, joint control.  DoF order of the
        // simulated model (legs pruned): torso 0..2, left arm 3..9, left hand 10..29, neck 30..32, right arm 33..39, right hand 40..59
        c->task = PBRE_TASK_REACH; c->use_ik = 0; c->control_orientation = 1; c->ik_absolute = 1;
        c->max_steps = 1 << 30; c->target_dist_min = -1.0;                          // no episode logic at the robot level
        c->num_controlled_joints = 37; c->num_joints_ctrl = 37;
        {   // _joints_to_control (:123-141; `a or b and c` keeps both arms) in joint-index order: torso, left arm, left hand, right arm
            int k = 0;
            for (int d = 0; d < 30; d++) c->act_dof[k++] = d;
            for (int d = 33; d < 40; d++) c->act_dof[k++] = d;
            for (; k < 64; k++) c->act_dof[k] = -1;
        }
        for (int k = 0; k < 64; k++) c->home[k] = 0.0;
        c->home[3] = -0.51; c->home[4] = 0.7; c->home[6] = 1.22; c->home[30] = 0.008; c->home[33] = -0.51; c->home[34] = 0.7; c->home[36] = 1.22;
        c->ws_lim[0][0] = 0.35; c->ws_lim[0][1] = 0.70; c->ws_lim[1][0] = -0.33; c->ws_lim[1][1] = 0.27;   // object dropped at (0.5, -0.03) (helloworld_icub.py:51)
        c->robot_ws[0][0] = 0.15; c->robot_ws[0][1] = 0.50; c->robot_ws[1][0] = -0.3; c->robot_ws[1][1] = 0.3; c->robot_ws[2][0] = 0.5; c->robot_ws[2][1] = 1.0;   // :63
        const double hl[6] = {0.2, 0.3, 0.8, -PI, 0.0, -PI / 2};                    // :77
        for (int k = 0; k < 6; k++) c->home_hand_pose[k] = hl[k];
        c->eu_lim[0][0] = -1.5 * PI; c->eu_lim[0][1] = -PI / 2; c->eu_lim[1][0] = -PI / 2; c->eu_lim[1][1] = PI / 2; c->eu_lim[2][0] = 0.0; c->eu_lim[2][1] = -PI;   // :78
        c->ik_pos_scale = 1.0; c->ik_rot_scale = 1.0;
        c->ik_link_offset[0] = -0.011682; c->ik_link_offset[1] = 0.051355; c->ik_link_offset[2] = 0.000577;   // :163
        c->phys.table_c[0] = 1.0;                                                   // table.urdf at (1, 0, 0) (helloworld_icub.py:50)
        c->phys.obj_h[0] = 0.025; c->phys.obj_h[1] = 0.0375; c->phys.obj_h[2] = 0.025;   // foam-brick sized box (the YCB asset is not available)
        c->phys.obj_mass = 0.028;
        c->phys.implicit_joint_damping = 1;                                         // finger joints: c dt / I = 3.7, see include/pbre.h
        for (int k = 0; k < 3; k++) {
            const double a = 2 * c->phys.obj_h[(k + 1) % 3], b = 2 * c->phys.obj_h[(k + 2) % 3];
            c->phys.obj_inertia[k] = c->phys.obj_mass * (a * a + b * b) / 12.0;
        }
    }
    if (robot == PBRE_ROBOT_ICUB) {
        // iCub*GymEnv defaults, left arm (icub_env.py:52-82, icub_reach_gym_env.py:27-51, icub_push_gym_env.py:27-57);
        // the right arm differs in home_hand_pose, eu_lim[2], ik_link_offset and act_dof, which the caller sets
        c->use_ik = 1; c->control_orientation = 0; c->max_steps = 2000; c->target_dist_min = 0.03;
        c->num_controlled_joints = 10; c->num_joints_ctrl = 10;
        for (int k = 0; k < 64; k++) c->act_dof[k] = k < 10 ? k : -1;                // torso 0..2, left arm 3..9 of the simulated model (legs pruned)
        for (int k = 0; k < 64; k++) c->home[k] = 0.0;
        c->home[3] = -0.51; c->home[4] = 0.7; c->home[6] = 1.22;                     // l_shoulder_pitch, l_shoulder_roll, l_elbow
        c->home[13] = -0.51; c->home[14] = 0.7; c->home[16] = 1.22;                  // right arm
        c->home[10] = 0.008;                                                        // neck_pitch
        c->ws_lim[0][0] = 0.1; c->ws_lim[0][1] = 0.45;                              // icub_env.py:62
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) c->robot_ws[a][b] = c->ws_lim[a][b];
        c->robot_ws[2][0] = c->h_table; c->robot_ws[2][1] = 1.0;                    // icub_reach_gym_env.py:78-80
        const double hl[6] = {0.3, 0.26, 0.8, 0.0, 0.0, 0.0};                       // icub_env.py:67
        for (int k = 0; k < 6; k++) c->home_hand_pose[k] = hl[k];
        for (int k = 0; k < 3; k++) { c->eu_lim[k][0] = -PI / 2; c->eu_lim[k][1] = PI / 2; }
        c->ik_pos_scale = 0.005; c->ik_rot_scale = 0.02;                            // 0.01 / 0.02 with control_orientation=1
        c->ik_link_offset[0] = -0.064768; c->ik_link_offset[1] = -0.00563; c->ik_link_offset[2] = -0.02266;   // icub_env.py:256
    }
    return PBRE_OK;
}

// pbre_physics -> the device parameter block (the robot tables do not depend on pbre_physics); false on bad values
// Does a pbre_set_physics call change what the settled snapshot of the last full reset depends on (scene geometry, contact margin /
// slop / ERP, gravity, time step)?  Then the snapshot -- settled robot pose, object rest height, cached end-effector pose -- describes a
// scene that no longer exists: pbre_reset_snapshot and the in-kernel auto-reset must not use it (mass, friction, damping and the
// solver's impulse bounds leave the rest pose alone: domain randomisation keeps the snapshot).
inline bool snapshot_relevant_change(const pbre_physics& a, const pbre_physics& b) {
    bool ch = a.dt != b.dt || a.gravity_z != b.gravity_z || a.erp != b.erp || a.linear_slop != b.linear_slop || a.contact_margin != b.contact_margin ||
              a.ground_z != b.ground_z;
    for (int k = 0; k < 3; k++) ch = ch || a.table_c[k] != b.table_c[k] || a.table_h[k] != b.table_h[k] || a.obj_h[k] != b.obj_h[k];
    ch = ch || a.obj_shape != b.obj_shape;
    return ch;
}
inline const char* stale_snapshot_msg() {
    return "the scene changed (pbre_set_physics: geometry / contact parameters) since the last full pbre_reset: the settled snapshot that "
           "pbre_reset_snapshot and PBRE_F_AUTO_RESET restart from is stale -- call pbre_reset for the whole batch first";
}
inline bool apply_physics(const pbre_physics& p, Params& P2) {
    if (p.solver_iters <= 0 || p.dt <= 0 || p.obj_mass <= 0) return false;
    P2.dt = (float)p.dt; P2.inv_dt = (float)(1.0 / p.dt); P2.gz = (float)p.gravity_z; P2.iters = p.solver_iters;
    P2.erp = (float)p.erp; P2.slop = (float)p.linear_slop; P2.margin = (float)p.contact_margin;
    P2.kl = (float)p.lin_damping; P2.ka = (float)p.ang_damping; P2.vmax = (float)p.max_coord_vel;
    P2.motor_imp = (float)p.max_motor_impulse; P2.limit_imp = (float)p.limit_max_impulse;
    P2.jd_dt = p.implicit_joint_damping ? (float)p.dt : 0.f;
    for (int k = 0; k < 3; k++) { P2.tab_c[k] = (float)p.table_c[k]; P2.tab_h[k] = (float)p.table_h[k]; P2.obj_h[k] = (float)p.obj_h[k]; P2.obj_I[k] = (float)p.obj_inertia[k]; }
    P2.tab_mu = (float)p.table_mu; P2.ground_z = (float)p.ground_z; P2.obj_m = (float)p.obj_mass; P2.obj_mu = (float)p.obj_mu;
    P2.obj_iso = (P2.obj_I[0] == P2.obj_I[1] && P2.obj_I[1] == P2.obj_I[2]) ? 1 : 0;
    if (p.obj_shape < 0 || p.obj_shape > 3 || (p.obj_shape == 3 && !P2.hull)) return false;      // (a hull needs pbre_set_object_hull first)
    P2.obj_shape = p.obj_shape;
    if (!(p.solver_residual_threshold >= 0)) return false;
    P2.res_lim = (float)std::sqrt(p.solver_residual_threshold);
    return true;
}

// The face table of a convex-hull object from its vertices (pbre_set_object_hull; layout: pbre_tables.hpp HullTable).  Supporting planes by
// brute force over the vertex triples (n <= 32), coplanar triples merged into one polygonal face (vertices ordered by angle about the
// face's centroid, fan-triangulated): at most 2 n - 4 triangles for vertices in general position.  Returns "" or an error text.
inline std::string build_hull(const double* v, int n, HullTable& H) {
    if (!v || n < 4 || n > HULL_MAXV) return "pbre_set_object_hull: n_verts must be 4..32";
    double scale = 0;
    for (int i = 0; i < 3 * n; i++) { if (!std::isfinite(v[i])) return "pbre_set_object_hull: non-finite vertex"; scale = std::max(scale, std::fabs(v[i])); }
    if (!(scale > 0)) return "pbre_set_object_hull: degenerate vertex set";
    const double eps = 1e-7 * scale;
    struct Plane { double n[3], d; };
    std::vector<Plane> planes;
    auto V = [&](int i, int k) { return v[3 * i + k]; };
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) for (int k = j + 1; k < n; k++) {
        const double u[3] = {V(j, 0) - V(i, 0), V(j, 1) - V(i, 1), V(j, 2) - V(i, 2)}, w[3] = {V(k, 0) - V(i, 0), V(k, 1) - V(i, 1), V(k, 2) - V(i, 2)};
        double nn[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
        const double len = std::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
        if (len < 1e-9 * scale * scale) continue;                      // collinear
        for (double& x : nn) x /= len;
        int pos = 0, neg = 0;
        for (int l = 0; l < n; l++) {
            const double d = nn[0] * (V(l, 0) - V(i, 0)) + nn[1] * (V(l, 1) - V(i, 1)) + nn[2] * (V(l, 2) - V(i, 2));
            if (d > eps) pos++; else if (d < -eps) neg++;
        }
        if (pos && neg) continue;
        if (!pos && !neg) return "pbre_set_object_hull: the vertices are coplanar";
        const double sg = pos ? -1.0 : 1.0;
        Plane pl; for (int t = 0; t < 3; t++) pl.n[t] = sg * nn[t];
        pl.d = pl.n[0] * V(i, 0) + pl.n[1] * V(i, 1) + pl.n[2] * V(i, 2);
        bool dup = false;
        for (const Plane& q : planes) if (q.n[0] * pl.n[0] + q.n[1] * pl.n[1] + q.n[2] * pl.n[2] > 1.0 - 1e-10 && std::fabs(q.d - pl.d) <= eps) { dup = true; break; }
        if (!dup) planes.push_back(pl);
    }
    if (planes.size() < 4) return "pbre_set_object_hull: degenerate vertex set (no volume)";
    H.nv = n; H.nf = 0; H.rb = 0.f;
    std::memset(H.data, 0, sizeof H.data);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < n; i++) {
        double r2 = 0;
        for (int k = 0; k < 3; k++) { H.data[HULL_V0 + 4 * i + k] = (float)V(i, k); r2 += V(i, k) * V(i, k); lo[k] = std::min(lo[k], V(i, k)); hi[k] = std::max(hi[k], V(i, k)); }
        H.rb = std::max(H.rb, (float)std::sqrt(r2));
    }
    for (int k = 0; k < 3; k++) H.half[k] = std::max(hi[k], -lo[k]);      // (about the origin = the centre of mass: what the rest-height guess of a reset needs)
    for (const Plane& pl : planes) {
        std::vector<int> on;
        double c[3] = {0, 0, 0};
        for (int l = 0; l < n; l++) if (std::fabs(pl.n[0] * V(l, 0) + pl.n[1] * V(l, 1) + pl.n[2] * V(l, 2) - pl.d) <= eps) { on.push_back(l); for (int k = 0; k < 3; k++) c[k] += V(l, k); }
        if (on.size() < 3) continue;
        for (double& x : c) x /= (double)on.size();
        double u[3] = {V(on[0], 0) - c[0], V(on[0], 1) - c[1], V(on[0], 2) - c[2]};
        const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (!(ul > 0)) continue;
        for (double& x : u) x /= ul;
        const double w[3] = {pl.n[1] * u[2] - pl.n[2] * u[1], pl.n[2] * u[0] - pl.n[0] * u[2], pl.n[0] * u[1] - pl.n[1] * u[0]};
        std::vector<std::pair<double, int>> ang;
        for (int l : on) {
            const double r[3] = {V(l, 0) - c[0], V(l, 1) - c[1], V(l, 2) - c[2]};
            ang.push_back({std::atan2(r[0] * w[0] + r[1] * w[1] + r[2] * w[2], r[0] * u[0] + r[1] * u[1] + r[2] * u[2]), l});
        }
        std::sort(ang.begin(), ang.end());                            // counter-clockwise about the outward normal
        for (size_t t = 1; t + 1 < ang.size(); t++) {
            if (H.nf >= HULL_MAXF) return "pbre_set_object_hull: more than 64 triangles";
            const int a = ang[0].second, b = ang[t].second, cc = ang[t + 1].second;
            float* T = H.data + HULL_T0 + 12 * H.nf++;
            for (int k = 0; k < 3; k++) { T[k] = (float)V(a, k); T[3 + k] = (float)(V(b, k) - V(a, k)); T[6 + k] = (float)(V(cc, k) - V(a, k)); T[9 + k] = (float)pl.n[k]; }
        }
    }
    return "";
}

// number of DoF the RobotTable declares (0 if it is not a table): selects the lane shape
inline int table_ndof(const pbre_config& c) {
    return (c.robot_table && c.robot_table_len >= 24 && c.robot_table[0] == 1346523717.0) ? (int)c.robot_table[3] : 0;
}

template <class S>
inline std::string make_tables(const pbre_config& c, TablesT<S>& T, Params& P) {
    constexpr int NJ = S::NJ;
    if (c.num_envs <= 0) return "num_envs must be positive";
    if (c.action_repeat < 1 || c.action_repeat > 64) return "action_repeat out of range";
    if (c.num_controlled_joints < 1 || c.num_controlled_joints > 64 || c.num_controlled_joints > NJ) return "num_controlled_joints out of range";
    int act_dof[64], n_ctrl = c.num_joints_ctrl;
    for (int k = 0; k < 64; k++) act_dof[k] = c.act_dof[k];
    if (c.robot == PBRE_ROBOT_PANDA) { n_ctrl = c.num_controlled_joints; for (int k = 0; k < 64; k++) act_dof[k] = k; }   // panda_env.py:293-310: the first n joints
    if (n_ctrl < c.num_controlled_joints || n_ctrl > 64) return "num_joints_ctrl out of range";
    if (c.use_ik && (c.ik_max_iters <= 0 || c.ik_damping <= 0)) return "bad IK parameters";
    if (c.robot != PBRE_ROBOT_PANDA && c.robot != PBRE_ROBOT_ICUB && c.robot != PBRE_ROBOT_ICUB_HANDS) return "unknown robot";
    if ((c.robot == PBRE_ROBOT_ICUB_HANDS || c.robot_level != 0) != S::MREC) return "the robot-level interface (iCub with hands, pbre_config.robot_level) needs a motor-record kernel shape (and only those shapes use it)";
    if (S::MREC && (c.action_repeat != 1 || (c.flags & PBRE_F_AUTO_RESET))) return "robot-level interface: action_repeat / auto-reset are not part of it";
    const double gains[4] = {c.kp_act, c.kd_act, c.kp_hold, c.kd_hold};
    std::string e = build_tables<S>(c.robot_table, c.robot_table_len, c.home, gains, c.num_controlled_joints, act_dof, n_ctrl,
                                    c.robot == PBRE_ROBOT_PANDA, T);
    if (!e.empty()) return e;
    const pbre_physics& p = c.phys;
    if (p.solver_iters <= 0 || p.dt <= 0) return "bad physics parameters";
    std::memset(&P, 0, sizeof P);
    P.dt = (float)p.dt; P.inv_dt = (float)(1.0 / p.dt); P.gz = (float)p.gravity_z; P.iters = p.solver_iters;
    P.erp = (float)p.erp; P.slop = (float)p.linear_slop; P.margin = (float)p.contact_margin;
    P.kl = (float)p.lin_damping; P.ka = (float)p.ang_damping; P.vmax = (float)p.max_coord_vel;
    P.motor_imp = (float)p.max_motor_impulse; P.limit_imp = (float)p.limit_max_impulse;
    P.jd_dt = p.implicit_joint_damping ? (float)p.dt : 0.f;
    for (int k = 0; k < 3; k++) { P.tab_c[k] = (float)p.table_c[k]; P.tab_h[k] = (float)p.table_h[k]; P.obj_h[k] = (float)p.obj_h[k]; P.obj_I[k] = (float)p.obj_inertia[k]; }
    P.tab_mu = (float)p.table_mu; P.ground_z = (float)p.ground_z; P.obj_m = (float)p.obj_mass; P.obj_mu = (float)p.obj_mu;
    P.obj_iso = (P.obj_I[0] == P.obj_I[1] && P.obj_I[1] == P.obj_I[2]) ? 1 : 0;
    if (p.obj_shape < 0 || p.obj_shape > 2) return "bad physics parameters (obj_shape; a convex hull is set with pbre_set_object_hull after pbre_create)";
    P.obj_shape = p.obj_shape;
    if (!(p.solver_residual_threshold >= 0)) return "bad physics parameters (solver_residual_threshold)";
    P.res_lim = (float)std::sqrt(p.solver_residual_threshold);
    P.task = c.task; P.max_steps = c.max_steps; P.flags = c.flags;
    P.dist_min = (float)c.target_dist_min; P.act_scale = (float)c.act_scale;
    P.obj_std = (float)c.obj_pose_rnd_std; P.tg_std = (float)c.tg_pose_rnd_std;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 2; b++) P.ws[a][b] = (float)c.ws_lim[a][b];
    P.h_table = (float)c.h_table;
    P.seed_lo = (unsigned)c.seed; P.seed_hi = (unsigned)(c.seed >> 32);
    P.env_id_base = c.env_id_base;
    P.use_ik = c.use_ik ? 1 : 0; P.ik_iters = c.ik_max_iters; P.ik_l2 = (float)(c.ik_damping * c.ik_damping); P.ik_res = (float)c.ik_residual;
    for (int k = 0; k < 6; k++) P.home_hand[k] = (float)c.home_hand_pose[k];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 2; b++) P.rws[a][b] = (float)c.robot_ws[a][b];
    P.robot = c.robot; P.reward_type = c.reward_type; P.ctrl_ori = c.control_orientation ? 1 : 0;
    P.ik_ps = (float)c.ik_pos_scale; P.ik_rs = (float)c.ik_rot_scale; P.ik_abs = c.ik_absolute ? 1 : 0;
    for (int a = 0; a < 3; a++) { P.ik_off[a] = (float)c.ik_link_offset[a]; for (int b = 0; b < 2; b++) P.eu_lim[a][b] = (float)c.eu_lim[a][b]; }
    for (int k = 0; k < NJ; k++) P.rst_q[k] = T.home[k];
    P.rst_objz = (float)(c.h_table + p.obj_h[2]);      // refined from the settled state after the first full reset
    return "";
}

inline int act_dim_of(const pbre_config& c) { return c.use_ik ? (c.control_orientation ? 6 : 3) : c.num_controlled_joints; }
template <class S>
inline int obs_dim_of(const TablesT<S>& T, const Params& P) { return 9 + T.n_obs_j + 12 + (P.task != PBRE_TASK_REACH ? 3 : 0) + (S::NTIP ? S::NTIP + 2 : 0); }

// Observation limits exactly as the reference assembles them (panda_env.py:141-193 limits list,
// panda_push_gym_env.py:73-75 / panda_reach_gym_env.py:68-70 z-min, :177-185 extras).
// iCub: icub_env.py:202-249 (workspace, Euler limits of the arm, +-1 velocity, limits of the controlled joints).
template <class S>
inline void obs_limits(const pbre_config& c, const TablesT<S>& T, float* lo, float* hi) {
    const double PI = 3.14159265358979323846;
    int o = 0;
    auto put = [&](double a, double b) { lo[o] = (float)a; hi[o] = (float)b; o++; };
    if (c.robot != PBRE_ROBOT_PANDA) {
        for (int k = 0; k < 3; k++) put(c.robot_ws[k][0], c.robot_ws[k][1]);
        for (int k = 0; k < 3; k++) put(c.eu_lim[k][0], c.eu_lim[k][1]);
        for (int k = 0; k < 3; k++) put(-1, 1);
        for (int k = 0; k < c.num_joints_ctrl; k++) put(T.lower[c.act_dof[k]], T.upper[c.act_dof[k]]);
    } else {
    const double zmin = c.task != PBRE_TASK_REACH ? c.h_table - 0.2 : c.h_table;
    put(0.3, 0.65); put(-0.3, 0.3); put(zmin, 1.5);                 // robot workspace (panda_env.py:37)
    for (int k = 0; k < 3; k++) put(-PI, PI);
    for (int k = 0; k < 3; k++) put(-1, 1);
    for (int k = 0; k < T.ndof; k++) put(T.lower[k], T.upper[k]);
    }
    for (int k = 0; k < 3; k++) put(c.ws_lim[k][0], c.ws_lim[k][1]);
    for (int k = 0; k < 3; k++) put(-PI, PI);
    for (int k = 0; k < 3; k++) put(-0.5, 0.5);
    for (int k = 0; k < 3; k++) put(0, 2 * PI);
    if (c.task != PBRE_TASK_REACH) for (int k = 0; k < 3; k++) put(c.ws_lim[k][0], c.ws_lim[k][1]);
    if (S::NTIP) { for (int k = 0; k < S::NTIP; k++) put(0, 100); put(0, S::NTIP); put(0, S::NC_RO); }   // fingertip forces (N), tips in contact, contact points
}

}  // namespace pbre
