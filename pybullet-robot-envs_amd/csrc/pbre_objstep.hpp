// pbre_objstep.hpp -- the object's half of a simulation step in which the robot does not touch the object.
//
// Without a robot-object contact the constraint system of a step splits into two blocks that share no unknown: the robot rows
// (motors, joint limits, robot-table contacts) act on the joint velocities only, the object-table contact rows on the object twist
// only.  Projected Gauss-Seidel sweeps over rows of independent blocks commute, so the interleaved Bullet row order (SURVEY.md
// App. D; reference call site p.stepSimulation, icub_push_gym_env.py:263) gives each block exactly the iterates it would get alone.
// The lane-group kernels (pbre_core.hpp) spread one env over 16..64 lanes and pay every object row once per *group*; here one
// thread steps one env's object (64 envs per wave, 32x fewer issue slots per row for the iCub shape).  `kw_obj` runs this for every env
// ahead of `kw_step`, which uses the result in the groups it finds without a robot-object contact and solves the coupled system,
// object rows included, in the others (Core::step, `objv`).
//
// As in the object rows of the Panda's lane-per-env kernel (pbre_fast.hpp, step_t) the table normal is +z, so Bullet's
// btPlaneSpace1 friction directions are the constants (0,-1,0), (1,0,0), and rows are evaluated against the running velocity;
// unlike there the box may have any principal inertia (the foam-brick stand-in of the hands scene is not a cube).
#pragma once
#include <math.h>
#include "pbre_tables.hpp"
#ifndef PBRE_HD
#define PBRE_HD
#endif
#include "pbre_math.hpp"

PBRE_FP_CONTRACT_FAST      // (pbre_math.hpp: contraction is stated per header)
namespace pbre {

// The object's collision primitive (Params::obj_shape, include/pbre.h PBRE_SHAPE_*): the reference's object list (world_env.py:18-25,
// 179-216) has round members -- YcbTennisBall, the cans, pear, duck_vhacd -- that a box stand-in makes slide where they roll.
// Candidate contact points against the support surface, as offsets from the centre in WORLD axes, 8 slots like the box's vertices:
//   box       the 8 vertices;
//   sphere    slot 0: the lowest point (0, 0, -r);
//   cylinder  (axis = local z, radius h[0], half height h[2]) per cap three rim points at 0 / 120 / 240 degrees (slots 0-2 bottom cap,
//             4-6 top cap: an upright can stands on a tripod) and the rim's lowest point (slots 3 / 7: the generator a lying can rolls
//             on; unused while the axis is vertical).
// Same rules as oracle/pbre_oracle.c: shape_candidate().
struct Shapes {
    // R: row-major rotation object -> world; returns false for a slot the shape does not use
    static PBRE_HD bool candidate(int shape, const float* h, const float* R, int v, float& rx, float& ry, float& rz) {
        float lx, ly, lz;
        if (shape == 0) { lx = (v & 1) ? h[0] : -h[0]; ly = (v & 2) ? h[1] : -h[1]; lz = (v & 4) ? h[2] : -h[2]; }
        else if (shape == 1) { rx = 0.f; ry = 0.f; rz = -h[0]; return v == 0; }
        else {
            const float s = v < 4 ? -1.f : 1.f;
            const int k = v & 3;
            if (k < 3) {
                const float cs = k == 0 ? 1.f : -0.5f, sn = k == 0 ? 0.f : (k == 1 ? 0.86602540378443865f : -0.86602540378443865f);
                lx = h[0] * cs; ly = h[0] * sn; lz = s * h[2];
            } else {
                const float dx = -R[6], dy = -R[7];
                const float len = sqrtf(fmaf(dx, dx, dy * dy));
                if (!(len >= 1e-6f)) { rx = 0.f; ry = 0.f; rz = 0.f; return false; }
                lx = h[0] * dx / len; ly = h[0] * dy / len; lz = s * h[2];
            }
        }
        rx = fmaf(R[0], lx, fmaf(R[1], ly, R[2] * lz)); ry = fmaf(R[3], lx, fmaf(R[4], ly, R[5] * lz)); rz = fmaf(R[6], lx, fmaf(R[7], ly, R[8] * lz));
        return true;
    }
    // signed distance of a sphere (centre s, radius sr) to a round object at c with rotation R (shape 1 or 2; the box has its own code
    // in the callers); n: world normal object -> sphere, pb: point on the object.  oracle: sphere_shape().
    static PBRE_HD float sphere_round(int shape, const float* s, float sr, const float* c, const float* R, const float* h, float* n, float* pb) {
        const float d[3] = {s[0] - c[0], s[1] - c[1], s[2] - c[2]};
        if (shape == 1) {
            const float len = sqrtf(fmaf(d[0], d[0], fmaf(d[1], d[1], d[2] * d[2])));
            const bool deg = len < 1e-9f;
            const float il = 1.f / fmaxf(len, 1e-30f);
            n[0] = deg ? 0.f : d[0] * il; n[1] = deg ? 0.f : d[1] * il; n[2] = deg ? 1.f : d[2] * il;
            pb[0] = fmaf(n[0], h[0], c[0]); pb[1] = fmaf(n[1], h[0], c[1]); pb[2] = fmaf(n[2], h[0], c[2]);
            return len - h[0] - sr;
        }
        const float dl[3] = {fmaf(R[0], d[0], fmaf(R[3], d[1], R[6] * d[2])), fmaf(R[1], d[0], fmaf(R[4], d[1], R[7] * d[2])),
                             fmaf(R[2], d[0], fmaf(R[5], d[1], R[8] * d[2]))};
        const float rho = sqrtf(fmaf(dl[0], dl[0], dl[1] * dl[1]));
        const bool ax0 = !(rho > 1e-12f);
        const float ir = 1.f / fmaxf(rho, 1e-30f);
        const float ux = ax0 ? 1.f : dl[0] * ir, uy = ax0 ? 0.f : dl[1] * ir;
        const float rc = fminf(rho, h[0]), zc = fminf(fmaxf(dl[2], -h[2]), h[2]);
        float cl[3] = {ux * rc, uy * rc, zc};
        const float df[3] = {dl[0] - cl[0], dl[1] - cl[1], dl[2] - cl[2]};
        const float len = sqrtf(fmaf(df[0], df[0], fmaf(df[1], df[1], df[2] * df[2])));
        float nl[3], dist;
        if (len >= 1e-9f) { const float il = 1.f / len; nl[0] = df[0] * il; nl[1] = df[1] * il; nl[2] = df[2] * il; dist = len - sr; }
        else {
            const float er = h[0] - rho, ez = h[2] - fabsf(dl[2]);
            if (er <= ez) { nl[0] = ux; nl[1] = uy; nl[2] = 0.f; cl[0] = ux * h[0]; cl[1] = uy * h[0]; cl[2] = dl[2]; dist = -er - sr; }
            else { nl[0] = 0.f; nl[1] = 0.f; nl[2] = dl[2] >= 0.f ? 1.f : -1.f; cl[0] = dl[0]; cl[1] = dl[1]; cl[2] = nl[2] * h[2]; dist = -ez - sr; }
        }
        PBRE_UNROLL for (int k = 0; k < 3; k++) {
            n[k] = fmaf(R[3 * k], nl[0], fmaf(R[3 * k + 1], nl[1], R[3 * k + 2] * nl[2]));
            pb[k] = c[k] + fmaf(R[3 * k], cl[0], fmaf(R[3 * k + 1], cl[1], R[3 * k + 2] * cl[2]));
        }
        return dist;
    }
};

struct ObjStep {
    static constexpr int NK = 4;      // object-table contact slots (ShapeT::NC_OT of every shape)
    static PBRE_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
#if defined(__HIP_DEVICE_COMPILE__)
    static PBRE_HD float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
#else
    static PBRE_HD float med3(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
#endif
    // pose: position[3] | quaternion xyzw[4]; tw: linear[3] | angular[3] velocity at t.  o[6]: the twist at t + dt (after the
    // constraint solve, clamped to the velocity bound), which is all the caller needs to integrate the pose itself.
    static PBRE_HD void run(const Params& P, const float* pose, const float* tw, float* o) { run_p(P, pose, tw, o, P.obj_m, P.obj_mu, P.kl); }
    // o_m, o_mu, o_kl: this env's object mass / lateral friction / linear damping (pbre_set_physics_per_env; the principal inertias
    // P.obj_I scale with o_m / P.obj_m)
    static PBRE_HD void run_p(const Params& P, const float* pose, const float* tw, float* o, float o_m, float o_mu, float o_kl) {
        ObjStep s;
        s.setup(P, pose, tw, o_m, o_mu, o_kl);
        for (int it = 0; it < P.iters; it++) s.sweep();
        s.result(P, o);
    }

    // The same step in three parts, for a caller that runs the object rows inside its own solver loop (pbre_lane.hpp: one sweep per
    // iteration next to the robot's rows, so that the two dependency chains overlap): setup, P.iters x sweep, result.
    float vx, vy, vz, wx, wy, wz, mu;
    float Ii[6];       // m * I_w^-1 = m R diag(1/I) R^T: xx yy zz xy xz yz (a caller with robot-object rows needs it too)
    float c_rx[NK], c_ry[NK], c_rz[NK], g[NK][3][3], r_dinv[NK][3], r_app[NK][3], r_rhs[NK];
    float r_den[NK][3];      // 1 / r_dinv (0 for an unused slot): only sweep_res() reads it
    PBRE_HD void setup(const Params& P, const float* pose, const float* tw, float o_m, float o_mu, float o_kl) {
        const float dt = P.dt, inv_dt = P.inv_dt, vmax = P.vmax;
        const float isc = o_m / P.obj_m;
        const float px = pose[0], py = pose[1], pz = pose[2];
        const float x = pose[3], y = pose[4], z = pose[5], w = pose[6];
        float R[9];
        R[0] = 1.f - 2.f * (y*y + z*z); R[1] = 2.f * (x*y - w*z);       R[2] = 2.f * (x*z + w*y);
        R[3] = 2.f * (x*y + w*z);       R[4] = 1.f - 2.f * (x*x + z*z); R[5] = 2.f * (y*z - w*x);
        R[6] = 2.f * (x*z - w*y);       R[7] = 2.f * (y*z + w*x);       R[8] = 1.f - 2.f * (x*x + y*y);
        vx = tw[0]; vy = tw[1]; vz = tw[2]; wx = tw[3]; wy = tw[4]; wz = tw[5];
        {   // unconstrained velocity: gravity, linear / angular damping, gyroscopic torque  w x (I_w w)
            const float I0 = P.obj_I[0] * isc, I1 = P.obj_I[1] * isc, I2 = P.obj_I[2] * isc;
            const float lx = I0 * (R[0]*wx + R[3]*wy + R[6]*wz), ly = I1 * (R[1]*wx + R[4]*wy + R[7]*wz), lz = I2 * (R[2]*wx + R[5]*wy + R[8]*wz);
            const float Lx = R[0]*lx + R[1]*ly + R[2]*lz, Ly = R[3]*lx + R[4]*ly + R[5]*lz, Lz = R[6]*lx + R[7]*ly + R[8]*lz;
            const float sl = fmaf(o_kl, sqrtf(fmaf(vx, vx, fmaf(vy, vy, vz * vz))), o_kl);
            const float sa = fmaf(P.ka, sqrtf(fmaf(wx, wx, fmaf(wy, wy, wz * wz))), P.ka);
            const float tx = -(wy * Lz - wz * Ly) - Lx * sa, ty = -(wz * Lx - wx * Lz) - Ly * sa, tz = -(wx * Ly - wy * Lx) - Lz * sa;
            // I_w^-1 tq = R diag(1/I) R^T tq
            const float ex = (R[0]*tx + R[3]*ty + R[6]*tz) / I0, ey = (R[1]*tx + R[4]*ty + R[7]*tz) / I1, ez = (R[2]*tx + R[5]*ty + R[8]*tz) / I2;
            const float ax = R[0]*ex + R[1]*ey + R[2]*ez, ay = R[3]*ex + R[4]*ey + R[5]*ez, az = R[6]*ex + R[7]*ey + R[8]*ez;
            vx = clampf(fmaf(dt, -sl * vx, vx), -vmax, vmax); vy = clampf(fmaf(dt, -sl * vy, vy), -vmax, vmax);
            vz = clampf(fmaf(dt, P.gz - sl * vz, vz), -vmax, vmax);
            wx = clampf(fmaf(dt, ax, wx), -vmax, vmax); wy = clampf(fmaf(dt, ay, wy), -vmax, vmax); wz = clampf(fmaf(dt, az, wz), -vmax, vmax);
        }
        // object-table contacts: the (at most NK) box vertices closest to their support surface within the margin, in vertex order.
        // Impulses in delta-v units (a = lambda / m): a row along dir at lever arm r has J = [dir, r x dir], changes the twist by
        // (a dir, a g) with g = m I_w^-1 (r x dir), and J M^-1 J^T = (1 + (r x dir) . g) / m.
        mu = o_mu * P.tab_mu;
        {
            const float a = P.obj_m / P.obj_I[0], b = P.obj_m / P.obj_I[1], c = P.obj_m / P.obj_I[2];      // (m / I is independent of the per-env mass)
            Ii[0] = a*R[0]*R[0] + b*R[1]*R[1] + c*R[2]*R[2]; Ii[1] = a*R[3]*R[3] + b*R[4]*R[4] + c*R[5]*R[5]; Ii[2] = a*R[6]*R[6] + b*R[7]*R[7] + c*R[8]*R[8];
            Ii[3] = a*R[0]*R[3] + b*R[1]*R[4] + c*R[2]*R[5]; Ii[4] = a*R[0]*R[6] + b*R[1]*R[7] + c*R[2]*R[8]; Ii[5] = a*R[3]*R[6] + b*R[4]*R[7] + c*R[5]*R[8];
        }
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            c_rx[c] = c_ry[c] = c_rz[c] = 0.f; r_rhs[c] = 0.f;
            PBRE_UNROLL for (int d = 0; d < 3; d++) { r_dinv[c][d] = 0.f; r_den[c][d] = 0.f; r_app[c][d] = 0.f; g[c][d][0] = g[c][d][1] = g[c][d][2] = 0.f; }
        }
        {
            float vd[8], rx[8], ry[8], rz[8];
            const float top = P.tab_c[2] + P.tab_h[2], bot = P.tab_c[2] - P.tab_h[2];
            PBRE_UNROLL for (int v = 0; v < 8; v++) {
                bool used = true;
                if (P.obj_shape == 0) {
                    const float lx = (v & 1) ? P.obj_h[0] : -P.obj_h[0], ly = (v & 2) ? P.obj_h[1] : -P.obj_h[1], lz = (v & 4) ? P.obj_h[2] : -P.obj_h[2];
                    rx[v] = fmaf(R[0], lx, fmaf(R[1], ly, R[2] * lz)); ry[v] = fmaf(R[3], lx, fmaf(R[4], ly, R[5] * lz));
                    rz[v] = fmaf(R[6], lx, fmaf(R[7], ly, R[8] * lz));
                } else used = Shapes::candidate(P.obj_shape, P.obj_h, R, v, rx[v], ry[v], rz[v]);      // (wave-uniform branch: the shape is a batch constant)
                const float X = px + rx[v], Y = py + ry[v], Z = pz + rz[v];
                const bool in = fabsf(X - P.tab_c[0]) <= P.tab_h[0] && fabsf(Y - P.tab_c[1]) <= P.tab_h[1];
                vd[v] = used ? Z - ((in && Z > bot) ? top : P.ground_z) : 3e38f;
            }
            int slot = 0;
            PBRE_UNROLL for (int v = 0; v < 8; v++) {
                int r = 0;
                PBRE_UNROLL for (int u = 0; u < 8; u++) {
                    if (u == v) continue;
                    const bool before = vd[u] < vd[v] || (vd[u] == vd[v] && u < v);
                    r += (vd[u] < P.margin && before) ? 1 : 0;
                }
                if (vd[v] < P.margin && r < NK) {
                    PBRE_UNROLL for (int c = 0; c < NK; c++) if (slot == c) {
                        c_rx[c] = rx[v]; c_ry[c] = ry[v]; c_rz[c] = rz[v];
                        // r x dir for dir = +z (normal), -y, +x (btPlaneSpace1 of +z)
                        const float Ja[3][3] = {{ry[v], -rx[v], 0.f}, {rz[v], 0.f, -rx[v]}, {0.f, rz[v], -ry[v]}};
                        PBRE_UNROLL for (int d = 0; d < 3; d++) {
                            g[c][d][0] = Ii[0] * Ja[d][0] + Ii[3] * Ja[d][1] + Ii[4] * Ja[d][2];
                            g[c][d][1] = Ii[3] * Ja[d][0] + Ii[1] * Ja[d][1] + Ii[5] * Ja[d][2];
                            g[c][d][2] = Ii[4] * Ja[d][0] + Ii[5] * Ja[d][1] + Ii[2] * Ja[d][2];
                            r_den[c][d] = 1.f + Ja[d][0] * g[c][d][0] + Ja[d][1] * g[c][d][1] + Ja[d][2] * g[c][d][2];
                            r_dinv[c][d] = 1.f / r_den[c][d];
                        }
                        const float pen = vd[v] + P.slop;     // setupMultiBodyContactConstraint, restitution 0
                        r_rhs[c] = (pen > 0.f ? -pen * inv_dt : -pen * P.erp * inv_dt) * r_dinv[c][0];
                    }
                    slot++;
                }
            }
        }
    }
    // one sweep: normals then frictions; an unused slot has dinv = rhs = 0 and its rows change nothing
    PBRE_HD void sweep() { sweep_normals(); sweep_frictions(); }
    PBRE_HD void sweep_normals() {
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            // rows in delta form (clamp(applied + delta) - applied = clamp(delta, lo - applied, hi - applied)): one operation less
            // on the row-to-row dependency chain, which is all this kernel's time; the upper bound 1e10 of a normal row never binds
            const float jv = vz + c_ry[c] * wx - c_rx[c] * wy;
            const float dd = fmaxf(fmaf(-jv, r_dinv[c][0], r_rhs[c]), -r_app[c][0]);
            r_app[c][0] += dd;
            vz += dd; wx = fmaf(dd, g[c][0][0], wx); wy = fmaf(dd, g[c][0][1], wy); wz = fmaf(dd, g[c][0][2], wz);
        }
    }
    PBRE_HD void sweep_frictions() {
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            const float hi = mu * r_app[c][0];
            {
                const float jv = -vy + c_rz[c] * wx - c_rx[c] * wz;
                float dd = med3(-jv * r_dinv[c][1], -hi - r_app[c][1], hi - r_app[c][1]);
                dd = hi > 0.f ? dd : 0.f;
                r_app[c][1] += dd;
                vy -= dd; wx = fmaf(dd, g[c][1][0], wx); wy = fmaf(dd, g[c][1][1], wy); wz = fmaf(dd, g[c][1][2], wz);
            }
            {
                const float jv = vx + c_rz[c] * wy - c_ry[c] * wz;
                float dd = med3(-jv * r_dinv[c][2], -hi - r_app[c][2], hi - r_app[c][2]);
                dd = hi > 0.f ? dd : 0.f;
                r_app[c][2] += dd;
                vx += dd; wx = fmaf(dd, g[c][2][0], wx); wy = fmaf(dd, g[c][2][1], wy); wz = fmaf(dd, g[c][2][2], wz);
            }
        }
    }
    // one sweep as sweep(), returning the largest |delta impulse / jacDiagABInv| of its rows: the object's contribution to Bullet's
    // least-squares residual (pbre_physics.solver_residual_threshold; callers compare against Params::res_lim = its square root)
    PBRE_HD float sweep_res() {
        float lsr = 0.f;
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            const float jv = vz + c_ry[c] * wx - c_rx[c] * wy;
            const float dd = fmaxf(fmaf(-jv, r_dinv[c][0], r_rhs[c]), -r_app[c][0]);
            r_app[c][0] += dd;
            lsr = fmaxf(lsr, fabsf(dd * r_den[c][0]));
            vz += dd; wx = fmaf(dd, g[c][0][0], wx); wy = fmaf(dd, g[c][0][1], wy); wz = fmaf(dd, g[c][0][2], wz);
        }
        PBRE_UNROLL for (int c = 0; c < NK; c++) {
            const float hi = mu * r_app[c][0];
            {
                const float jv = -vy + c_rz[c] * wx - c_rx[c] * wz;
                float dd = med3(-jv * r_dinv[c][1], -hi - r_app[c][1], hi - r_app[c][1]);
                dd = hi > 0.f ? dd : 0.f;
                r_app[c][1] += dd;
                lsr = fmaxf(lsr, fabsf(dd * r_den[c][1]));
                vy -= dd; wx = fmaf(dd, g[c][1][0], wx); wy = fmaf(dd, g[c][1][1], wy); wz = fmaf(dd, g[c][1][2], wz);
            }
            {
                const float jv = vx + c_rz[c] * wy - c_ry[c] * wz;
                float dd = med3(-jv * r_dinv[c][2], -hi - r_app[c][2], hi - r_app[c][2]);
                dd = hi > 0.f ? dd : 0.f;
                r_app[c][2] += dd;
                lsr = fmaxf(lsr, fabsf(dd * r_den[c][2]));
                vx += dd; wx = fmaf(dd, g[c][2][0], wx); wy = fmaf(dd, g[c][2][1], wy); wz = fmaf(dd, g[c][2][2], wz);
            }
        }
        return lsr;
    }
    PBRE_HD void result(const Params& P, float* o) const {
        const float vmax = P.vmax;
        o[0] = clampf(vx, -vmax, vmax); o[1] = clampf(vy, -vmax, vmax); o[2] = clampf(vz, -vmax, vmax);
        o[3] = clampf(wx, -vmax, vmax); o[4] = clampf(wy, -vmax, vmax); o[5] = clampf(wz, -vmax, vmax);
    }
};

}  // namespace pbre
