/* pbre_oracle.c -- TEST INFRASTRUCTURE ONLY (see pbre_oracle.h header comment).
 *
 * PARITY UNPINNED for physics: restates Bullet3's btMultiBodyDynamicsWorld step
 * (third-party `pybullet`, unpinned in reference requirements.txt:2; not present in
 * /root/reference) from its published algorithm.  Call sites restated:
 *   p.stepSimulation                 panda_push_gym_env.py:236 (+133,140,148), panda_reach_gym_env.py:220
 *   p.setJointMotorControl2          panda_env.py:74-77 (reset gains), :305-310 (step gains)
 *   p.getLinkState/getJointStates    panda_env.py:147,187
 *   p.getBasePositionAndOrientation  world_env.py:114
 *   p.getEulerFromQuaternion / getQuaternionFromEuler / invertTransform / multiplyTransforms
 *                                    panda_push_gym_env.py:168-174
 * Glue restated (pinned by tests/golden): panda_push_gym_env.py:105-360,
 * panda_reach_gym_env.py:105-313, world_env.py:145-176, utils.py:11-14.
 *
 * Formulation: Featherstone articulated-body algorithm in link coordinates
 * (spatial vectors [angular; linear]), one impulse-response ABA pass per
 * constraint row (Bullet: calcAccelerationDeltasMultiDof), velocity-level
 * constraint rows, projected Gauss-Seidel in Bullet's row order, semi-implicit
 * Euler.  The device path (csrc/) uses a different but mathematically
 * equivalent formulation (world-frame RNEA + CRBA + explicit inverse), so
 * agreement between the two is a genuine cross-check.
 */
#include "pbre_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

#define PI 3.14159265358979323846

/* ------------------------------------------------------------------ small math */
static void m3_mul(const real* A, const real* B, real* C) {
    real t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        real s = 0; for (int k = 0; k < 3; k++) s += A[i*3+k] * B[k*3+j];
        t[i*3+j] = s;
    }
    memcpy(C, t, sizeof t);
}
static void m3_T(const real* A, real* B) {
    real t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[i*3+j] = A[j*3+i];
    memcpy(B, t, sizeof t);
}
static void m3_v(const real* A, const real* v, real* o) {
    real t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i*3]*v[0] + A[i*3+1]*v[1] + A[i*3+2]*v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void m3T_v(const real* A, const real* v, real* o) {
    real t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i]*v[0] + A[3+i]*v[1] + A[6+i]*v[2];
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void cross(const real* a, const real* b, real* o) {
    real t0 = a[1]*b[2] - a[2]*b[1], t1 = a[2]*b[0] - a[0]*b[2], t2 = a[0]*b[1] - a[1]*b[0];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static real dot3(const real* a, const real* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static real norm3(const real* a) { return (real)sqrt((double)dot3(a, a)); }
static void axis_angle(const real* a, real th, real* R) { /* Rodrigues */
    real c = (real)cos((double)th), s = (real)sin((double)th), C = 1 - c;
    R[0] = c + a[0]*a[0]*C;      R[1] = a[0]*a[1]*C - a[2]*s; R[2] = a[0]*a[2]*C + a[1]*s;
    R[3] = a[1]*a[0]*C + a[2]*s; R[4] = c + a[1]*a[1]*C;      R[5] = a[1]*a[2]*C - a[0]*s;
    R[6] = a[2]*a[0]*C - a[1]*s; R[7] = a[2]*a[1]*C + a[0]*s; R[8] = c + a[2]*a[2]*C;
}
static void quat_to_R(const real* q, real* R) { /* (x,y,z,w), local->world */
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2*(y*y + z*z); R[1] = 2*(x*y - w*z);     R[2] = 2*(x*z + w*y);
    R[3] = 2*(x*y + w*z);     R[4] = 1 - 2*(x*x + z*z); R[5] = 2*(y*z - w*x);
    R[6] = 2*(x*z - w*y);     R[7] = 2*(y*z + w*x);     R[8] = 1 - 2*(x*x + y*y);
}
static void quat_mul(const real* a, const real* b, real* o) {
    real t[4];
    t[0] = a[3]*b[0] + a[0]*b[3] + a[1]*b[2] - a[2]*b[1];
    t[1] = a[3]*b[1] - a[0]*b[2] + a[1]*b[3] + a[2]*b[0];
    t[2] = a[3]*b[2] + a[0]*b[1] - a[1]*b[0] + a[2]*b[3];
    t[3] = a[3]*b[3] - a[0]*b[0] - a[1]*b[1] - a[2]*b[2];
    memcpy(o, t, sizeof t);
}
/* btMatrix3x3::getRotation [EXT-UNVERIFIED, standard Shepperd branches] */
static void R_to_quat(const real* R, real* q) {
    real tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        real s = (real)sqrt((double)(tr + 1));
        q[3] = s * (real)0.5; s = (real)0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = R[0] < R[4] ? (R[4] < R[8] ? 2 : 1) : (R[0] < R[8] ? 2 : 0);
        int j = (i + 1) % 3, k = (i + 2) % 3;
        real s = (real)sqrt((double)(R[i*3+i] - R[j*3+j] - R[k*3+k] + 1));
        real t[4];
        t[i] = s * (real)0.5; s = (real)0.5 / s;
        t[3] = (R[k*3+j] - R[j*3+k]) * s;
        t[j] = (R[j*3+i] + R[i*3+j]) * s;
        t[k] = (R[k*3+i] + R[i*3+k]) * s;
        memcpy(q, t, sizeof t);
    }
}
/* pybullet.getQuaternionFromEuler / getEulerFromQuaternion [EXT-UNVERIFIED] */
void orc_quat_from_euler(const real e[3], real q[4]) {
    double hr = e[0] * 0.5, hp = e[1] * 0.5, hy = e[2] * 0.5;
    double cr = cos(hr), sr = sin(hr), cp = cos(hp), sp = sin(hp), cy = cos(hy), sy = sin(hy);
    q[0] = (real)(sr*cp*cy - cr*sp*sy);
    q[1] = (real)(cr*sp*cy + sr*cp*sy);
    q[2] = (real)(cr*cp*sy - sr*sp*cy);
    q[3] = (real)(cr*cp*cy + sr*sp*sy);
}
void orc_euler_from_quat(const real q[4], real e[3]) {
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double sqx = x*x, sqy = y*y, sqz = z*z, squ = w*w;
    double sarg = -2.0 * (x*z - w*y);
    if (sarg <= -0.99999) { e[0] = 0; e[1] = (real)(-0.5*PI); e[2] = (real)(2*atan2(x, -y)); }
    else if (sarg >= 0.99999) { e[0] = 0; e[1] = (real)(0.5*PI); e[2] = (real)(2*atan2(-x, y)); }
    else {
        e[0] = (real)atan2(2*(y*z + w*x), squ - sqx - sqy + sqz);
        e[1] = (real)asin(sarg);
        e[2] = (real)atan2(2*(x*y + w*z), squ + sqx - sqy - sqz);
    }
}

/* ------------------------------------------------------------------ model */
int orc_sizeof_real(void) { return (int)sizeof(real); }
/* state record layout (pbre_oracle.h) */
static int lay_w(const orc_model* m) { return m->ndof <= 9 ? (m->ntip > 0 ? 32 : 16) : (m->ndof <= 20 ? 32 : (m->ndof <= 32 ? 64 : 128)); }   /* a <= 9-DoF robot with fingertip statistics (pandaEnv alone) needs the 32-lane record */
int orc_state_floats(const orc_model* m) { return 2 * lay_w(m) + 16; }
#define OQ(m) ((m)->ndof)                 /* object position (then quaternion) inside Q */
#define OV(m) (lay_w(m))                  /* V record */
#define OX(m) (2 * lay_w(m))              /* X record */

int orc_model_from_table(const double* t, size_t n, orc_model* m) {
    memset(m, 0, sizeof *m);
    if (n < 24 || t[0] != 1346523717.0 || t[1] != 1.0) return -1;
    m->nl = (int)t[2]; m->ndof = (int)t[3]; m->ee_link = (int)t[4]; m->ns = (int)t[5];
    if (m->nl > ORC_MAXL || m->ndof > ORC_MAXD || m->ns > ORC_MAXS) return -2;
    if (n < (size_t)(24 + m->nl * 40 + m->ns * 8)) return -3;
    for (int i = 0; i < 3; i++) m->base_pos[i] = (real)t[6+i];
    for (int i = 0; i < 9; i++) m->base_R[i] = (real)t[9+i];
    m->fixed_base = (int)t[18];
    for (int i = 0; i < m->nl; i++) {
        const double* r = t + 24 + i * 40;
        m->parent[i] = (int)r[0]; m->jtype[i] = (int)r[1];
        for (int k = 0; k < 3; k++) { m->axis[i][k] = (real)r[2+k]; m->Xp[i][k] = (real)r[5+k]; m->com[i][k] = (real)r[18+k]; }
        for (int k = 0; k < 9; k++) { m->XR[i][k] = (real)r[8+k]; m->inertia[i][k] = (real)r[21+k]; }
        m->mass[i] = (real)r[17]; m->lower[i] = (real)r[30]; m->upper[i] = (real)r[31];
        m->damping[i] = (real)r[32]; m->dof[i] = (int)r[33]; m->friction[i] = (real)r[34]; m->passive[i] = (int)r[37];
        m->max_force[i] = (m->passive[i] && r[35] > 0) ? (real)r[35] : (real)0;
        if (m->dof[i] >= 0) m->link_of_dof[m->dof[i]] = i;
    }
    for (int k = 0; k < m->ns; k++) {
        const double* s = t + 24 + m->nl * 40 + k * 8;
        m->s_link[k] = (int)s[0];
        for (int c = 0; c < 3; c++) m->s_c[k][c] = (real)s[1+c];
        m->s_r[k] = (real)s[4]; m->s_mu[k] = (real)s[5];
        m->s_tip[k] = (int)s[6];
        if (m->s_tip[k] > ORC_NTIP) return -4;
        if (m->s_tip[k] > 0) m->ntip = ORC_NTIP;
    }
    return 0;
}

void orc_default_params(orc_params* p) {
    memset(p, 0, sizeof *p);
    p->dt = 1.0 / 240.0;            /* panda_push_gym_env.py:39 */
    p->gravity_z = -9.8;            /* :126 */
    p->solver_iters = 150;          /* :122 */
    p->solver_residual_threshold = 0;   /* all sweeps (pbre_oracle.h); PyBullet's documented default would be 1e-7 [EXT-UNVERIFIED] */
    p->erp = 0.2;                   /* Bullet contact/joint ERP default [EXT-UNVERIFIED] */
    p->linear_slop = 1e-5;          /* [EXT-UNVERIFIED] */
    p->contact_margin = 1e-3;       /* ~ contact breaking threshold of a 5 cm cube [EXT-UNVERIFIED] */
    p->lin_damping = 0.04; p->ang_damping = 0.04;   /* btMultiBody defaults [EXT-UNVERIFIED] */
    p->max_coord_vel = 100.0;       /* btMultiBody::m_maxCoordinateVelocity [EXT-UNVERIFIED] */
    p->max_motor_impulse = 100000.0 / 240.0;  /* pybullet default force 1e5 * dt [EXT-UNVERIFIED] */
    p->limit_max_impulse = 100.0;   /* btMultiBodyConstraint default [EXT-UNVERIFIED] */
    /* pybullet_data/table/table.urdf at (0.85,0,0): top box 1.5 x 1.0 x 0.05 centred z=0.6 */
    p->table_c[0] = 0.85; p->table_c[1] = 0.0; p->table_c[2] = 0.6;
    p->table_h[0] = 0.75; p->table_h[1] = 0.5; p->table_h[2] = 0.025;
    p->table_mu = 0.5; p->ground_z = 0.0;
    /* cube_small.urdf: 0.05 m box, 0.1 kg, lateral friction 1.0, inertia from shape */
    p->obj_h[0] = p->obj_h[1] = p->obj_h[2] = 0.025;
    p->obj_mass = 0.1;
    p->obj_inertia[0] = p->obj_inertia[1] = p->obj_inertia[2] = 0.1 * (0.05*0.05 + 0.05*0.05) / 12.0;
    p->obj_mu = 1.0;
    p->flags = 0;
}

/* ------------------------------------------------------------------ FK */
void orc_fk(const orc_model* m, const real* q, real* R, real* p) {
    for (int i = 0; i < m->nl; i++) {
        const real* Rp = m->parent[i] < 0 ? m->base_R : R + 9 * m->parent[i];
        const real* pp = m->parent[i] < 0 ? m->base_pos : p + 3 * m->parent[i];
        real Rj[9], t[3], tmp[9];
        real qi = m->dof[i] >= 0 ? q[m->dof[i]] : 0;
        m3_mul(Rp, m->XR[i], tmp);
        if (m->jtype[i] == 1) { axis_angle(m->axis[i], qi, Rj); m3_mul(tmp, Rj, R + 9*i); }
        else memcpy(R + 9*i, tmp, sizeof tmp);
        m3_v(Rp, m->Xp[i], t);
        for (int k = 0; k < 3; k++) p[3*i+k] = pp[k] + t[k];
        if (m->jtype[i] == 2) {
            real aw[3]; m3_v(R + 9*i, m->axis[i], aw);
            for (int k = 0; k < 3; k++) p[3*i+k] += aw[k] * qi;
        }
    }
}

/* ------------------------------------------------------------------ spatial algebra */
typedef struct {
    real E[9], r[3];          /* parent->link rotation, link origin in parent coords */
    real S[6];                /* joint motion subspace in link coords */
    real v[6], c[6];          /* spatial velocity, velocity-product acceleration */
    real I[36];               /* rigid-body spatial inertia at link origin */
    real IA[36], U[6], d;     /* articulated inertia, IA*S, S.U */
    real pA[6], u, a[6];
} lws;

typedef struct {
    lws L[ORC_MAXL];
    real Rw[ORC_MAXL][9], pw[ORC_MAXL][3];
} aba_ws;

static void xm(const lws* L, const real* in, real* out) { /* motion transform parent->link */
    real t[3], w[3], v[3];
    cross(L->r, in, t);                       /* r x w */
    m3_v(L->E, in, w);
    t[0] = in[3] - t[0]; t[1] = in[4] - t[1]; t[2] = in[5] - t[2];
    m3_v(L->E, t, v);
    out[0] = w[0]; out[1] = w[1]; out[2] = w[2]; out[3] = v[0]; out[4] = v[1]; out[5] = v[2];
}
static void xf_T(const lws* L, const real* in, real* out) { /* force transform link->parent (X^T) */
    real n[3], f[3], t[3];
    m3T_v(L->E, in, n); m3T_v(L->E, in + 3, f);
    cross(L->r, f, t);
    out[0] = n[0] + t[0]; out[1] = n[1] + t[1]; out[2] = n[2] + t[2];
    out[3] = f[0]; out[4] = f[1]; out[5] = f[2];
}
static void crm(const real* v, const real* m_, real* o) { /* v x m (motion) */
    real a[3], b[3], c[3];
    cross(v, m_, a); cross(v, m_ + 3, b); cross(v + 3, m_, c);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
static void crf(const real* v, const real* f, real* o) { /* v x* f (force) */
    real a[3], b[3], c[3];
    cross(v, f, a); cross(v + 3, f + 3, b); cross(v, f + 3, c);
    o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
static void m6_v(const real* A, const real* v, real* o) {
    real t[6];
    for (int i = 0; i < 6; i++) { real s = 0; for (int k = 0; k < 6; k++) s += A[i*6+k] * v[k]; t[i] = s; }
    memcpy(o, t, sizeof t);
}
static void build_X(const lws* L, real* X) { /* 6x6 motion transform */
    real rx[9] = {0, -L->r[2], L->r[1], L->r[2], 0, -L->r[0], -L->r[1], L->r[0], 0};
    real Erx[9]; m3_mul(L->E, rx, Erx);
    memset(X, 0, 36 * sizeof(real));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        X[i*6+j] = L->E[i*3+j]; X[(i+3)*6+(j+3)] = L->E[i*3+j]; X[(i+3)*6+j] = -Erx[i*3+j];
    }
}
static void spatial_inertia(real mass, const real* com, const real* Ic, real* I) {
    real cx[9] = {0, -com[2], com[1], com[2], 0, -com[0], -com[1], com[0], 0};
    real cxT[9], cc[9];
    m3_T(cx, cxT); m3_mul(cx, cxT, cc);
    memset(I, 0, 36 * sizeof(real));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        I[i*6+j] = Ic[i*3+j] + mass * cc[i*3+j];
        I[i*6+3+j] = mass * cx[i*3+j];
        I[(i+3)*6+j] = mass * cxT[i*3+j];
    }
    I[21] = I[28] = I[35] = mass;
}

/* configuration-dependent part: transforms, articulated inertias (pass 2 without forces) */
static void aba_config_d(const orc_model* m, const real* q, aba_ws* w, real jd_dt);
static void aba_config(const orc_model* m, const real* q, aba_ws* w) { aba_config_d(m, q, w, 0); }
/* jd_dt > 0: implicit joint damping -- the joint-space inertia every pass divides by gets dt * damping added (an armature term) */
static void aba_config_d(const orc_model* m, const real* q, aba_ws* w, real jd_dt) {
    orc_fk(m, q, &w->Rw[0][0], &w->pw[0][0]);
    for (int i = 0; i < m->nl; i++) {
        lws* L = &w->L[i];
        real qi = m->dof[i] >= 0 ? q[m->dof[i]] : 0;
        real Rloc[9];
        if (m->jtype[i] == 1) { real Rj[9]; axis_angle(m->axis[i], qi, Rj); m3_mul(m->XR[i], Rj, Rloc); }
        else memcpy(Rloc, m->XR[i], sizeof Rloc);
        m3_T(Rloc, L->E);
        for (int k = 0; k < 3; k++) L->r[k] = m->Xp[i][k];
        if (m->jtype[i] == 2) { real ap[3]; m3_v(m->XR[i], m->axis[i], ap); for (int k = 0; k < 3; k++) L->r[k] += ap[k] * qi; }
        memset(L->S, 0, sizeof L->S);
        if (m->jtype[i] == 1) for (int k = 0; k < 3; k++) L->S[k] = m->axis[i][k];
        if (m->jtype[i] == 2) for (int k = 0; k < 3; k++) L->S[3+k] = m->axis[i][k];
        spatial_inertia(m->mass[i], m->com[i], m->inertia[i], L->I);
        memcpy(L->IA, L->I, sizeof L->IA);
    }
    for (int i = m->nl - 1; i >= 0; i--) {
        lws* L = &w->L[i];
        real Ia[36];
        memcpy(Ia, L->IA, sizeof Ia);
        if (m->jtype[i] != 0) {
            m6_v(L->IA, L->S, L->U);
            L->d = jd_dt * m->damping[i]; for (int k = 0; k < 6; k++) L->d += L->S[k] * L->U[k];
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Ia[a*6+b] -= L->U[a] * L->U[b] / L->d;
        } else { memset(L->U, 0, sizeof L->U); L->d = 1; }
        if (m->parent[i] >= 0) {
            real X[36], T[36];
            build_X(L, X);
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { real s = 0; for (int k = 0; k < 6; k++) s += Ia[a*6+k] * X[k*6+b]; T[a*6+b] = s; }
            lws* P = &w->L[m->parent[i]];
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { real s = 0; for (int k = 0; k < 6; k++) s += X[k*6+a] * T[k*6+b]; P->IA[a*6+b] += s; }
        }
    }
}

/* force-dependent passes.  pA0[i]: bias force per link (link coords) or NULL; abase: base spatial accel */
static void aba_solve(const orc_model* m, aba_ws* w, const real* tau, int use_vel, const real (*pA0)[6],
                      const real* abase, real* qdd) {
    for (int i = 0; i < m->nl; i++) {
        lws* L = &w->L[i];
        if (pA0) memcpy(L->pA, pA0[i], sizeof L->pA); else memset(L->pA, 0, sizeof L->pA);
        if (!use_vel) memset(L->c, 0, sizeof L->c);
    }
    for (int i = m->nl - 1; i >= 0; i--) {
        lws* L = &w->L[i];
        real pa[6], Iac[6];
        m6_v(L->IA, L->c, Iac);
        if (m->jtype[i] != 0) {
            real Sp = 0; for (int k = 0; k < 6; k++) Sp += L->S[k] * L->pA[k];
            L->u = tau[m->dof[i]] - Sp;
            /* Ia c = IA c - U (U.c)/d */
            real Uc = 0; for (int k = 0; k < 6; k++) Uc += L->U[k] * L->c[k];
            for (int k = 0; k < 6; k++) pa[k] = L->pA[k] + Iac[k] - L->U[k] * Uc / L->d + L->U[k] * L->u / L->d;
        } else {
            for (int k = 0; k < 6; k++) pa[k] = L->pA[k] + Iac[k];
        }
        if (m->parent[i] >= 0) {
            real t[6]; xf_T(L, pa, t);
            for (int k = 0; k < 6; k++) w->L[m->parent[i]].pA[k] += t[k];
        }
    }
    for (int i = 0; i < m->nl; i++) {
        lws* L = &w->L[i];
        real ap[6];
        if (m->parent[i] >= 0) xm(L, w->L[m->parent[i]].a, ap); else xm(L, abase, ap);
        for (int k = 0; k < 6; k++) ap[k] += L->c[k];
        if (m->jtype[i] != 0) {
            real Ua = 0; for (int k = 0; k < 6; k++) Ua += L->U[k] * ap[k];
            real qd2 = (L->u - Ua) / L->d;
            qdd[m->dof[i]] = qd2;
            for (int k = 0; k < 6; k++) L->a[k] = ap[k] + L->S[k] * qd2;
        } else memcpy(L->a, ap, sizeof ap);
    }
}

/* velocity pass + bias forces: gyroscopic, Bullet's per-link velocity damping
 * f = m v_c (K + K|v_c|), n = I w (K + K|w|) [EXT-UNVERIFIED: btMultiBody.cpp DAMPING_K1/K2 = m_linearDamping] */
static void aba_velocity(const orc_model* m, const orc_params* prm, const real* qd, aba_ws* w, real (*pA0)[6]) {
    for (int i = 0; i < m->nl; i++) {
        lws* L = &w->L[i];
        real vp[6] = {0, 0, 0, 0, 0, 0}, vj[6];
        if (m->parent[i] >= 0) xm(L, w->L[m->parent[i]].v, vp); else { real z[6] = {0,0,0,0,0,0}; xm(L, z, vp); }
        real qdi = m->dof[i] >= 0 ? qd[m->dof[i]] : 0;
        for (int k = 0; k < 6; k++) { vj[k] = L->S[k] * qdi; L->v[k] = vp[k] + vj[k]; }
        crm(L->v, vj, L->c);
        real Iv[6]; m6_v(L->I, L->v, Iv);
        crf(L->v, Iv, pA0[i]);
        /* damping on the COM velocity */
        real vc[3], t[3];
        cross(L->v, m->com[i], t);
        for (int k = 0; k < 3; k++) vc[k] = L->v[3+k] + t[k];
        real kl = (real)prm->lin_damping, ka = (real)prm->ang_damping;
        real fl[3], na[3], Iw[3];
        real sl = m->mass[i] * (kl + kl * norm3(vc));
        for (int k = 0; k < 3; k++) fl[k] = sl * vc[k];
        m3_v(m->inertia[i], L->v, Iw);
        real sa = ka + ka * norm3(L->v);
        for (int k = 0; k < 3; k++) na[k] = sa * Iw[k];
        cross(m->com[i], fl, t);
        for (int k = 0; k < 3; k++) { pA0[i][k] += na[k] + t[k]; pA0[i][3+k] += fl[k]; }
    }
}

void orc_forward_dynamics(const orc_model* m, const orc_params* prm, const real* q, const real* qd,
                          const real* tau_in, real* qdd) {
    aba_ws* w = (aba_ws*)malloc(sizeof *w);
    real pA0[ORC_MAXL][6], tau[ORC_MAXD];
    aba_config(m, q, w);
    aba_velocity(m, prm, qd, w, pA0);
    for (int i = 0; i < m->nl; i++) if (m->dof[i] >= 0)
        tau[m->dof[i]] = (tau_in ? tau_in[m->dof[i]] : 0) - m->damping[i] * qd[m->dof[i]];
    real gb[3] = {0, 0, (real)(-prm->gravity_z)}, ab[6] = {0, 0, 0, 0, 0, 0};
    m3T_v(m->base_R, gb, ab + 3);
    aba_solve(m, w, tau, 1, pA0, ab, qdd);
    free(w);
}

void orc_mass_matrix_inverse(const orc_model* m, const orc_params* prm, const real* q, real* Minv) {
    (void)prm;
    aba_ws* w = (aba_ws*)malloc(sizeof *w);
    aba_config(m, q, w);
    real ab[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < m->ndof; j++) {
        real tau[ORC_MAXD] = {0}, col[ORC_MAXD];
        tau[j] = 1;
        aba_solve(m, w, tau, 0, NULL, ab, col);
        for (int i = 0; i < m->ndof; i++) Minv[i * m->ndof + j] = col[i];
    }
    free(w);
}

/* ------------------------------------------------------------------ collision */
/* sphere vs oriented box.  Returns signed distance; n = world normal from box to sphere; pb = point on box */
static real sphere_box(const real* sc, real sr, const real* bc, const real* Rb, const real* h, real* n, real* pb) {
    real d[3], dl[3], cl[3], nl[3];
    for (int k = 0; k < 3; k++) d[k] = sc[k] - bc[k];
    m3T_v(Rb, d, dl);
    int inside = 1;
    for (int k = 0; k < 3; k++) {
        cl[k] = dl[k] < -h[k] ? -h[k] : (dl[k] > h[k] ? h[k] : dl[k]);
        if (cl[k] != dl[k]) inside = 0;
    }
    real dist;
    real df[3] = {dl[0] - cl[0], dl[1] - cl[1], dl[2] - cl[2]};
    real len = norm3(df);
    if (len < (real)1e-9) inside = 1;   /* centre on/inside the surface: use the face-normal branch */
    if (!inside) {
        for (int k = 0; k < 3; k++) nl[k] = df[k] / len;
        dist = len - sr;
    } else {
        int ax = 0; real best = h[0] - (real)fabs((double)dl[0]);
        for (int k = 1; k < 3; k++) { real e = h[k] - (real)fabs((double)dl[k]); if (e < best) { best = e; ax = k; } }
        nl[0] = nl[1] = nl[2] = 0; nl[ax] = dl[ax] >= 0 ? (real)1 : (real)-1;
        cl[ax] = nl[ax] * h[ax];
        dist = -best - sr;
    }
    m3_v(Rb, nl, n);
    real t[3]; m3_v(Rb, cl, t);
    for (int k = 0; k < 3; k++) pb[k] = bc[k] + t[k];
    return dist;
}

/* sphere vs the object primitive (orc_params.obj_shape; stand-ins of the reference's round objects -- YcbTennisBall, the cans, pear,
 * duck_vhacd: world_env.py:18-25, 179-216).  Same contract as sphere_box: signed distance, n = world normal object -> sphere, pb = point
 * on the object.  shape 1: sphere of radius h[0]; shape 2: capped cylinder about the object's local z, radius h[0], half height h[2]. */
static real sphere_shape(int shape, const real* sc, real sr, const real* bc, const real* Rb, const real* h, real* n, real* pb) {
    if (shape == 0) return sphere_box(sc, sr, bc, Rb, h, n, pb);
    real d[3];
    for (int k = 0; k < 3; k++) d[k] = sc[k] - bc[k];
    if (shape == 1) {
        real len = norm3(d);
        if (len < (real)1e-9) { n[0] = 0; n[1] = 0; n[2] = 1; } else for (int k = 0; k < 3; k++) n[k] = d[k] / len;
        for (int k = 0; k < 3; k++) pb[k] = bc[k] + n[k] * h[0];
        return len - h[0] - sr;
    }
    real dl[3], cl[3], nl[3], dist;
    m3T_v(Rb, d, dl);
    const real rho = (real)sqrt((double)(dl[0] * dl[0] + dl[1] * dl[1]));
    const real ux = rho > (real)1e-12 ? dl[0] / rho : 1, uy = rho > (real)1e-12 ? dl[1] / rho : 0;      /* radial unit vector */
    const real rc = rho > h[0] ? h[0] : rho, zc = dl[2] < -h[2] ? -h[2] : (dl[2] > h[2] ? h[2] : dl[2]);
    cl[0] = ux * rc; cl[1] = uy * rc; cl[2] = zc;
    real df[3] = {dl[0] - cl[0], dl[1] - cl[1], dl[2] - cl[2]};
    real len = norm3(df);
    if (len >= (real)1e-9) {                            /* centre outside the solid */
        for (int k = 0; k < 3; k++) nl[k] = df[k] / len;
        dist = len - sr;
    } else {                                            /* inside: leave through the nearer of the lateral surface and the caps */
        const real er = h[0] - rho, ez = h[2] - (real)fabs((double)dl[2]);
        if (er <= ez) { nl[0] = ux; nl[1] = uy; nl[2] = 0; cl[0] = ux * h[0]; cl[1] = uy * h[0]; cl[2] = dl[2]; dist = -er - sr; }
        else { nl[0] = 0; nl[1] = 0; nl[2] = dl[2] >= 0 ? (real)1 : (real)-1; cl[0] = dl[0]; cl[1] = dl[1]; cl[2] = nl[2] * h[2]; dist = -ez - sr; }
    }
    m3_v(Rb, nl, n);
    real t[3]; m3_v(Rb, cl, t);
    for (int k = 0; k < 3; k++) pb[k] = bc[k] + t[k];
    return dist;
}

/* ---- convex-hull object (obj_shape 3; SURVEY 8(f4): duck_vhacd / teddy_vhacd / the YCB meshes of world_env.py:18-25, 179-216 are convex
 * decompositions whose pieces Bullet collides as btConvexHullShape).  The hull is given by its vertices alone.  This restatement is
 * deliberately the brute-force definition, not the engine's face table: the closest point of the hull to an outside point is the
 * closest point of the nearest triangle spanned by ANY three hull vertices (every such triangle lies inside the hull, and the surface
 * is made of such triangles); a point is inside iff it is behind every SUPPORTING plane (a plane through three vertices with all other
 * vertices on one side), and then leaves through the nearest of them -- the same inside rule as the box and the cylinder above.
 * The supporting planes are found once per vertex set (O(n^4)) and cached per thread. */
typedef struct { int n; double v[ORC_MAXHV][3]; int np; double (*pl)[4]; } hull_cache_t;
static __thread hull_cache_t g_hull;
static void hull_planes(const orc_params* prm) {
    const int n = prm->obj_hull_n;
    if (g_hull.n == n && g_hull.pl && memcmp(g_hull.v, prm->obj_hull, sizeof(double) * 3 * n) == 0) return;
    free(g_hull.pl);
    g_hull.n = n; memcpy(g_hull.v, prm->obj_hull, sizeof(double) * 3 * n);
    g_hull.pl = (double (*)[4])malloc(sizeof(double) * 4 * (size_t)(n * n * n / 6 + 8));
    g_hull.np = 0;
    double scale = 0;
    for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) if (fabs(prm->obj_hull[i][k]) > scale) scale = fabs(prm->obj_hull[i][k]);
    const double eps = 1e-9 * (scale > 0 ? scale : 1);
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) for (int k = j + 1; k < n; k++) {
        const double* a = prm->obj_hull[i]; const double* b = prm->obj_hull[j]; const double* c = prm->obj_hull[k];
        const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        double nn[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
        const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
        if (len < 1e-12 * (scale > 0 ? scale * scale : 1)) continue;        /* collinear */
        for (int t = 0; t < 3; t++) nn[t] /= len;
        int pos = 0, neg = 0;
        for (int l = 0; l < n; l++) {
            const double d = nn[0] * (prm->obj_hull[l][0] - a[0]) + nn[1] * (prm->obj_hull[l][1] - a[1]) + nn[2] * (prm->obj_hull[l][2] - a[2]);
            if (d > eps) pos++; else if (d < -eps) neg++;
        }
        if (pos && neg) continue;                                          /* cuts through the hull */
        const double sg = pos ? -1.0 : 1.0;                                  /* outward: every other vertex behind the plane */
        double* q = g_hull.pl[g_hull.np++];
        q[0] = sg * nn[0]; q[1] = sg * nn[1]; q[2] = sg * nn[2]; q[3] = q[0] * a[0] + q[1] * a[1] + q[2] * a[2];
    }
}
/* closest point of triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5; degenerate triangles fall through to their edges) */
static void closest_on_triangle(const double* p, const double* a, const double* b, const double* c, double* out) {
    double ab[3], ac[3], ap[3];
    for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = p[k] - a[k]; }
#define D3(x, y) ((x)[0] * (y)[0] + (x)[1] * (y)[1] + (x)[2] * (y)[2])
    const double d1 = D3(ab, ap), d2 = D3(ac, ap);
    if (d1 <= 0 && d2 <= 0) { for (int k = 0; k < 3; k++) out[k] = a[k]; return; }
    double bp[3]; for (int k = 0; k < 3; k++) bp[k] = p[k] - b[k];
    const double d3 = D3(ab, bp), d4 = D3(ac, bp);
    if (d3 >= 0 && d4 <= d3) { for (int k = 0; k < 3; k++) out[k] = b[k]; return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); for (int k = 0; k < 3; k++) out[k] = a[k] + v * ab[k]; return; }
    double cp[3]; for (int k = 0; k < 3; k++) cp[k] = p[k] - c[k];
    const double d5 = D3(ab, cp), d6 = D3(ac, cp);
    if (d6 >= 0 && d5 <= d6) { for (int k = 0; k < 3; k++) out[k] = c[k]; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); for (int k = 0; k < 3; k++) out[k] = a[k] + w * ac[k]; return; }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int k = 0; k < 3; k++) out[k] = b[k] + w * (c[k] - b[k]);
        return;
    }
    const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
    for (int k = 0; k < 3; k++) out[k] = a[k] + ab[k] * v + ac[k] * w;
#undef D3
}
static real sphere_hull(const orc_params* prm, const real* sc, real sr, const real* bc, const real* Rb, real* n, real* pb) {
    hull_planes(prm);
    const int nv = prm->obj_hull_n;
    real d[3], dl[3];
    for (int k = 0; k < 3; k++) d[k] = sc[k] - bc[k];
    m3T_v(Rb, d, dl);
    const double p[3] = {(double)dl[0], (double)dl[1], (double)dl[2]};
    /* inside: behind every supporting plane */
    int inside = 1, best = -1; double depth = 1e300;
    for (int f = 0; f < g_hull.np; f++) {
        const double sd = g_hull.pl[f][0] * p[0] + g_hull.pl[f][1] * p[1] + g_hull.pl[f][2] * p[2] - g_hull.pl[f][3];
        if (sd > 0) { inside = 0; break; }
        if (-sd < depth) { depth = -sd; best = f; }
    }
    real nl[3], cl[3], dist;
    if (inside && best >= 0) {
        for (int k = 0; k < 3; k++) { nl[k] = (real)g_hull.pl[best][k]; cl[k] = (real)(p[k] + depth * g_hull.pl[best][k]); }
        dist = (real)(-depth) - sr;
    } else {
        double bd2 = 1e300, bcp[3] = {0, 0, 0};
        for (int i = 0; i < nv; i++) for (int j = i + 1; j < nv; j++) for (int k = j + 1; k < nv; k++) {
            double cp[3];
            closest_on_triangle(p, prm->obj_hull[i], prm->obj_hull[j], prm->obj_hull[k], cp);
            const double e[3] = {p[0] - cp[0], p[1] - cp[1], p[2] - cp[2]};
            const double d2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
            if (d2 < bd2) { bd2 = d2; for (int t = 0; t < 3; t++) bcp[t] = cp[t]; }
        }
        const double len = sqrt(bd2);
        if (len < 1e-12) { nl[0] = 0; nl[1] = 0; nl[2] = 1; }          /* on the surface to rounding: any normal; distance -sr */
        else for (int k = 0; k < 3; k++) nl[k] = (real)((p[k] - bcp[k]) / len);
        for (int k = 0; k < 3; k++) cl[k] = (real)bcp[k];
        dist = (real)len - sr;
    }
    m3_v(Rb, nl, n);
    real t[3]; m3_v(Rb, cl, t);
    for (int k = 0; k < 3; k++) pb[k] = bc[k] + t[k];
    return dist;
}

/* The object's candidate contact points against its support surface, as offsets from its centre in WORLD axes (8 slots; returns 0 for
 * a slot the shape does not use).  box: the 8 vertices.  sphere: the lowest point.  cylinder: per cap three rim points at 0 / 120 /
 * 240 degrees (slots 0-2 bottom cap, 4-6 top cap; an upright can stands on a tripod) and the rim's lowest point (slots 3 / 7: the
 * generator a lying can rolls on; undefined -- unused -- while the axis is vertical). */
static int shape_candidate(int shape, const real* h, const real* Ro, int v, real* r) {
    if (shape == 0) {
        real l[3] = {(v & 1 ? h[0] : -h[0]), (v & 2 ? h[1] : -h[1]), (v & 4 ? h[2] : -h[2])};
        m3_v(Ro, l, r);
        return 1;
    }
    if (shape == 1) { if (v != 0) return 0; r[0] = 0; r[1] = 0; r[2] = -h[0]; return 1; }
    const real s = v < 4 ? (real)-1 : (real)1;
    const int k = v & 3;
    real l[3];
    if (k < 3) {
        static const double cs[3] = {1.0, -0.5, -0.5}, sn[3] = {0.0, 0.86602540378443865, -0.86602540378443865};
        l[0] = h[0] * (real)cs[k]; l[1] = h[0] * (real)sn[k]; l[2] = s * h[2];
    } else {
        /* lowest point of the rim: the direction of -z (world) projected into the cap's plane, in local axes (-R^T z)_xy */
        const real dx = -Ro[6], dy = -Ro[7];
        const real len = (real)sqrt((double)(dx * dx + dy * dy));
        if (len < (real)1e-6) return 0;
        l[0] = h[0] * dx / len; l[1] = h[0] * dy / len; l[2] = s * h[2];
    }
    m3_v(Ro, l, r);
    return 1;
}

/* support height under a world point: table top inside the footprint (unless the point is below the slab), else ground */
static real support_height(const orc_params* p, const real* x) {
    real top = (real)(p->table_c[2] + p->table_h[2]);
    real bot = (real)(p->table_c[2] - p->table_h[2]);
    int in = fabs((double)x[0] - p->table_c[0]) <= p->table_h[0] && fabs((double)x[1] - p->table_c[1]) <= p->table_h[1];
    return (in && x[2] > bot) ? top : (real)p->ground_z;
}

typedef struct { int idx; real dist; real n[3], pA[3], pB[3]; int link; real mu; } cand_t;

/* keep the `cap` smallest-distance candidates with dist < margin, then order by idx */
static int select_contacts(cand_t* c, int n, int cap, real margin, cand_t* out) {
    int used[ORC_MAXS] = {0}, cnt = 0;
    for (int s = 0; s < cap; s++) {
        int best = -1;
        for (int i = 0; i < n; i++) if (!used[i] && c[i].dist < margin && (best < 0 || c[i].dist < c[best].dist)) best = i;
        if (best < 0) break;
        used[best] = 1; cnt++;
    }
    int k = 0;
    for (int i = 0; i < n; i++) if (used[i]) out[k++] = c[i];
    return cnt;
}

/* ------------------------------------------------------------------ rows + PGS */
typedef struct {
    real jA[ORC_MAXD], bA[ORC_MAXD], jB[6], bB[6];
    int useA, useB;
    real rhs, dinv, lo, hi, app, mu;
    int fidx;
} row_t;

static __thread int g_last_sweeps[2];      /* sweeps_used, sweeps_to_1e7 of this thread's last simulation step (orc_batch_step_sweeps) */
void orc_last_sweeps(int out[2]) { out[0] = g_last_sweeps[0]; out[1] = g_last_sweeps[1]; }
static real resolve_row(row_t* r, real* dvA, real* dvB, int nd) {
    /* btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric, cfm = 0 */
    real delta = r->rhs, dot = 0;
    if (r->useA) for (int k = 0; k < nd; k++) dot += r->jA[k] * dvA[k];
    if (r->useB) for (int k = 0; k < 6; k++) dot += r->jB[k] * dvB[k];
    delta -= dot * r->dinv;
    real sum = r->app + delta;
    if (sum < r->lo) { delta = r->lo - r->app; r->app = r->lo; }
    else if (sum > r->hi) { delta = r->hi - r->app; r->app = r->hi; }
    else r->app = sum;
    if (r->useA) for (int k = 0; k < nd; k++) dvA[k] += r->bA[k] * delta;
    if (r->useB) for (int k = 0; k < 6; k++) dvB[k] += r->bB[k] * delta;
    return delta;
}

/* linear-velocity Jacobian row of world point p on link `link` projected on direction dir */
static void point_jacobian(const orc_model* m, const aba_ws* w, int link, const real* p, const real* dir, real* J) {
    for (int k = 0; k < m->ndof; k++) J[k] = 0;
    for (int i = link; i >= 0; i = m->parent[i]) {
        if (m->jtype[i] == 0) continue;
        real aw[3]; m3_v(w->Rw[i], m->axis[i], aw);
        if (m->jtype[i] == 1) {
            real rr[3] = {p[0] - w->pw[i][0], p[1] - w->pw[i][1], p[2] - w->pw[i][2]}, t[3];
            cross(aw, rr, t);
            J[m->dof[i]] = dot3(dir, t);
        } else J[m->dof[i]] = dot3(dir, aw);
    }
}

void orc_sim_step(const orc_model* m, const orc_params* prm, real* st, const real* q_des, const real* kp,
                  const real* kd, orc_step_info* info) {
    orc_sim_step_f(m, prm, st, q_des, kp, kd, NULL, info);
}
static void sim_step_fv(const orc_model* m, const orc_params* prm, real* st, const real* q_des, const real* kp,
                        const real* kd, const real* fscale, const real* vmaxv, orc_step_info* info);
void orc_sim_step_f(const orc_model* m, const orc_params* prm, real* st, const real* q_des, const real* kp,
                    const real* kd, const real* fscale, orc_step_info* info) {
    sim_step_fv(m, prm, st, q_des, kp, kd, fscale, NULL, info);
}
/* vmaxv: NULL or per-DoF `maxVelocity` of the POSITION_CONTROL motors (0: none) */
static void sim_step_fv(const orc_model* m, const orc_params* prm_in, real* st, const real* q_des, const real* kp,
                        const real* kd, const real* fscale, const real* vmaxv, orc_step_info* info) {
    const int nd = m->ndof;
    /* per-env object parameters of the Panda task envs (domain randomisation; change_physics_params, panda_push_gym_env.py:362-368):
     * X[12] mass, X[13] lateral friction, X[15] 1 + linear damping of the object, 0 = the batch value; inertia scales with mass */
    orc_params pp = *prm_in;
    real okl = (real)pp.lin_damping;
    if (lay_w(m) == 16) {
        const real* X = st + OX(m);
        if (X[12] > 0) { for (int k = 0; k < 3; k++) pp.obj_inertia[k] *= (double)X[12] / pp.obj_mass; pp.obj_mass = (double)X[12]; }
        if (X[13] > 0) pp.obj_mu = (double)X[13];
        if (X[15] > 0) okl = X[15] - 1;
        /* ... and V[15]: 1 + linear damping of the robot's links (the `robot_damping` argument, :365-367), 0 = the batch value */
        if (st[OV(m) + 15] > 0) pp.lin_damping = (double)st[OV(m) + 15] - 1;
    }
    const orc_params* prm = &pp;
    /* robot-object contact slots: the Panda (task envs: SURVEY a6 "<= 4 cube-robot points"; robot-level interface: both spheres of both
     * fingers) keeps the 4 deepest, the 20-DoF iCub the 2 deepest (six stand-in spheres, three per arm), the iCub with hands 6 */
    const int nc_ro = m->ndof > 32 ? ORC_NC_RO_HANDS : ((m->ntip > 0 || m->ndof <= 9) ? ORC_NC_RO_PANDA : ORC_NC_RO);
    const real dt = (real)prm->dt;
    real* q = st; real* qd = st + OV(m);
    real* op = st + OQ(m); real* oq = op + 3; real* ov = qd + nd; real* ow = ov + 3;
    const int obj_on = !(prm->flags & ORC_F_NO_OBJECT);
    aba_ws* w = (aba_ws*)malloc(sizeof *w);
    orc_step_info li; if (!info) info = &li;
    memset(info, 0, sizeof *info);

    /* 1. kinematics + articulated inertias at q_t */
    aba_config_d(m, q, w, prm->implicit_joint_damping ? dt : 0);
    real Ro[9]; quat_to_R(oq, Ro);

    /* 2. collision detection at q_t (Bullet: performDiscreteCollisionDetection before the solve) */
    cand_t sel[ORC_NC]; int nsel = 0, n_ot = 0, n_ro = 0, n_rt = 0;
    real margin = (real)prm->contact_margin;
    real oh[3] = {(real)prm->obj_h[0], (real)prm->obj_h[1], (real)prm->obj_h[2]};
    real sc[ORC_MAXS][3];
    for (int s = 0; s < m->ns; s++) {
        real t[3]; m3_v(w->Rw[m->s_link[s]], m->s_c[s], t);
        for (int k = 0; k < 3; k++) sc[s][k] = w->pw[m->s_link[s]][k] + t[k];
    }
    if (obj_on) {
        /* (hull: every vertex is a candidate; the ORC_NC_OT deepest within the margin become the contact points, in vertex order -- the
         * box's rule over its 8 vertices) */
        const int ncand = prm->obj_shape == 3 ? prm->obj_hull_n : 8;
        cand_t c[ORC_MAXHV];
        for (int v = 0; v < ncand; v++) {
            real x[3];
            int used;
            if (prm->obj_shape == 3) { real l[3] = {(real)prm->obj_hull[v][0], (real)prm->obj_hull[v][1], (real)prm->obj_hull[v][2]}; m3_v(Ro, l, x); used = 1; }
            else used = shape_candidate(prm->obj_shape, oh, Ro, v, x);
            if (!used) { x[0] = 0; x[1] = 0; x[2] = (real)1e6; }      /* unused slot: far above any support */
            for (int k = 0; k < 3; k++) x[k] += op[k];
            real hs = support_height(prm, x);
            c[v].idx = v; c[v].dist = x[2] - hs; c[v].n[0] = 0; c[v].n[1] = 0; c[v].n[2] = 1;
            for (int k = 0; k < 3; k++) { c[v].pA[k] = x[k]; c[v].pB[k] = x[k]; }
            c[v].pB[2] = hs; c[v].link = -1; c[v].mu = (real)(prm->obj_mu * prm->table_mu);
        }
        n_ot = select_contacts(c, ncand, ORC_NC_OT, margin, sel);
        cand_t cs[ORC_MAXS];
        for (int s = 0; s < m->ns; s++) {
            cs[s].idx = s; cs[s].link = m->s_link[s]; cs[s].mu = (real)(m->s_mu[s] * prm->obj_mu);
            cs[s].dist = prm->obj_shape == 3 ? sphere_hull(prm, sc[s], m->s_r[s], op, Ro, cs[s].n, cs[s].pB)
                                             : sphere_shape(prm->obj_shape, sc[s], m->s_r[s], op, Ro, oh, cs[s].n, cs[s].pB);
            for (int k = 0; k < 3; k++) cs[s].pA[k] = cs[s].pB[k] + cs[s].n[k] * cs[s].dist;
        }
        n_ro = select_contacts(cs, m->ns, nc_ro, margin, sel + n_ot);
    }
    {
        cand_t cs[ORC_MAXS];
        real Rt[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        real tc[3] = {(real)prm->table_c[0], (real)prm->table_c[1], (real)prm->table_c[2]};
        real th[3] = {(real)prm->table_h[0], (real)prm->table_h[1], (real)prm->table_h[2]};
        for (int s = 0; s < m->ns; s++) {
            cs[s].idx = s; cs[s].link = m->s_link[s]; cs[s].mu = (real)(m->s_mu[s] * prm->table_mu);
            cs[s].dist = sphere_box(sc[s], m->s_r[s], tc, Rt, th, cs[s].n, cs[s].pB);
            for (int k = 0; k < 3; k++) cs[s].pA[k] = cs[s].pB[k] + cs[s].n[k] * cs[s].dist;
        }
        n_rt = select_contacts(cs, m->ns, ORC_NC_RT, margin, sel + n_ot + n_ro);
    }
    nsel = n_ot + n_ro + n_rt;

    /* 3. unconstrained accelerations; v* = v + dt a (Bullet: solveExternalForces -> applyDeltaVee) */
    real pA0[ORC_MAXL][6], tau[ORC_MAXD], qdd[ORC_MAXD], vs[ORC_MAXD], ovs[6];
    aba_velocity(m, prm, qd, w, pA0);
    for (int i = 0; i < m->nl; i++) if (m->dof[i] >= 0) tau[m->dof[i]] = -m->damping[i] * qd[m->dof[i]];
    real gb[3] = {0, 0, (real)(-prm->gravity_z)}, ab[6] = {0, 0, 0, 0, 0, 0};
    m3T_v(m->base_R, gb, ab + 3);
    aba_solve(m, w, tau, 1, pA0, ab, qdd);
    real mcv = (real)prm->max_coord_vel;
    for (int k = 0; k < nd; k++) {
        info->qdd[k] = qdd[k];
        vs[k] = qd[k] + dt * qdd[k];
        if (vs[k] > mcv) vs[k] = mcv; if (vs[k] < -mcv) vs[k] = -mcv;
    }
    real Iinv[9] = {0};
    if (obj_on) {
        real Il[3] = {(real)prm->obj_inertia[0], (real)prm->obj_inertia[1], (real)prm->obj_inertia[2]};
        real RoT[9], D[9] = {1/Il[0], 0, 0, 0, 1/Il[1], 0, 0, 0, 1/Il[2]}, T[9];
        m3_T(Ro, RoT); m3_mul(Ro, D, T); m3_mul(T, RoT, Iinv);
        real Dm[9] = {Il[0], 0, 0, 0, Il[1], 0, 0, 0, Il[2]}, Iw[9];
        m3_mul(Ro, Dm, T); m3_mul(T, RoT, Iw);
        real kl = okl, ka = (real)prm->ang_damping;
        real acc[6], Lw[3], gy[3], tq[3];
        real sl = kl + kl * norm3(ov);
        acc[0] = -sl * ov[0]; acc[1] = -sl * ov[1]; acc[2] = (real)prm->gravity_z - sl * ov[2];
        m3_v(Iw, ow, Lw); cross(ow, Lw, gy);
        real sa = ka + ka * norm3(ow);
        for (int k = 0; k < 3; k++) tq[k] = -gy[k] - sa * Lw[k];
        m3_v(Iinv, tq, acc + 3);
        for (int k = 0; k < 3; k++) { ovs[k] = ov[k] + dt * acc[k]; ovs[3+k] = ow[k] + dt * acc[3+k]; }
        for (int k = 0; k < 6; k++) { info->obj_acc[k] = acc[k]; if (ovs[k] > mcv) ovs[k] = mcv; if (ovs[k] < -mcv) ovs[k] = -mcv; }
    } else for (int k = 0; k < 6; k++) ovs[k] = 0;

    /* 4. constraint rows */
    row_t* nc = (row_t*)calloc(3 * ORC_MAXD, sizeof(row_t)); int n_nc = 0;    /* limits then motors */
    row_t* rn = (row_t*)calloc(ORC_NC, sizeof(row_t));
    row_t* rf = (row_t*)calloc(2 * ORC_NC, sizeof(row_t));
    real zero_ab[6] = {0, 0, 0, 0, 0, 0};
    int motor_row[ORC_MAXD];
    /* joint limits: btMultiBodyJointLimitConstraint, created at URDF load before the motors [EXT-UNVERIFIED] */
    for (int i = 0; i < m->nl; i++) {
        if (m->dof[i] < 0) continue;
        int j = m->dof[i];
        for (int side = 0; side < 2; side++) {
            real pen = side == 0 ? q[j] - m->lower[i] : m->upper[i] - q[j];
            if (pen > 0) continue;
            row_t* r = &nc[n_nc++];
            real dir = side ? (real)-1 : (real)1;
            r->useA = 1; r->jA[j] = dir;
            aba_solve(m, w, r->jA, 0, NULL, zero_ab, r->bA);
            r->dinv = 1 / (r->jA[j] * r->bA[j]);
            real rel = dir * vs[j];
            r->rhs = (-pen * (real)prm->erp / dt - rel) * r->dinv;
            r->lo = 0; r->hi = (real)prm->limit_max_impulse;
        }
    }
    /* joint motors: btMultiBodyJointMotor (POSITION_CONTROL): target velocity kp*(q_des-q)/dt + (1-kd)*v */
    for (int i = 0; i < m->nl; i++) {
        if (m->dof[i] < 0) continue;
        int j = m->dof[i];
        motor_row[j] = n_nc;
        row_t* r = &nc[n_nc++];
        r->useA = 1; r->jA[j] = 1;
        aba_solve(m, w, r->jA, 0, NULL, zero_ab, r->bA);
        r->dinv = 1 / r->bA[j];
        real verr = kp[j] * (q_des[j] - q[j]) / dt - kd[j] * vs[j];
        if (vmaxv && vmaxv[j] > 0) {
            /* setJointMotorControl2(maxVelocity): the target velocity kp dq/dt + (1 - kd) v of the row is clamped to +-maxVelocity
             * (btMultiBodyJointMotor m_rhsClamp [EXT-UNVERIFIED]); the velocity error is target - v */
            real vt = kp[j] * (q_des[j] - q[j]) / dt + (1 - kd[j]) * vs[j];
            vt = vt > vmaxv[j] ? vmaxv[j] : (vt < -vmaxv[j] ? -vmaxv[j] : vt);
            verr = vt - vs[j];
        }
        r->rhs = verr * r->dinv;
        r->hi = (real)prm->max_motor_impulse * (fscale ? fscale[j] : (real)1);
        if (m->max_force[m->link_of_dof[j]] > 0) r->hi = m->max_force[m->link_of_dof[j]] * dt;      /* the base constraint's rows: its own maxForce */
        r->lo = -r->hi;
    }
    /* contacts: normal row + 2 friction rows (btPlaneSpace1 directions) */
    for (int c = 0; c < nsel; c++) {
        cand_t* cc = &sel[c];
        int type = c < n_ot ? 0 : (c < n_ot + n_ro ? 1 : 2);
        real t1[3], t2[3]; const real* n = cc->n;
        if (fabs((double)n[2]) > 0.7071067811865475244) {
            real a = n[1]*n[1] + n[2]*n[2], k = 1 / (real)sqrt((double)a);
            t1[0] = 0; t1[1] = -n[2]*k; t1[2] = n[1]*k;
            t2[0] = a*k; t2[1] = -n[0]*t1[2]; t2[2] = n[0]*t1[1];
        } else {
            real a = n[0]*n[0] + n[1]*n[1], k = 1 / (real)sqrt((double)a);
            t1[0] = -n[1]*k; t1[1] = n[0]*k; t1[2] = 0;
            t2[0] = -n[2]*t1[1]; t2[1] = n[2]*t1[0]; t2[2] = a*k;
        }
        const real* dirs[3] = {n, t1, t2};
        for (int d = 0; d < 3; d++) {
            row_t* r = d == 0 ? &rn[c] : &rf[2*c + d - 1];
            const real* dir = dirs[d];
            real rel = 0, denom = 0;
            if (type == 0) {            /* A = object (+dir), B = static */
                real rr[3] = {cc->pA[0] - op[0], cc->pA[1] - op[1], cc->pA[2] - op[2]}, t[3];
                cross(rr, dir, t);
                r->useB = 1;
                for (int k = 0; k < 3; k++) { r->jB[k] = dir[k]; r->jB[3+k] = t[k]; }
            } else {                    /* A = robot (+dir), B = object (-dir) or static */
                r->useA = 1;
                point_jacobian(m, w, cc->link, cc->pA, dir, r->jA);
                aba_solve(m, w, r->jA, 0, NULL, zero_ab, r->bA);
                if (type == 1) {
                    real rr[3] = {cc->pB[0] - op[0], cc->pB[1] - op[1], cc->pB[2] - op[2]}, t[3];
                    cross(rr, dir, t);
                    r->useB = 1;
                    for (int k = 0; k < 3; k++) { r->jB[k] = -dir[k]; r->jB[3+k] = -t[k]; }
                }
            }
            if (r->useB) {
                for (int k = 0; k < 3; k++) r->bB[k] = r->jB[k] / (real)prm->obj_mass;
                m3_v(Iinv, r->jB + 3, r->bB + 3);
                for (int k = 0; k < 6; k++) { rel += r->jB[k] * ovs[k]; denom += r->jB[k] * r->bB[k]; }
            }
            if (r->useA) for (int k = 0; k < nd; k++) { rel += r->jA[k] * vs[k]; denom += r->jA[k] * r->bA[k]; }
            r->dinv = 1 / denom;
            r->mu = cc->mu; r->fidx = c;
            if (d == 0) {
                /* btMultiBodyConstraintSolver::setupMultiBodyContactConstraint, restitution 0 */
                real pen = cc->dist + (real)prm->linear_slop;
                real perr = 0, verr = -rel;
                if (pen > 0) verr -= pen / dt; else perr = -pen * (real)prm->erp / dt;
                r->rhs = (perr + verr) * r->dinv; r->lo = 0; r->hi = (real)1e10;
            } else { r->rhs = -rel * r->dinv; r->lo = -cc->mu; r->hi = cc->mu; }
        }
    }

    /* 5. projected Gauss-Seidel, Bullet row order (btMultiBodyConstraintSolver::solveSingleIteration):
     *    non-contact rows (order alternates with iteration parity), all normals, all frictions */
    real dvA[ORC_MAXD] = {0}, dvB[6] = {0};
    /* Least-squares residual of a sweep as Bullet keeps it (btMultiBodyConstraintSolver::solveSingleIteration: the maximum over the rows
     * of (deltaImpulse / jacDiagABInv)^2, i.e. of the squared velocity-level change of the row) and the loop's exit test
     * (btSequentialImpulseConstraintSolver::solveGroupCacheFriendlyIterations: residual <= m_leastSquaresResidualThreshold) [EXT-UNVERIFIED].
     * With the threshold at 0 the loop ends early only after a sweep that changed nothing, which leaves the result as it is. */
    info->sweeps_used = prm->solver_iters; info->sweeps_to_1e7 = prm->solver_iters + 1; info->last_sq_residual = 0;
#define ORC_RES(row) do { real d_ = resolve_row((row), dvA, dvB, nd) / (row)->dinv; d_ *= d_; if (d_ > lsr) lsr = d_; } while (0)
    for (int it = 0; it < prm->solver_iters; it++) {
        real lsr = 0;
        for (int j = 0; j < n_nc; j++) {
            int idx = (it & 1) ? j : n_nc - 1 - j;
            ORC_RES(&nc[idx]);
        }
        for (int c = 0; c < nsel; c++) ORC_RES(&rn[c]);
        for (int f = 0; f < 2 * nsel; f++) {
            real tot = rn[rf[f].fidx].app;
            if (tot > 0) { rf[f].lo = -rf[f].mu * tot; rf[f].hi = rf[f].mu * tot; ORC_RES(&rf[f]); }
        }
        info->last_sq_residual = lsr;
        if (lsr <= (real)1e-7 && info->sweeps_to_1e7 > prm->solver_iters) info->sweeps_to_1e7 = it + 1;
        if (lsr <= (real)prm->solver_residual_threshold) { info->sweeps_used = it + 1; break; }
    }
#undef ORC_RES
    g_last_sweeps[0] = info->sweeps_used; g_last_sweeps[1] = info->sweeps_to_1e7;

    /* 6. velocity update + position integration (stepPositionsMultiDof) */
    for (int k = 0; k < nd; k++) {
        real v = vs[k] + dvA[k];
        if (v > mcv) v = mcv; if (v < -mcv) v = -mcv;
        qd[k] = v; q[k] += dt * v;
        info->motor_impulse[k] = nc[motor_row[k]].app;
    }
    if (obj_on) {
        for (int k = 0; k < 6; k++) { real v = ovs[k] + dvB[k]; if (v > mcv) v = mcv; if (v < -mcv) v = -mcv; ov[k] = v; }
        for (int k = 0; k < 3; k++) op[k] += dt * ov[k];
        /* quaternion exponential-map update (btMultiBody pQuatUpdateFun, world-frame omega) [EXT-UNVERIFIED] */
        real ang = norm3(ow), ax[3];
        if (ang * dt > (real)(0.5 * PI * 0.5)) ang = (real)(0.5 * PI * 0.5) / dt;
        real sc_;
        if (ang < (real)0.001) sc_ = (real)0.5 * dt - dt*dt*dt * (real)0.020833333333 * ang * ang;
        else sc_ = (real)sin(0.5 * (double)ang * (double)dt) / ang;
        for (int k = 0; k < 3; k++) ax[k] = ow[k] * sc_;
        real dq[4] = {ax[0], ax[1], ax[2], (real)cos((double)ang * (double)dt * 0.5)}, nq[4];
        quat_mul(dq, oq, nq);
        real nn = (real)sqrt((double)(nq[0]*nq[0] + nq[1]*nq[1] + nq[2]*nq[2] + nq[3]*nq[3]));
        for (int k = 0; k < 4; k++) oq[k] = nq[k] / nn;
    }
    if (m->ntip > 0) {
        /* check_contact_fingertips / check_collision (icub_env_with_hands.py:246-318): mean normal force (impulse / dt) of the
         * contact points of each fingertip with the object, tips in contact, robot-object contact points */
        real tf[ORC_NTIP] = {0}; int tc[ORC_NTIP] = {0}, ntip = 0;
        for (int c = n_ot; c < n_ot + n_ro; c++) {
            const int tp = m->s_tip[sel[c].idx];
            if (tp > 0) { tf[tp - 1] += rn[c].app / dt; tc[tp - 1]++; }
        }
        real* T = st + OQ(m) + 7;
        for (int k = 0; k < ORC_NTIP; k++) { T[k] = tc[k] ? tf[k] / (real)tc[k] : 0; ntip += tc[k] > 0; }
        T[ORC_NTIP] = (real)ntip; T[ORC_NTIP + 1] = (real)n_ro;
    }
    info->ncontacts = nsel;
    for (int c = 0; c < nsel; c++) {
        info->type[c] = c < n_ot ? 0 : (c < n_ot + n_ro ? 1 : 2);
        info->link[c] = sel[c].link; info->idx[c] = sel[c].idx; info->dist[c] = sel[c].dist; info->mu[c] = sel[c].mu;
        for (int k = 0; k < 3; k++) { info->n[c][k] = sel[c].n[k]; info->pA[c][k] = sel[c].pA[k]; info->pB[c][k] = sel[c].pB[k]; }
        info->lambda_n[c] = rn[c].app; info->lambda_f1[c] = rf[2*c].app; info->lambda_f2[c] = rf[2*c+1].app;
    }
    free(nc); free(rn); free(rf); free(w);
}

/* ------------------------------------------------------------------ RNG */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static real u01(uint32_t x) { return (real)(x >> 8) * (real)(1.0 / 16777216.0); }


/* ------------------------------------------------------------------ inverse kinematics
 * PyBullet's calculateInverseKinematics(maxNumIterations=100, residualThreshold=1e-3) (panda_env.py:269-272,
 * icub_env.py:307-312) iterates a damped-least-squares update of the end-effector link pose from the current joint
 * angles [EXT-UNVERIFIED: Bullet's BussIK "DLS with orientation"; the damping value and the exact stopping rule are
 * restated, not verified].  Here:
 *   e = [p_target - p_link ; axis-angle(R_target R_link^T)],  dq = J^T (J J^T + lambda^2 I)^-1 e,  stop when |e_pos| < threshold.
 * The target is given for the hand COM frame and moved to the link frame with the constant ik_link_offset
 * (icub_env.py:252-258, 300-305; zero for the Panda whose end-effector link has no COM offset).  Only joints on the
 * chain to the end effector move (on the iCub those are exactly the controlled torso + arm joints, whose joint
 * damping the reference sets to 0.1; the blocked joints get 100 and are overwritten with their rest pose anyway,
 * icub_env.py:171, 316-317); the others keep their current value.  Returns iterations. */
static void euler_to_R(const real* e, real* R) { real q[4]; orc_quat_from_euler(e, q); quat_to_R(q, R); }
int orc_ik(const orc_model* m, const orc_task* t, const real* q_start, const real* pos, const real* euler, real* q) {
    const int nd = m->ndof, ee = m->ee_link;
    real Rt[9]; euler_to_R(euler, Rt);
    real off[3] = {(real)t->ik_link_offset[0], (real)t->ik_link_offset[1], (real)t->ik_link_offset[2]}, tp[3];
    m3_v(Rt, off, tp);
    for (int k = 0; k < 3; k++) tp[k] += pos[k];
    for (int k = 0; k < nd; k++) q[k] = q_start[k];
    real R[ORC_MAXL*9], p[ORC_MAXL*3];
    int it = 0;
    for (; it < t->ik_max_iters; it++) {
        orc_fk(m, q, R, p);
        const real* pe = p + 3*ee;
        real e[6];
        for (int k = 0; k < 3; k++) e[k] = tp[k] - pe[k];
        if (norm3(e) < (real)t->ik_residual) break;
        /* orientation error as a world-frame rotation vector */
        real ReT[9], Rerr[9]; m3_T(R + 9*ee, ReT); m3_mul(Rt, ReT, Rerr);
        real sx = Rerr[7] - Rerr[5], sy = Rerr[2] - Rerr[6], sz = Rerr[3] - Rerr[1];
        real s2 = (real)sqrt((double)(sx*sx + sy*sy + sz*sz)) * (real)0.5;          /* sin(angle) */
        real c2 = (Rerr[0] + Rerr[4] + Rerr[8] - 1) * (real)0.5;                     /* cos(angle) */
        real ang = (real)atan2((double)s2, (double)c2);
        real f = s2 > (real)1e-9 ? ang / (2 * s2) : (real)0.5;
        e[3] = f * sx; e[4] = f * sy; e[5] = f * sz;
        /* Jacobian columns of the joints on the chain */
        real J[6][ORC_MAXD]; memset(J, 0, sizeof J);
        for (int i = ee; i >= 0; i = m->parent[i]) {
            if (m->jtype[i] == 0 || m->passive[i]) continue;
            real aw[3], tt[3]; m3_v(R + 9*i, m->axis[i], aw);
            const int d = m->dof[i];
            if (m->jtype[i] == 1) { real rr[3] = {pe[0]-p[3*i], pe[1]-p[3*i+1], pe[2]-p[3*i+2]}; cross(aw, rr, tt); for (int k = 0; k < 3; k++) { J[k][d] = tt[k]; J[3+k][d] = aw[k]; } }
            else for (int k = 0; k < 3; k++) J[k][d] = aw[k];
        }
        /* A = J J^T + lambda^2 I ; solve A y = e by Cholesky ; dq = J^T y */
        real A[6][6], L[6][6], y[6];
        const real l2 = (real)(t->ik_damping * t->ik_damping);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { real sum = a == b ? l2 : 0; for (int d = 0; d < nd; d++) sum += J[a][d] * J[b][d]; A[a][b] = sum; }
        memset(L, 0, sizeof L);
        for (int a = 0; a < 6; a++) for (int b = 0; b <= a; b++) {
            real sum = A[a][b]; for (int k = 0; k < b; k++) sum -= L[a][k] * L[b][k];
            L[a][b] = a == b ? (real)sqrt((double)sum) : sum / L[b][b];
        }
        for (int a = 0; a < 6; a++) { real sum = e[a]; for (int k = 0; k < a; k++) sum -= L[a][k] * y[k]; y[a] = sum / L[a][a]; }
        for (int a = 5; a >= 0; a--) { real sum = y[a]; for (int k = a + 1; k < 6; k++) sum -= L[k][a] * y[k]; y[a] = sum / L[a][a]; }
        for (int d = 0; d < nd; d++) { real sum = 0; for (int a = 0; a < 6; a++) sum += J[a][d] * y[a]; q[d] += sum; }
    }
    return it;
}

/* ------------------------------------------------------------------ task layer */
void orc_default_task(orc_task* t, int task) {
    memset(t, 0, sizeof *t);
    t->task = task; t->max_steps = 1000;
    t->target_dist_min = task == 0 ? 0.03 : 0.1;   /* panda_reach_gym_env.py:47, panda_push_gym_env.py:52 */
    t->h_table = 0.625;
    /* WorldEnv workspace = robot workspace x,y (panda_env.py:37) with z = [h, h+0.3] (world_env.py:72) */
    t->ws_lim[0][0] = 0.3; t->ws_lim[0][1] = 0.65; t->ws_lim[1][0] = -0.3; t->ws_lim[1][1] = 0.3;
    t->ws_lim[2][0] = 0.625; t->ws_lim[2][1] = 0.925;
    const double home[9] = {0.0, -0.54, 0.0, -2.6, -0.30, 2.0, 1.0, 0.02, 0.02};  /* panda_env.py:19-23 */
    for (int k = 0; k < 9; k++) t->home[k] = home[k];
    t->act_scale = 0.05;                            /* panda_push_gym_env.py:225 */
    t->kp_act = 0.5; t->kd_act = 1.0;               /* panda_env.py:308 */
    t->kp_hold = 0.2; t->kd_hold = 1.0;             /* panda_env.py:76 */
    t->n_act = 7; t->seed = 1234;
    t->use_ik = 0; t->ik_damping = 0.1; t->ik_residual = 1e-3; t->ik_max_iters = 100;   /* panda_env.py:269-272; damping [EXT-UNVERIFIED] */
    const double hh[6] = {0.2, 0.0, 0.8, PI, 0.0, 0.0};                                /* panda_env.py:85-88 */
    for (int k = 0; k < 6; k++) t->home_hand_pose[k] = hh[k];
    t->robot_ws[0][0] = 0.3; t->robot_ws[0][1] = 0.65; t->robot_ws[1][0] = -0.3; t->robot_ws[1][1] = 0.3;   /* panda_env.py:37 */
    t->robot_ws[2][0] = task == 0 ? 0.625 : 0.425; t->robot_ws[2][1] = 1.5;            /* z-min: panda_reach_gym_env.py:69 / panda_push_gym_env.py:74 */
    t->robot = 0; t->n_joints_ctrl = 7;
    for (int k = 0; k < ORC_MAXACT; k++) t->act_dof[k] = k < 7 ? k : -1;
    t->control_orientation = 1; t->ik_pos_scale = 0.005; t->ik_rot_scale = 0.01;       /* panda_push_gym_env.py:200-203 */
    for (int k = 0; k < 3; k++) { t->eu_lim[k][0] = -PI; t->eu_lim[k][1] = PI; }        /* panda_env.py:38 */
}

/* iCub variants: icub_env.py:52-82 (workspace, Euler limits, home hand pose per arm), icub_reach_gym_env.py:27-35,
 * icub_push_gym_env.py:27-37 (thresholds 0.03, max_steps 2000, reward_type 1) */
void orc_task_icub(orc_task* t, int task, int right_arm, int use_ik, int control_orientation, const int* ctrl_dof,
                   const double* home, int ndof) {
    orc_default_task(t, task);
    t->robot = 1; t->max_steps = 2000; t->target_dist_min = 0.03;
    t->ws_lim[0][0] = 0.1; t->ws_lim[0][1] = 0.45; t->ws_lim[1][0] = -0.3; t->ws_lim[1][1] = 0.3;
    t->ws_lim[2][0] = t->h_table; t->ws_lim[2][1] = t->h_table + 0.3;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) t->robot_ws[a][b] = t->ws_lim[a][b];
    t->robot_ws[2][0] = t->h_table; t->robot_ws[2][1] = 1.0;            /* icub_reach_gym_env.py:78-80 */
    for (int k = 0; k < ORC_MAXD; k++) t->home[k] = k < ndof ? home[k] : 0.0;
    t->n_joints_ctrl = 10;
    for (int k = 0; k < ORC_MAXACT; k++) t->act_dof[k] = k < 10 ? ctrl_dof[k] : -1;
    t->use_ik = use_ik; t->control_orientation = control_orientation;
    t->n_act = use_ik ? (control_orientation ? 6 : 3) : 10;
    t->ik_pos_scale = control_orientation ? 0.01 : 0.005; t->ik_rot_scale = 0.02;   /* icub_reach_gym_env.py:206-212 */
    const double hl[6] = {0.3, 0.26, 0.8, 0, 0, 0}, hr[6] = {0.3, -0.26, 0.8, 0, 0, PI};
    for (int k = 0; k < 6; k++) t->home_hand_pose[k] = right_arm ? hr[k] : hl[k];
    for (int k = 0; k < 3; k++) { t->eu_lim[k][0] = -PI / 2; t->eu_lim[k][1] = PI / 2; }
    if (right_arm) { t->eu_lim[2][0] = PI / 2; t->eu_lim[2][1] = 1.5 * PI; }
    const double ol[3] = {-0.064768, -0.00563, -0.02266}, orr[3] = {0.064668, -0.0056, -0.022681};   /* icub_env.py:252-258 */
    for (int k = 0; k < 3; k++) t->ik_link_offset[k] = right_arm ? orr[k] : ol[k];
    t->reward_type = 1;
}
static int n_obs_joints(const orc_task* t, const orc_model* m) { return t->robot >= 1 ? t->n_joints_ctrl : m->ndof; }
int orc_obs_dim(const orc_task* t, const orc_model* m) { return 9 + n_obs_joints(t, m) + 6 + 6 + (t->task >= 1 ? 3 : 0) + (m->ntip > 0 ? ORC_NTIP + 2 : 0); }

/* end-effector state as p.getLinkState(computeLinkVelocity=1) reports it: COM frame position, orientation, linear velocity */
static void ee_state(const orc_model* m, const real* st, real* pos, real* quat, real* vlin) {
    real R[ORC_MAXL*9], p[ORC_MAXL*3];
    orc_fk(m, st, R, p);
    int e = m->ee_link;
    real c[3]; m3_v(R + 9*e, m->com[e], c);           /* getLinkState[0] is the link COM frame */
    for (int k = 0; k < 3; k++) pos[k] = p[3*e+k] + c[k];
    R_to_quat(R + 9*e, quat);
    vlin[0] = vlin[1] = vlin[2] = 0;
    for (int i = e; i >= 0; i = m->parent[i]) {
        if (m->jtype[i] == 0) continue;
        real aw[3], t[3]; m3_v(R + 9*i, m->axis[i], aw);
        real qd = st[OV(m) + m->dof[i]];
        if (m->jtype[i] == 1) { real rr[3] = {pos[0]-p[3*i], pos[1]-p[3*i+1], pos[2]-p[3*i+2]}; cross(aw, rr, t); }
        else { t[0] = aw[0]; t[1] = aw[1]; t[2] = aw[2]; }
        for (int k = 0; k < 3; k++) vlin[k] += t[k] * qd;
    }
}

void orc_observation(const orc_model* m, const orc_task* t, const real* st, real* obs) {
    /* panda_env.py:141-193 + panda_push_gym_env.py:150-187; icub_env.py:202-249 + icub_reach_gym_env.py:150-180 */
    real pos[3], quat[4], vl[3], eu[3];
    const real* ob = st + OQ(m);
    ee_state(m, st, pos, quat, vl);
    orc_euler_from_quat(quat, eu);
    int o = 0;
    for (int k = 0; k < 3; k++) obs[o++] = pos[k];
    for (int k = 0; k < 3; k++) obs[o++] = eu[k];
    if (t->robot >= 1) {                                   /* iCub: raw velocity, controlled joints only */
        for (int k = 0; k < 3; k++) obs[o++] = vl[k];
        for (int k = 0; k < t->n_joints_ctrl; k++) obs[o++] = st[t->act_dof[k]];
    } else {
        const real vmean[3] = {0, (real)0.01, 0}, vstd[3] = {(real)0.04, (real)0.07, (real)0.03};
        for (int k = 0; k < 3; k++) obs[o++] = (vl[k] - vmean[k]) / vstd[k];
        for (int k = 0; k < m->ndof; k++) obs[o++] = st[k];
    }
    real oe[3];
    orc_euler_from_quat(ob + 3, oe);
    for (int k = 0; k < 3; k++) obs[o++] = ob[k];
    for (int k = 0; k < 3; k++) obs[o++] = oe[k];
    /* object pose in the hand frame: invertTransform(ee_pos, quatFromEuler(ee_eul)) * (obj_pos, quatFromEuler(obj_eul)) */
    real qh[4], qo[4], Rh[9], d[3], rel[3], qhi[4], qr[4], er[3];
    orc_quat_from_euler(eu, qh); orc_quat_from_euler(oe, qo);
    quat_to_R(qh, Rh);
    for (int k = 0; k < 3; k++) d[k] = ob[k] - pos[k];
    m3T_v(Rh, d, rel);
    qhi[0] = -qh[0]; qhi[1] = -qh[1]; qhi[2] = -qh[2]; qhi[3] = qh[3];
    quat_mul(qhi, qo, qr);
    orc_euler_from_quat(qr, er);
    for (int k = 0; k < 3; k++) obs[o++] = rel[k];
    for (int k = 0; k < 3; k++) obs[o++] = er[k];
    if (t->task >= 1) for (int k = 0; k < 3; k++) obs[o++] = st[OX(m)+k];
    if (m->ntip > 0) for (int k = 0; k < ORC_NTIP + 2; k++) obs[o++] = st[OQ(m) + 7 + k];    /* fingertip forces and counts */
}

/* reward + termination; pre_increment=1 reproduces the in-loop `_termination()` + counter++ of apply_action
 * (panda_push_gym_env.py:239-242) before the final `_termination()`/`_compute_reward()` of step (:252-253) */
void orc_reward_done(const orc_model* m, const orc_task* t, real* st, int pre_increment, real* reward, real* done) {
    real pos[3], quat[4], vl[3];
    real* X = st + OX(m); const real* ob = st + OQ(m);
    ee_state(m, st, pos, quat, vl);
    real d1 = 0, d2 = 0;
    for (int k = 0; k < 3; k++) { real a = pos[k] - ob[k], b = ob[k] - X[k]; d1 += a*a; d2 += b*b; }
    d1 = (real)sqrt((double)d1); d2 = (real)sqrt((double)d2);
    real dsucc = t->task >= 1 ? d2 : d1;
    const real thr = (real)t->target_dist_min;
    int succ = dsucc <= thr;
    int cnt = (int)X[3], term = (int)X[4];
    if (t->task == 2) {
        /* pandaPushGymGoalEnv / iCubPushGymGoalEnv (panda_push_gym_goal_env.py:89-122, icub_push_gym_goal_env.py:103-139):
         * _termination() is only the step budget (so the in-loop check of apply_action never sees success),
         * done = budget or success, reward = -(d > thr) */
        if (pre_increment && !(cnt > t->max_steps)) cnt++;
        *done = (cnt > t->max_steps || succ) ? (real)1 : (real)0;
        *reward = succ ? (real)0 : (real)-1;
        X[3] = (real)cnt;
        return;
    }
    if (pre_increment) {
        int d0 = succ || term || cnt > t->max_steps;
        if (succ) term = 1;
        if (!d0) cnt++;
    }
    if (succ) term = 1;
    *done = (succ || term || cnt > t->max_steps) ? (real)1 : (real)0;
    if (t->robot >= 1) {
        if (t->task == 0) {                               /* icub_reach_gym_env.py:318-330: the bonus is ADDED */
            *reward = -d1; if (d1 <= thr) *reward += (real)1000 + ((real)100 - d1 * 80);
        } else if (t->reward_type == 0) {                 /* icub_push_gym_env.py:353-356 */
            *reward = -d1 - d2; if (d2 <= thr) *reward += (real)1000;
        } else {                                          /* :359-371, distances normalised by their values at reset */
            const real rew1 = (real)0.125, rew2 = (real)0.25;
            *reward = rew1 * (1 - d1 / X[12]);
            if (!(d1 > (real)0.1)) *reward += rew2 * (1 - d2 / X[13]);
            if (d2 <= thr) *reward += (real)1000;
        }
    }
    else if (t->task == 1) { *reward = -d1 - d2; if (d2 <= thr) *reward = (real)1000 + ((real)100 - d2 * 80); }
    else { *reward = -d1; if (d1 <= thr) *reward = (real)1000 + ((real)100 - d1 * 80); }
    X[3] = (real)cnt; X[4] = (real)term;
}

static void hold_targets(const orc_task* t, int nd, real* qdes, real* kp, real* kd) {
    for (int k = 0; k < nd; k++) { qdes[k] = (real)t->home[k]; kp[k] = (real)t->kp_hold; kd[k] = (real)t->kd_hold; }
}
static real clampr(real x, double lo, double hi) { return x < (real)lo ? (real)lo : (x > (real)hi ? (real)hi : x); }
/* joint targets of the IK branch: chain joints from the IK solution; iCub: every other joint at its rest pose
 * (icub_env.py:316-317); Panda: the IK returns the current finger positions */
static void ik_targets(const orc_model* m, const orc_task* t, const real* st, const real* hp, real* qdes) {
    real sol[ORC_MAXD];
    orc_ik(m, t, st, hp, hp + 3, sol);
    if (t->robot >= 1) { for (int k = 0; k < t->n_joints_ctrl; k++) qdes[t->act_dof[k]] = sol[t->act_dof[k]]; }
    else for (int k = 0; k < m->ndof; k++) qdes[k] = sol[k];
}

static void env_reset_impl(const orc_model* m, const orc_params* prm, const orc_task* t, uint64_t env_id, uint32_t episode,
                           real* st, real* obs, real* mrec);
void orc_env_reset(const orc_model* m, const orc_params* prm, const orc_task* t, uint64_t env_id, uint32_t episode,
                   real* st, real* obs) { env_reset_impl(m, prm, t, env_id, episode, st, obs, NULL); }
void orc_hands_reset(const orc_model* m, const orc_params* prm, const orc_task* t, uint64_t env_id, uint32_t episode,
                     real* st, real* mrec, real* obs) { env_reset_impl(m, prm, t, env_id, episode, st, obs, mrec); }
static void env_reset_impl(const orc_model* m, const orc_params* prm, const orc_task* t, uint64_t env_id, uint32_t episode,
                           real* st, real* obs, real* mrec) {
    /* reset_simulation (panda_push_gym_env.py:117-148, icub_reach_gym_env.py:121-148): robot at home, 100 steps,
     * load world, 100 steps, 1 step */
    const int nd = m->ndof;
    real* X = st + OX(m); real* ob = st + OQ(m);
    {   /* the per-env object parameters (X[12], X[13], X[15] of a Panda task env) are not part of the episode: a reset keeps them */
        real keep[3] = {0, 0, 0};
        const int pe = lay_w(m) == 16;
        if (pe) { keep[0] = st[OX(m) + 12]; keep[1] = st[OX(m) + 13]; keep[2] = st[OX(m) + 15]; }
        memset(st, 0, (size_t)orc_state_floats(m) * sizeof(real));
        if (pe) { st[OX(m) + 12] = keep[0]; st[OX(m) + 13] = keep[1]; st[OX(m) + 15] = keep[2]; }
    }
    for (int k = 0; k < nd; k++) st[k] = (real)t->home[k];
    /* WorldEnv._sample_pose (world_env.py:145-176) */
    real x_min = (real)t->ws_lim[0][0] + (real)0.05, x_max = (real)t->ws_lim[0][1] - (real)0.1;
    real y_min = (real)t->ws_lim[1][0] + (real)0.05, y_max = (real)t->ws_lim[1][1] - (real)0.05;
    real px = x_min + (real)0.5 * (x_max - x_min), py = y_min + (real)0.5 * (y_max - y_min);
    /* (the robot-level scenes -- ik_absolute: helloworld_icub.py:51, helloworld_panda.py:78 -- load their object with p.loadURDF(path,
     * position): identity orientation, not WorldEnv's yaw of pi/4) */
    real pz = (real)t->h_table + (real)0.07, yaw = t->ik_absolute ? (real)0 : (real)(0.25 * PI);
    uint32_t r[4];
    orc_philox4x32((uint32_t)env_id, (uint32_t)(env_id >> 32), episode, 0u, (uint32_t)t->seed, (uint32_t)(t->seed >> 32), r);
    if (t->obj_pose_rnd_std > 0) {
        real s = (real)t->obj_pose_rnd_std;
        px += -s + 2 * s * u01(r[0]);
        py += -s + 2 * s * u01(r[1]);
        yaw = (real)(-0.25 * PI) + (real)(0.5 * PI) * u01(r[2]);
    }
    px = px < x_min ? x_min : (px > x_max ? x_max : px);
    py = py < y_min ? y_min : (py > y_max ? y_max : py);
    real e[3] = {0, 0, yaw};
    ob[0] = px; ob[1] = py; ob[2] = pz;
    orc_quat_from_euler(e, ob + 3);
    real qdes_l[ORC_MAXD], kp_l[ORC_MAXD], kd[ORC_MAXD];
    /* iCub with hands: the motors' commands persist in the env's motor record (iCubHandsEnv.reset, icub_env_with_hands.py:108-121) */
    real* qdes = mrec ? mrec : qdes_l; real* kp = mrec ? mrec + ORC_MAXD : kp_l; real* fs = mrec ? mrec + 2 * ORC_MAXD : NULL;
    hold_targets(t, nd, qdes, kp, kd);
    if (fs) for (int k = 0; k < ORC_MAXD; k++) { fs[k] = 1; mrec[3 * ORC_MAXD + k] = 0; }
    orc_params p1 = *prm; p1.flags |= ORC_F_NO_OBJECT;
    if (t->use_ik) {
        /* robot.reset with use_IK (panda_env.py:83-91, icub_env.py:147-148): apply_action(home_hand_pose) -> IK once from
         * the joint home pose, motors (all DoF, kp 0.2) hold that solution for the whole settle */
        real hp[6]; for (int k = 0; k < 6; k++) hp[k] = (real)t->home_hand_pose[k];
        for (int k = t->robot >= 1 ? 0 : 2; k < 3; k++) hp[k] = clampr(hp[k], t->robot_ws[k][0], t->robot_ws[k][1]);
        ik_targets(m, t, st, hp, qdes);
        for (int k = 0; k < 6; k++) X[6 + k] = (real)t->home_hand_pose[k];
    }
    /* one extra stepSimulation at the end of robot.reset: Panda only in IK mode (panda_env.py:91), iCub always (icub_env.py:151) */
    if (t->use_ik || t->robot >= 1) orc_sim_step_f(m, &p1, st, qdes, kp, kd, fs, NULL);
    for (int i = 0; i < 100; i++) orc_sim_step_f(m, &p1, st, qdes, kp, kd, fs, NULL);
    for (int i = 0; i < 101; i++) orc_sim_step_f(m, prm, st, qdes, kp, kd, fs, NULL);
    /* sample_tg_pose (panda_push_gym_env.py:333-360, icub_push_gym_env.py:375-401) */
    if (t->task >= 1) {
        real tx_min = (real)t->ws_lim[0][0] + (real)0.07, tx_max = (real)t->ws_lim[0][1] - (real)0.07;
        real ty_min = (real)t->ws_lim[1][0], ty_max = (real)t->ws_lim[1][1];
        real tx = ob[0] + (real)0.05, ty = ob[1] + (real)0.05, tz = ob[2];
        if (t->tg_pose_rnd_std > 0) {
            orc_philox4x32((uint32_t)env_id, (uint32_t)(env_id >> 32), episode, 1u, (uint32_t)t->seed, (uint32_t)(t->seed >> 32), r);
            real u1 = (real)((r[0] >> 8) + 1) * (real)(1.0 / 16777216.0), u2 = u01(r[1]);
            real rad = (real)sqrt(-2.0 * log((double)u1)) * (real)t->tg_pose_rnd_std;
            tx = ob[0] + rad * (real)cos(2 * PI * (double)u2);
            ty = ob[1] + rad * (real)sin(2 * PI * (double)u2);
        }
        tx = tx < tx_min ? tx_min : (tx > tx_max ? tx_max : tx);
        ty = ty < ty_min ? ty_min : (ty > ty_max ? ty_max : ty);
        X[0] = tx; X[1] = ty; X[2] = tz;
    }
    X[3] = 0; X[4] = 0; X[5] = (real)episode;
    if (t->robot >= 1 && t->task >= 1) {
        /* iCubPushGymEnv.reset (icub_push_gym_env.py:124-127): distances the normalised reward divides by */
        real pos[3], quat[4], vl[3], d1 = 0, d2 = 0;
        ee_state(m, st, pos, quat, vl);
        for (int k = 0; k < 3; k++) { real a = pos[k] - ob[k], b = ob[k] - X[k]; d1 += a*a; d2 += b*b; }
        X[12] = (real)sqrt((double)d1); X[13] = (real)sqrt((double)d2);
    }
    if (obs) orc_observation(m, t, st, obs);
}

/* in-loop `if self._termination(): break` of apply_action: the termination test without side effects other than the latch */
static int terminated_now(const orc_model* m, const orc_task* t, real* st) {
    real pos[3], quat[4], vl[3];
    real* X = st + OX(m); const real* ob = st + OQ(m);
    ee_state(m, st, pos, quat, vl);
    real d1 = 0, d2 = 0;
    for (int k = 0; k < 3; k++) { real a = pos[k] - ob[k], b = ob[k] - X[k]; d1 += a*a; d2 += b*b; }
    d1 = (real)sqrt((double)d1); d2 = (real)sqrt((double)d2);
    const int cnt = (int)X[3];
    if (t->task == 2) return cnt > t->max_steps;
    const int succ = (t->task >= 1 ? d2 : d1) <= (real)t->target_dist_min;
    if (succ) X[4] = 1;
    return succ || (int)X[4] || cnt > t->max_steps;
}

void orc_env_step(const orc_model* m, const orc_params* prm, const orc_task* t, real* st, const real* action,
                  real* obs, real* reward, real* done) {
    /* step -> apply_action (panda_push_gym_env.py:189-242; icub_reach_gym_env.py:182-246): action_repeat iterations of
     * [targets from the CURRENT state, stepSimulation, break on termination, counter++], then observation / done / reward */
    const int nd = m->ndof;
    real* X = st + OX(m);
    const int reps = t->action_repeat > 1 ? t->action_repeat : 1;
    /* the reference scales the action IN PLACE inside the loop (`action *= 0.05`, panda_push_gym_env.py:199-203,225;
     * icub_reach_gym_env.py:206-212,232), so iteration r applies action * scale^(r+1) */
    real sj = 1, sp = 1, sr = 1;
    for (int rep = 0; rep < reps; rep++) {
    sj *= (real)t->act_scale; sp *= (real)t->ik_pos_scale; sr *= (real)t->ik_rot_scale;
    real qdes[ORC_MAXD], kp[ORC_MAXD], kd[ORC_MAXD];
    hold_targets(t, nd, qdes, kp, kd);
    if (t->use_ik) {
        /* IK branch (panda_push_gym_env.py:197-222, panda_env.py:229-291; icub_reach_gym_env.py:204-230, icub_env.py:262-330):
         * accumulate the scaled action on the commanded hand pose, clip rotation and workspace, solve IK */
        real hp[6];
        for (int k = 0; k < 6; k++) hp[k] = X[6 + k];
        for (int k = 0; k < 3; k++) hp[k] += action[k] * sp;
        if (t->control_orientation)
            for (int k = 3; k < 6; k++) hp[k] = clampr(hp[k] + action[k] * sr, t->eu_lim[k-3][0], t->eu_lim[k-3][1]);
        for (int k = 0; k < 3; k++) hp[k] = clampr(hp[k], t->robot_ws[k][0], t->robot_ws[k][1]);
        for (int k = 0; k < 6; k++) X[6 + k] = hp[k];
        ik_targets(m, t, st, hp, qdes);
    } else
    for (int k = 0; k < t->n_act; k++) {
        const int d = t->act_dof[k], li = m->link_of_dof[d];
        real tgt = st[d] + action[k] * sj;                                 /* :225-230 */
        tgt = tgt < m->lower[li] ? m->lower[li] : (tgt > m->upper[li] ? m->upper[li] : tgt);   /* panda_env.py:303, icub_env.py:347 */
        qdes[d] = tgt; kp[d] = (real)t->kp_act; kd[d] = (real)t->kd_act;
    }
    orc_sim_step(m, prm, st, qdes, kp, kd, NULL);
    if (rep + 1 < reps) {                    /* not the last iteration: `if self._termination(): break` / counter++ here */
        if (terminated_now(m, t, st)) { orc_reward_done(m, t, st, 0, reward, done); orc_observation(m, t, st, obs); return; }
        X[3] += 1;
    }
    }
    orc_reward_done(m, t, st, 1, reward, done);    /* last iteration's termination test + counter, then the final evaluation */
    orc_observation(m, t, st, obs);
}

/* ---- iCub with hands: robot-level interface (icub_env_with_hands.py; icub_env.py:260-361) */
void orc_task_hands(orc_task* t, int right_arm, int use_ik, const int* ctrl_dof, int n_ctrl, const double* home, int ndof) {
    orc_default_task(t, 0);
    t->robot = 2; t->max_steps = 1 << 30; t->target_dist_min = -1.0;      /* no episode logic at the robot level */
    t->ws_lim[0][0] = 0.35; t->ws_lim[0][1] = 0.70; t->ws_lim[1][0] = -0.33; t->ws_lim[1][1] = 0.27;   /* object dropped at (0.5, -0.03), helloworld_icub.py:51 */
    t->ws_lim[2][0] = t->h_table; t->ws_lim[2][1] = t->h_table + 0.3;
    t->robot_ws[0][0] = 0.15; t->robot_ws[0][1] = 0.50; t->robot_ws[1][0] = -0.3; t->robot_ws[1][1] = 0.3;
    t->robot_ws[2][0] = 0.5; t->robot_ws[2][1] = 1.0;                       /* icub_env_with_hands.py:63 */
    for (int k = 0; k < ORC_MAXD; k++) t->home[k] = k < ndof ? home[k] : 0.0;
    t->n_joints_ctrl = n_ctrl;
    for (int k = 0; k < ORC_MAXACT; k++) t->act_dof[k] = k < n_ctrl ? ctrl_dof[k] : -1;
    t->use_ik = use_ik; t->control_orientation = 1; t->ik_absolute = 1;
    t->n_act = use_ik ? 6 : n_ctrl;
    t->ik_pos_scale = 1.0; t->ik_rot_scale = 1.0;
    const double hl[6] = {0.2, 0.3, 0.8, -PI, 0, -PI / 2}, hr[6] = {0.2, -0.3, 0.8, 0, 0, PI / 2};       /* :77-81 */
    for (int k = 0; k < 6; k++) t->home_hand_pose[k] = right_arm ? hr[k] : hl[k];
    const double el[3][2] = {{-1.5 * PI, -PI / 2}, {-PI / 2, PI / 2}, {0, -PI}}, er[3][2] = {{-PI / 2, PI / 2}, {-PI / 2, PI / 2}, {0, PI}};
    for (int a = 0; a < 3; a++) for (int b = 0; b < 2; b++) t->eu_lim[a][b] = right_arm ? er[a][b] : el[a][b];
    const double ol[3] = {-0.011682, 0.051355, 0.000577}, orr[3] = {-0.011682, 0.051682, -0.000577};   /* :159-165 */
    for (int k = 0; k < 3; k++) t->ik_link_offset[k] = right_arm ? orr[k] : ol[k];
}
void orc_hands_set_motors(const orc_params* prm, real* mrec, int n, const int* dofs, const real* targets, double kp, double max_force, double max_vel) {
    /* p.setJointMotorControlArray(POSITION_CONTROL, targetPositions, positionGains[, forces]) of open_hand / pre_grasp / grasp (:167-244);
     * pandaEnv.apply_action_fingers: setJointMotorControl2(targetPosition, force=10, maxVelocity=1) (panda_env.py:218-225) */
    const real fs = max_force > 0 ? (real)(max_force * prm->dt / prm->max_motor_impulse) : (real)1;
    for (int k = 0; k < n; k++) {
        mrec[dofs[k]] = targets[k]; mrec[ORC_MAXD + dofs[k]] = (real)kp; mrec[2 * ORC_MAXD + dofs[k]] = fs;
        mrec[3 * ORC_MAXD + dofs[k]] = max_vel > 0 ? (real)max_vel : 0;
    }
}
void orc_hands_settle(const orc_model* m, const orc_params* prm, const orc_task* t, real* st, const real* mrec, int n) {
    real kd[ORC_MAXD];
    for (int k = 0; k < ORC_MAXD; k++) kd[k] = (real)t->kd_hold;
    for (int i = 0; i < n; i++) sim_step_fv(m, prm, st, mrec, mrec + ORC_MAXD, kd, mrec + 2 * ORC_MAXD, mrec + 3 * ORC_MAXD, NULL);
}
/* the command half of apply_action(action, max_vel) alone (icub_env.py:260-361, panda_env.py:227-310): writes the motor record */
void orc_hands_apply_action(const orc_model* m, const orc_params* prm, const orc_task* t, real* st, real* mrec, const real* action, double max_vel) {
    (void)prm;
    real* X = st + OX(m);
    real* qdes = mrec; real* kp = mrec + ORC_MAXD; real* fs = mrec + 2 * ORC_MAXD; real* vm = mrec + 3 * ORC_MAXD;
    const real mv = max_vel > 0 ? (real)max_vel : 0;
    if (t->use_ik) {
        real hp[6];
        for (int k = 0; k < 3; k++) hp[k] = (t->robot >= 1 || k == 2) ? clampr(action[k], t->robot_ws[k][0], t->robot_ws[k][1]) : action[k];   /* pandaEnv clips z only */
        for (int k = 3; k < 6; k++) hp[k] = t->control_orientation ? clampr(action[k], t->eu_lim[k-3][0], t->eu_lim[k-3][1]) : X[6 + k];
        for (int k = 0; k < 6; k++) X[6 + k] = hp[k];
        if (t->robot >= 1) {
            /* every joint is commanded (setJointMotorControlArray / per-joint calls over all joints, gain 0.2[, maxVelocity]) */
            for (int k = 0; k < m->ndof; k++) { qdes[k] = (real)t->home[k]; kp[k] = (real)t->kp_hold; fs[k] = 1; vm[k] = mv; }
            ik_targets(m, t, st, hp, qdes);
        } else {
            /* pandaEnv: all 9 joints with gain 0.2; with max_vel only the 7 arm joints, PyBullet's default gain 0.1 [EXT-UNVERIFIED] */
            real sol[ORC_MAXD];
            for (int k = 0; k < m->ndof; k++) sol[k] = st[k];
            ik_targets(m, t, st, hp, sol);
            const int nj = mv > 0 ? 7 : m->ndof;
            for (int k = 0; k < nj; k++) { qdes[k] = sol[k]; kp[k] = mv > 0 ? (real)0.1 : (real)t->kp_hold; fs[k] = 1; vm[k] = mv; }
        }
    } else
    for (int k = 0; k < t->n_act; k++) {
        const int d = t->act_dof[k], li = m->link_of_dof[d];
        real tgt = action[k];
        tgt = tgt < m->lower[li] ? m->lower[li] : (tgt > m->upper[li] ? m->upper[li] : tgt);     /* icub_env.py:347, panda_env.py:303 */
        qdes[d] = tgt; kp[d] = (real)t->kp_act; fs[d] = 1; vm[d] = t->robot >= 1 ? mv : 0;       /* the iCub's joint branch passes maxVelocity */
    }
}
/* pandaEnv used alone: the robot-level interface of the Panda (panda_env.py:25-91, 195-365), scene of helloworld_panda.py:72-85 */
void orc_task_panda_arm(orc_task* t, int use_ik, int control_orientation) {
    orc_default_task(t, 0);
    t->robot = 0; t->max_steps = 1 << 30; t->target_dist_min = -1.0;
    t->ws_lim[0][0] = 0.35; t->ws_lim[0][1] = 0.70;
    t->robot_ws[2][0] = 0.65; t->robot_ws[2][1] = 1.5;
    t->n_joints_ctrl = 9;
    for (int k = 0; k < ORC_MAXACT; k++) t->act_dof[k] = k < 9 ? k : -1;
    t->use_ik = use_ik; t->control_orientation = control_orientation; t->ik_absolute = 1;
    t->n_act = use_ik ? (control_orientation ? 6 : 3) : 9;
    t->ik_pos_scale = 1.0; t->ik_rot_scale = 1.0;
}
void orc_hands_step(const orc_model* m, const orc_params* prm, const orc_task* t, real* st, real* mrec, const real* action,
                    real* obs, real* reward, real* done) {
    /* apply_action (absolute commands, no max_vel), one stepSimulation, observation */
    real kd[ORC_MAXD];
    for (int k = 0; k < ORC_MAXD; k++) kd[k] = (real)t->kd_hold;
    orc_hands_apply_action(m, prm, t, st, mrec, action, -1.0);
    sim_step_fv(m, prm, st, mrec, mrec + ORC_MAXD, kd, mrec + 2 * ORC_MAXD, mrec + 3 * ORC_MAXD, NULL);
    orc_reward_done(m, t, st, 1, reward, done);
    orc_observation(m, t, st, obs);
}

void orc_batch_reset(const orc_model* m, const orc_params* prm, const orc_task* t, int n, uint64_t env_id0,
                     real* states, real* obs) {
    const int od = orc_obs_dim(t, m), sf = orc_state_floats(m);
    for (int e = 0; e < n; e++) orc_env_reset(m, prm, t, env_id0 + (uint64_t)e, 0, states + (size_t)e * sf, obs ? obs + (size_t)e * od : NULL);
}
/* orc_batch_step that also reports, per env, how many sweeps the solver ran and after how many Bullet's residual test with PyBullet's
 * documented default threshold (1e-7) would have ended the loop (orc_step_info.sweeps_used / sweeps_to_1e7) */
void orc_batch_step_sweeps(const orc_model* m, const orc_params* prm, const orc_task* t, int n, real* states,
                           const real* actions, real* out, int* sweeps_used, int* sweeps_to_1e7) {
    const int od = orc_obs_dim(t, m), sf = orc_state_floats(m);
    for (int e = 0; e < n; e++) {
        real* o = out + (size_t)e * (od + 2);
        g_last_sweeps[0] = g_last_sweeps[1] = 0;
        orc_env_step(m, prm, t, states + (size_t)e * sf, actions + (size_t)e * t->n_act, o, o + od, o + od + 1);
        sweeps_used[e] = g_last_sweeps[0]; sweeps_to_1e7[e] = g_last_sweeps[1];
    }
}
void orc_batch_step(const orc_model* m, const orc_params* prm, const orc_task* t, int n, real* states,
                    const real* actions, real* out) {
    const int od = orc_obs_dim(t, m), sf = orc_state_floats(m);
    for (int e = 0; e < n; e++) {
        real* o = out + (size_t)e * (od + 2);
        orc_env_step(m, prm, t, states + (size_t)e * sf, actions + (size_t)e * t->n_act, o, o + od, o + od + 1);
    }
}
