/* pbre_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (double precision by default, -DORC_FLOAT for a float build)
 * of the env.step() hot path of hsp-iit/pybullet-robot-envs for the Panda and iCub
 * reach/push tasks and the robot-level interface of the iCub with hands.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libpbre.so) never does.
 *
 * PARITY UNPINNED for the physics: the arithmetic of `p.stepSimulation()`
 * lives in the third-party PyBullet/Bullet3 C++ extension (reference
 * requirements.txt:2, unpinned; setup.py:35 hints pybullet==2.5.0), which is not
 * vendored, not installed, and the reference ships no tests/golden vectors
 * (SURVEY §8c).  This file restates Bullet's published multibody algorithm
 * (Featherstone ABA -> constraint rows -> projected Gauss-Seidel -> semi-
 * implicit Euler) from the call sites in the reference; every Bullet-internal
 * choice is tagged [EXT-UNVERIFIED] in pbre_oracle.c.  The Python-glue part
 * (observation layout, reward, termination, sampling) IS pinned: it is checked
 * against vectors captured from the reference's own classes
 * (tests/golden/, tools/make_golden.py).
 */
#ifndef PBRE_ORACLE_H
#define PBRE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef ORC_FLOAT
typedef float real;
#else
typedef double real;
#endif

#define ORC_MAXL 80      /* links (fixed joints kept as 0-DoF links) */
#define ORC_MAXD 64      /* joint DoF */
#define ORC_MAXS 64      /* collision spheres */
#define ORC_NC_OT 4      /* object-table contact slots */
#define ORC_NC_RO 2      /* robot-object contact slots (6 for the iCub with hands: 5 fingertips + palm) */
#define ORC_NC_RO_HANDS 6
#define ORC_NC_RO_PANDA 4 /* the Panda (<= 9 DoF) */
#define ORC_NC_RT 2      /* robot-table contact slots */
#define ORC_NC (ORC_NC_OT + ORC_NC_RO_HANDS + ORC_NC_RT)
#define ORC_NTIP 5       /* fingertips of the controlled hand (icub_env_with_hands.py:248) */
#define ORC_STATE 48     /* floats per env state record of a <= 9-DoF robot (see include/pbre.h) */
#define ORC_MAXACT 64    /* controlled joints */
#define ORC_MAXHV 32     /* vertices of a convex-hull object (include/pbre.h: PBRE_HULL_MAXV) */
/* State record layout (include/pbre.h): three lane records Q | V | X.  Q and V are W floats wide (W = 16 for robots
 * with <= 9 DoF, 32 for <= 20 DoF, 64 otherwise), X is 16:  Q[0..nd) q, Q[nd..nd+3) object position, Q[nd+3..nd+7) object quaternion;
 * V[0..nd) qd, V[nd..nd+6) object twist;  X[0..2] target, X[3] counter, X[4] terminated, X[5] episode,
 * X[6..11] commanded hand pose, X[12] initial hand-object distance, X[13] initial object-target distance.
 * iCub with hands (W = 128): Q[nd+7..nd+12) mean normal force per fingertip, Q[nd+12] tips in contact, Q[nd+13] robot-object contact points. */

typedef struct {
    int nl, ndof, ee_link, ns, fixed_base;
    real base_pos[3], base_R[9];
    int parent[ORC_MAXL], jtype[ORC_MAXL], dof[ORC_MAXL];
    real axis[ORC_MAXL][3], Xp[ORC_MAXL][3], XR[ORC_MAXL][9];
    real mass[ORC_MAXL], com[ORC_MAXL][3], inertia[ORC_MAXL][9];
    real lower[ORC_MAXL], upper[ORC_MAXL], damping[ORC_MAXL], friction[ORC_MAXL];
    int s_link[ORC_MAXS];
    real s_c[ORC_MAXS][3], s_r[ORC_MAXS], s_mu[ORC_MAXS];
    int s_tip[ORC_MAXS];           /* fingertip slot + 1 of the sphere (0: not a fingertip) */
    int ntip;                      /* > 0: the model reports fingertip contact forces (iCub with hands) */
    int link_of_dof[ORC_MAXD];
    int passive[ORC_MAXL];         /* 1: a joint the inverse kinematics does not move (RobotTable link record [37]: the virtual joints of a soft-pinned floating base) */
    real max_force[ORC_MAXL];      /* > 0: force bound (N) of the joint's motor row instead of max_motor_impulse / dt -- the base constraint's maxForce on the virtual
                                    * joints of a soft-pinned floating base (link record [35] of a joint with [37] set; reference icub_env.py:95-101, PyBullet default 500) */
} orc_model;

/* numeric parameters of the simulated scene + task; mirrors pbre_params in include/pbre.h */
typedef struct {
    double dt, gravity_z;
    int solver_iters;
    double erp, linear_slop, contact_margin;
    double lin_damping, ang_damping, max_coord_vel;
    double max_motor_impulse;      /* force * dt */
    double limit_max_impulse;
    double table_c[3], table_h[3]; /* table-top box centre / half extents */
    double table_mu, ground_z;
    double obj_h[3], obj_mass, obj_inertia[3], obj_mu;
    int flags;                     /* ORC_F_* */
    int implicit_joint_damping;    /* 1: (M + dt C) dv = dt (tau - C v) instead of the explicit damping torque (include/pbre.h) */
    int obj_shape;                 /* object primitive (include/pbre.h PBRE_SHAPE_*): 0 box (half extents obj_h), 1 sphere (radius obj_h[0]),
                                      2 cylinder about its local z axis (radius obj_h[0], half height obj_h[2]), 3 convex hull of obj_hull[] */
    double solver_residual_threshold;  /* Bullet's btContactSolverInfo::m_leastSquaresResidualThreshold: the sweep loop is left once the
                                      largest squared velocity-level change of a row within one sweep is <= this value.  0 (the default here,
                                      and what the engine implements): all solver_iters sweeps unless a sweep changes nothing at all.
                                      [EXT-UNVERIFIED] PyBullet documents solverResidualThreshold with default 1e-7; see orc_step_info.sweeps_* */
    int obj_hull_n;                /* obj_shape 3 (PBRE_SHAPE_HULL): number of hull vertices (4..ORC_MAXHV) */
    double obj_hull[ORC_MAXHV][3]; /* ... and the vertices in the object's frame (origin = centre of mass).  obj_h then holds the half extents
                                      of the hull's bounding box (rest-height guess of the reset, as for the primitives) */
} orc_params;

#define ORC_F_NO_OBJECT 1   /* object frozen and contact-free (reset phase 1; reach config 2) */

typedef struct {
    int ncontacts;
    int  type[ORC_NC];      /* 0 obj-table, 1 robot-obj, 2 robot-table */
    int  link[ORC_NC], idx[ORC_NC];
    real n[ORC_NC][3], pA[ORC_NC][3], pB[ORC_NC][3], dist[ORC_NC], mu[ORC_NC];
    real lambda_n[ORC_NC], lambda_f1[ORC_NC], lambda_f2[ORC_NC];
    real motor_impulse[ORC_MAXD];
    real qdd[ORC_MAXD];          /* unconstrained joint accelerations (ABA) */
    real obj_acc[6];
    real residual;
    int  sweeps_used;            /* sweeps the solver ran (= solver_iters unless solver_residual_threshold ended it) */
    int  sweeps_to_1e7;          /* first sweep after which the largest squared velocity-level row change was <= 1e-7 (PyBullet's documented
                                    default threshold), solver_iters + 1 if never: how early real PyBullet would presumably leave the loop */
    real last_sq_residual;       /* that quantity for the last sweep run */
} orc_step_info;

#ifdef __cplusplus
extern "C" {
#endif
int  orc_sizeof_real(void);
int  orc_model_from_table(const double* tbl, size_t n, orc_model* m);
void orc_default_params(orc_params* p);
/* forward kinematics: world rotation (row-major 3x3) and origin of every link frame */
void orc_fk(const orc_model* m, const real* q, real* R /*[nl][9]*/, real* p /*[nl][3]*/);
/* joint-space mass matrix by ndof impulse-response ABA passes (tests) and bias accelerations */
void orc_mass_matrix_inverse(const orc_model* m, const orc_params* prm, const real* q, real* Minv /*[ndof][ndof]*/);
void orc_forward_dynamics(const orc_model* m, const orc_params* prm, const real* q, const real* qd,
                          const real* tau, real* qdd);
/* one stepSimulation(): state record [48], motor targets/gains per DoF */
void orc_sim_step(const orc_model* m, const orc_params* prm, real* state,
                  const real* q_des, const real* kp, const real* kd, orc_step_info* info);
/* same with a per-DoF scale of the motor impulse bound (setJointMotorControl `force` / default force); NULL = 1 */
void orc_sim_step_f(const orc_model* m, const orc_params* prm, real* state,
                    const real* q_des, const real* kp, const real* kd, const real* fscale, orc_step_info* info);

/* ---- task layer (reference Python glue restated) ---- */
typedef struct {
    int task;             /* 0 reach, 1 push */
    int max_steps;
    double target_dist_min;
    double obj_pose_rnd_std, tg_pose_rnd_std;
    double ws_lim[3][2];     /* world workspace (object) */
    double h_table;
    double home[ORC_MAXD];
    double act_scale;        /* 0.05 */
    double kp_act, kd_act, kp_hold, kd_hold;
    int n_act;               /* controlled joints (7); 6 in IK mode (dx,dy,dz,droll,dpitch,dyaw) */
    uint64_t seed;
    int use_ik;              /* 1: Cartesian control through inverse kinematics (use_IK=1) */
    double ik_damping, ik_residual; int ik_max_iters;
    double home_hand_pose[6];
    double robot_ws[3][2];   /* robot workspace used to clip the hand pose */
    int robot;               /* 0 Panda, 1 iCub: observation / reward / reset variants of the iCub envs */
    int act_dof[ORC_MAXACT]; /* DoF index of controlled joint k (k < n_joints_ctrl) */
    int n_joints_ctrl;       /* joints driven in joint mode and reported in the observation (Panda: 7 driven, all 9 observed) */
    int control_orientation; /* IK mode: 1 = 6-D action (pose), 0 = 3-D action (position, home orientation kept) */
    double ik_pos_scale, ik_rot_scale;
    double eu_lim[3][2];
    double ik_link_offset[3];/* hand COM frame -> link frame (icub_env.py:252-258) */
    int reward_type;         /* iCub push: 0 / 1 (icub_push_gym_env.py:353-373) */
    int action_repeat;       /* simulation steps per env.step() (apply_action loop, panda_push_gym_env.py:193-242); 0 = 1 */
    int ik_absolute;         /* IK actions are absolute hand poses (robot-level apply_action, icub_env.py:262-300) */
} orc_task;

void orc_default_task(orc_task* t, int task);
int  orc_obs_dim(const orc_task* t, const orc_model* m);
int  orc_state_floats(const orc_model* m);
/* set the iCub variants (robot=1): control_arm 0 left / 1 right, controlled DoF list, home pose per DoF */
void orc_task_icub(orc_task* t, int task, int right_arm, int use_ik, int control_orientation, const int* ctrl_dof, const double* home, int ndof);
/* iCub with hands (robot = 2, icub_env_with_hands.py): robot-level interface.  mrec = per-DoF persistent motor record
 * target[ORC_MAXD] | kp[ORC_MAXD] | force scale[ORC_MAXD] (PyBullet motors keep their last command) */
void orc_task_hands(orc_task* t, int right_arm, int use_ik, const int* ctrl_dof, int n_ctrl, const double* home, int ndof);
void orc_hands_reset(const orc_model* m, const orc_params* prm, const orc_task* t, uint64_t env_id, uint32_t episode,
                     real* state, real* mrec, real* obs);
void orc_hands_step(const orc_model* m, const orc_params* prm, const orc_task* t, real* state, real* mrec,
                    const real* action, real* obs, real* reward, real* done);
void orc_hands_settle(const orc_model* m, const orc_params* prm, const orc_task* t, real* state, const real* mrec, int n);
void orc_hands_set_motors(const orc_params* prm, real* mrec, int n, const int* dofs, const real* targets, double kp, double max_force, double max_vel);
void orc_hands_apply_action(const orc_model* m, const orc_params* prm, const orc_task* t, real* state, real* mrec, const real* action, double max_vel);
void orc_task_panda_arm(orc_task* t, int use_ik, int control_orientation);
void orc_observation(const orc_model* m, const orc_task* t, const real* state, real* obs);
void orc_reward_done(const orc_model* m, const orc_task* t, real* state, int pre_increment,
                     real* reward, real* done);
void orc_env_reset(const orc_model* m, const orc_params* prm, const orc_task* t, uint64_t env_id,
                   uint32_t episode, real* state, real* obs);
void orc_env_step(const orc_model* m, const orc_params* prm, const orc_task* t, real* state,
                  const real* action, real* obs, real* reward, real* done);
/* batched convenience loops (tests, cpu_baseline) */
void orc_batch_reset(const orc_model* m, const orc_params* prm, const orc_task* t, int n, uint64_t env_id0,
                     real* states, real* obs);
void orc_batch_step_sweeps(const orc_model* m, const orc_params* prm, const orc_task* t, int n, real* states,
                           const real* actions, real* out, int* sweeps_used, int* sweeps_to_1e7);
void orc_batch_step(const orc_model* m, const orc_params* prm, const orc_task* t, int n, real* states,
                    const real* actions, real* out /*[n][obs_dim+2]*/);

/* damped-least-squares IK of the end effector (restates p.calculateInverseKinematics call sites panda_env.py:269-272) */
int  orc_ik(const orc_model* m, const orc_task* t, const real* q_start, const real* pos, const real* euler, real* q_out);
/* counter-based RNG shared (by specification) with the device path */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);
/* Bullet pure-math helpers used by the observation code */
void orc_quat_from_euler(const real e[3], real q[4]);
void orc_euler_from_quat(const real q[4], real e[3]);
/* sweeps_used, sweeps_to_1e7 (orc_step_info) of the calling thread's most recent simulation step */
void orc_last_sweeps(int out[2]);

#ifdef __cplusplus
}
#endif
#endif
