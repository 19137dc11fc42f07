/* pbre.h -- C-ABI of libpbre.so, the MI355X batched rollout engine.
 *
 * Drop-in boundary.  The reference (hsp-iit/pybullet-robot-envs) has no FFI of its
 * own: its hot path calls the third-party `pybullet` Python module once per env and
 * per query.  This ABI is what the reference's Gym classes bind instead (ctypes stub
 * in INTEGRATION.md); each entry point names the reference call sites it replaces.
 * `R/` = pybullet_robot_envs/ in the reference tree.
 *
 * Conventions: plain C types only; caller owns every buffer it passes; the ctx owns
 * all device memory.  Every function returns 0 on success or a negative PBRE_E_* code
 * and never throws; pbre_last_error() gives the message.  A ctx is used by one host
 * thread at a time and is bound to one GPU (one process per GPU; multi-GPU sharding is
 * done above this ABI with one ctx per rank and env_id_base = rank * num_envs).
 *
 * Per-env state record: three lane records Q[W] | V[W] | X[16] floats; W = 16 for robots with <= 9 DoF
 * (Panda: 48 floats), W = 32 for <= 20 DoF (the iCub as simulated, without its legs: 80 floats), W = 64 for <= 32 DoF, W = 128 for <= 60
 * (the iCub with hands: 272 floats); the Panda's and the iCub's robot-level engines (robot_level = 1) use W = 32 (80 floats); nd = number of DoF:
 *   (nd below: the DoF LANES of the kernel shape -- 9, 20, 32 or 60 -- which every shipped model fills exactly except the soft-pinned
 *   floating-base iCub, 26 DoF on the 32-lane shape: its joints are Q[0..26), lanes 26..31 are unused, the object follows at 32)
 *   Q[0..nd)  joint positions         Q[nd..nd+3)  object position   Q[nd+3..nd+7) object quaternion (x,y,z,w)
 *   V[0..nd)  joint velocities        V[nd..nd+3)  object lin. vel.  V[nd+3..nd+6) object ang. vel.
 *   X[0..2]  push target   X[3] step counter  X[4] terminated flag  X[5] episode   X[6..11] commanded hand pose (IK mode)
 *   X[12], X[13] hand-object / object-target distance at reset (iCub push reward)   X[14] left the apply_action loop (action_repeat > 1)
 *   Panda task envs: X[12], X[13], X[15] per-env object mass / lateral friction / 1 + linear damping, V[15] 1 + the robot links' linear damping
 *   (pbre_set_physics_per_env; 0 = batch value)
 * Robot-level engines (iCub with hands: W = 128, nd = 60; Panda: W = 32, nd = 9, fingertip slots 0 / 1 = left / right finger): additionally Q[nd+7..nd+12) mean normal force on each fingertip of the controlled hand,
 *   Q[nd+12] fingertips in contact with the object, Q[nd+13] robot-object contact points (check_contact_fingertips /
 *   check_collision, icub_env_with_hands.py:246-318); these 7 values are also the tail of the observation.
 *
 * RobotTable (pbre_config.robot_table): float64 array, little endian
 *   [0] magic 1346523717 ('PBRE')  [1] version 1  [2] n_links  [3] n_dof  [4] ee_link  [5] n_spheres
 *   [6..8] base position  [9..17] base rotation (row major)  [18] fixed_base  [19..23] reserved
 *   then n_links records of 40: parent, jtype(0 fixed,1 revolute,2 prismatic), axis[3], origin_xyz[3],
 *        origin_R[9], mass, com[3], inertia[9] (about COM, link axes), lower, upper, damping,
 *        dof_index(-1 fixed), lateral_friction, effort, velocity, ik_passive (1: a joint the inverse kinematics must not move -- the
 *        virtual joints of a soft-pinned floating base, model/table.py: float_base), reserved[2]
 *   then n_spheres records of 8: link, centre[3], radius, friction, fingertip slot + 1 (0: not a fingertip), reserved
 */
#ifndef PBRE_H
#define PBRE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBRE_STATE_FLOATS 48       /* Panda; see pbre_state_floats() */

enum { PBRE_OK = 0, PBRE_E_ARG = -1, PBRE_E_TABLE = -2, PBRE_E_DEVICE = -3, PBRE_E_UNSUPPORTED = -4 };
enum { PBRE_ROBOT_PANDA_ARM = 3,   /* pbre_default_config only: the Panda with the robot-level interface -- fills the config with
                                     robot = PBRE_ROBOT_PANDA, robot_level = 1 and the scene of examples/helloworlds/helloworld_panda.py */
       PBRE_ROBOT_PANDA = 0,
       PBRE_ROBOT_ICUB = 1,       /* icub_model.sdf: observation / reward / reset variants of R/envs/icub_envs */
       PBRE_ROBOT_ICUB_HANDS = 2 };   /* icub_model_with_hands.sdf (R/envs/icub_envs/icub_env_with_hands.py): robot-level interface --
                                     absolute joint / hand-pose commands, persistent finger motors (pbre_set_motors), fingertip
                                     contact forces appended to the observation */
enum { PBRE_SHAPE_BOX = 0, PBRE_SHAPE_SPHERE = 1, PBRE_SHAPE_CYLINDER = 2, PBRE_SHAPE_HULL = 3 };   /* pbre_physics.obj_shape (reference world_env.py:18-25, 179-216: obj_name) */
enum { PBRE_TASK_REACH = 0, PBRE_TASK_PUSH = 1,
       PBRE_TASK_PUSH_GOAL = 2 };   /* pandaPushGymGoalEnv termination/reward (R/envs/panda_envs/panda_push_gym_goal_env.py:89-122) */
enum { PBRE_F_NO_OBJECT = 1,      /* object frozen and contact-free (BASELINE config 2) */
       PBRE_F_AUTO_RESET = 2,     /* done envs are re-initialised from the settled snapshot at the next step */
       PBRE_F_FORCE_GENERAL = 4,  /* step every env with the general 16-lane row kernel (disable the lane-per-env fast path) */
       /* Envs with robot contacts / joints at a limit ("complex") are stepped by one of two kernels; the engine picks by
        * their number (row kernel while they are few: lowest latency; lane-per-env k_fast_rc when many: highest throughput).
        * These two flags pin the choice (validation, A/B): */
       PBRE_F_COMPLEX_ROWS = 8, PBRE_F_COMPLEX_LANES = 16,
       /* Panda, envs without robot contact: the 9 joint-motor rows (p.setJointMotorControl2, panda_env.py:305-310) are a linear
        * iteration there and their `solver_iters` sweeps are evaluated in closed form (a matrix power); this flag runs
        * them as Bullet does, row by row (validation, A/B) */
       PBRE_F_SEQ_MOTORS = 32,
       /* ... and the tail of the sweeps over a resting cube's 12 object-table rows (unclamped, a sweep is one pass of
        * Kaczmarz's method -- an affine map of the object's twist --, applied as a matrix power where a per-env bound proves that no
        * clamp can bind); this flag runs all of them row by row (validation, A/B) */
       PBRE_F_SEQ_OBJECT = 64 };

typedef struct pbre_ctx pbre_ctx;

/* Simulation constants.  Defaults (pbre_default_config) are the values the reference
 * hard-codes or inherits from PyBullet: R/envs/panda_envs/panda_push_gym_env.py:39 (dt),
 * :122 (150 solver iterations), :126 (gravity); panda_env.py:76,308 (motor gains). */
typedef struct {
    double dt, gravity_z;
    int32_t solver_iters;
    double erp, linear_slop, contact_margin;
    double lin_damping, ang_damping, max_coord_vel;
    double max_motor_impulse, limit_max_impulse;
    double table_c[3], table_h[3], table_mu, ground_z;
    double obj_h[3], obj_mass, obj_inertia[3], obj_mu;
    int32_t implicit_joint_damping;   /* 0: the joint damping torque -c qd is an explicit force (the restated Bullet step; stable while
                                         c dt < 2 x the joint's effective inertia).  1: implicit, (M + dt C) dv = dt (tau - C v): needed
                                         by the iCub's finger joints (c = 1 on links of inertia 1e-3) once their motors are force-limited
                                         (grasp, force 10) -- default for PBRE_ROBOT_ICUB_HANDS only.  Not available on the Panda's
                                         lane-per-env kernels (PBRE_E_UNSUPPORTED). */
    int32_t obj_shape;                /* PBRE_SHAPE_*: the object's collision primitive.  BOX: half extents obj_h.  SPHERE: radius obj_h[0]
                                         (YcbTennisBall, pear, strawberry stand-ins: they roll).  CYLINDER: about the object's local z,
                                         radius obj_h[0], half height obj_h[2] (the cans, duck_vhacd).  Round objects are stepped by the
                                         per-env object solver / the lane-group kernels (every engine), never by k_fast's in-line cube rows. */
    double  solver_residual_threshold; /* PyBullet's setPhysicsEngineParameter(solverResidualThreshold=...), Bullet's
                                         btContactSolverInfo::m_leastSquaresResidualThreshold: an env leaves the sweep loop after the first
                                         sweep whose largest squared velocity-level row change, max over the rows of (delta impulse /
                                         jacDiagABInv)^2, is <= this value.  0 (default): all `solver_iters` sweeps -- the strict reading of
                                         `numSolverIterations=150` (R/envs/panda_envs/panda_push_gym_env.py:122), which leaves the threshold
                                         at PyBullet's default; PyBullet documents that default as 1e-7 [EXT-UNVERIFIED: no PyBullet on any
                                         box so far], so a PyBullet-pinned comparison may need 1e-7 here.  With a value > 0 every env's
                                         constraint system is swept as ONE system (no closed form of the motor block, no split of an env
                                         over two waves, no lane-per-env iCub pipeline): the exit test is a maximum over all of its rows.
                                         The sweeps each env ran in the last step: pbre_get_sweeps. */
} pbre_physics;

typedef struct {
    int32_t robot;             /* PBRE_ROBOT_* */
    int32_t task;              /* PBRE_TASK_* */
    int32_t num_envs;          /* envs owned by this ctx (this GPU) */
    int32_t device_id;         /* HIP device ordinal */
    uint64_t env_id_base;      /* global id of local env 0: RNG streams are keyed by global id, so results do
                                  not depend on how the batch is sharded */
    uint64_t seed;
    int32_t use_ik;            /* 0: joint control (R/__init__.py:62).  1: Cartesian control through inverse kinematics, Euler
                                  angles: act_dim = 6 (dx,dy,dz,droll,dpitch,dyaw) with control_orientation=1
                                  (R/envs/panda_envs/panda_push_gym_env.py:197-222), 3 (dx,dy,dz) with 0 (iCub ids) */
    int32_t num_controlled_joints;   /* joints driven by the action in joint mode: 7 Panda, 10 iCub (torso + arm) */
    int32_t action_repeat;     /* simulation steps per pbre_step: the apply_action loop (panda_push_gym_env.py:193-242); 1 in every registered id */
    int32_t max_steps;         /* 1000 */
    int32_t flags;             /* PBRE_F_* */
    double  obj_pose_rnd_std, tg_pose_rnd_std;
    double  target_dist_min;   /* 0.1 push / 0.03 reach */
    double  act_scale;         /* 0.05 rad per unit action (panda_push_gym_env.py:225) */
    double  kp_act, kd_act, kp_hold, kd_hold;
    double  ws_lim[3][2];      /* world (object) workspace, world_env.py:72 */
    double  h_table;
    double  home[64];          /* initial joint positions per DoF, panda_env.py:19-23 / icub_env.py:19-41 */
    pbre_physics phys;
    /* use_ik = 1: damped-least-squares IK (replaces p.calculateInverseKinematics, panda_env.py:269-272) */
    double  ik_damping, ik_residual; int32_t ik_max_iters;   /* 0.1 [EXT-UNVERIFIED], 1e-3, 100 */
    double  home_hand_pose[6];  /* panda_env.py:85-88 */
    double  robot_ws[3][2];     /* robot workspace used to clip the hand pose (panda_env.py:37, z-min set by the task env) */
    int32_t control_orientation;/* IK mode: 1 = the action carries droll,dpitch,dyaw; 0 = home orientation is kept (icub_env.py:281-283) */
    int32_t reward_type;        /* iCub push: 0 / 1 (icub_push_gym_env.py:353-373) */
    int32_t num_joints_ctrl;    /* controlled joints = observed joints of the iCub (10); Panda: = num_controlled_joints */
    int32_t act_dof[64];        /* DoF index of controlled joint k, in the reference's _joints_to_control order (icub_env.py:127-138) */
    double  ik_pos_scale, ik_rot_scale;   /* hand-pose increment per unit action (panda_push_gym_env.py:200-203; icub_reach_gym_env.py:206-212) */
    double  eu_lim[3][2];       /* Euler limits of the commanded hand orientation (panda_env.py:38, icub_env.py:63-74) */
    double  ik_link_offset[3];  /* hand COM frame -> hand link frame (icub_env.py:252-258); 0 for the Panda */
    int32_t ik_absolute;        /* 1: IK actions are absolute hand poses (robot-level apply_action, icub_env.py:262-300) instead of
                                   scaled increments accumulated by the task env; set for the robot-level interfaces */
    int32_t robot_level;        /* 1: the robot-level interface of pandaEnv / iCubEnv / iCubHandsEnv used without a task env (R/envs/panda_envs/
                                   panda_env.py:195-365, R/envs/icub_envs/icub_env.py:259-360, icub_env_with_hands.py): absolute commands, persistent
                                   POSITION_CONTROL motors (pbre_set_motors, pbre_apply_action: target | gain | force | max velocity
                                   per joint), fingertip contact statistics appended to the observation (robots with fingers), no episode logic.  Always
                                   1 for PBRE_ROBOT_ICUB_HANDS; PBRE_ROBOT_PANDA with 1 = what pbre_default_config(PBRE_ROBOT_PANDA_ARM) fills */
    const double* robot_table; size_t robot_table_len;   /* number of doubles */
} pbre_config;

/* Fills *cfg with the reference's defaults for (robot, task); caller then sets num_envs, robot_table, ... */
int pbre_default_config(pbre_config* cfg, int32_t robot, int32_t task);

/* replaces: p.connect + pandaEnv.__init__/reset (R/envs/panda_envs/panda_env.py:25-91: loadURDF, per-joint
 * POSITION_CONTROL motors) + WorldEnv.__init__/reset (R/envs/world_envs/world_env.py:35-84) for a whole batch */
int pbre_create(const pbre_config* cfg, pbre_ctx** out);
void pbre_destroy(pbre_ctx* ctx);
const char* pbre_last_error(const pbre_ctx* ctx);   /* borrowed; ctx may be NULL for create errors */

int pbre_dims(const pbre_ctx* ctx, int32_t* obs_dim, int32_t* act_dim, int32_t* num_envs);
/* floats per env state record (48 Panda, 80 iCub, 272 iCub with hands) */
int pbre_state_floats(const pbre_ctx* ctx);

/* replaces: pandaPushGymEnv.reset -> reset_simulation (panda_push_gym_env.py:105-148: resetSimulation,
 * robot.reset, 100 x stepSimulation, world.reset incl. WorldEnv._sample_pose (world_env.py:145-176),
 * 100 + 1 x stepSimulation) + sample_tg_pose (:333-360) + get_extended_observation.
 * env_mask: NULL = all envs, else num_envs bytes (non-zero = reset that env).
 * obs_out: NULL or host [num_envs][obs_dim] raw (unscaled) observation, float32. */
int pbre_reset(pbre_ctx* ctx, const uint8_t* env_mask, float* obs_out);

/* Fast reset of the envs selected by env_mask (num_envs bytes) from the settled snapshot of the last full pbre_reset: what
 * PBRE_F_AUTO_RESET does inside the step, as an entry point -- one small kernel instead of the 201 settle launches of a masked
 * pbre_reset.  The settled state of reset_simulation is invariant under the sampled object x, y, yaw (flat table, vertical drop), so
 * the new episode starts from the recorded settled robot pose / object height with freshly sampled object pose and target: within
 * 2e-5 (Panda) / 5e-5 (iCub) of the explicit reset of the same episode.  Needs one full pbre_reset before.  Task envs only. */
int pbre_reset_snapshot(pbre_ctx* ctx, const uint8_t* env_mask, float* obs_out);

/* replaces: pandaPushGymEnv.step (panda_push_gym_env.py:244-255) = apply_action (:189-242: action*0.05,
 * clip to joint limits panda_env.py:303, 7 x setJointMotorControl2 :305-310, p.stepSimulation :236,
 * _termination :239, counter :242) + get_extended_observation (:150-187) + _termination (:301-316) +
 * _compute_reward (:318-331), for every env.  actions: host [num_envs][act_dim] float32.
 * out: host [num_envs][obs_dim+2] float32 = raw observation | reward | done.  Synchronous. */
int pbre_step(pbre_ctx* ctx, const float* actions, float* out);

/* The same step PIPELINED across calls (round 6; SURVEY 8(d)'s metric counts upload + kernels + download): pbre_step_async enqueues the
 * upload of `actions`, the step and the download of its rows into `out` on three streams and returns; pbre_step_wait blocks until the
 * rows of the OLDEST step not yet waited for are in `out`.  At most two steps may be in flight: in an open loop
 *     pbre_step_async(a[0], out[0]);  for t = 1..: { pbre_step_async(a[t], out[t & 1]); pbre_step_wait(); consume out[(t - 1) & 1]; }
 * the download of step t - 1 (the PCIe floor: num_envs x (obs_dim + 2) x 4 bytes) overlaps the kernels of step t and the upload of the
 * actions of step t + 1.  Both host buffers must be page-locked (pbre_host_alloc) for the copies to be asynchronous, and stay untouched
 * until the step's pbre_step_wait returns.  Rows are bit-equal to pbre_step's.  Any host-synchronous entry point (pbre_get_state, ...)
 * also completes every step in flight (their pbre_step_wait calls then return at once).  Panda task envs; PBRE_E_UNSUPPORTED elsewhere. */
int pbre_step_async(pbre_ctx* ctx, const float* actions, float* out);
int pbre_step_wait(pbre_ctx* ctx);

/* Same with device-resident buffers (HIP device pointers on ctx's GPU) enqueued on `stream`; asynchronous.  For
 * device-resident policies (replaces the stable-baselines DummyVecEnv hop, R/examples/algos/train/.../train_ddpg_reaching.py:96).
 * stream: a hipStream_t.  The step reads d_actions and writes d_out in stream order on THAT stream, so work that produced the
 * actions / consumes the rows on the same stream needs no further synchronisation.
 *   PBRE_STREAM_LEGACY  HIP's legacy default stream (hipStreamLegacy) -- what `torch.cuda.current_stream().cuda_stream == 0` means;
 *   NULL                the ctx's own NON-BLOCKING stream: it is not ordered against any other stream (not even the legacy
 *                       default stream); the caller orders inputs / outputs with events or pbre_sync().
 * Every host-synchronous entry point (pbre_reset, pbre_get_state, pbre_set_state, pbre_observe, pbre_settle, pbre_set_physics,
 * pbre_step, pbre_sync, ...) first waits for all steps enqueued this way, whatever stream they went to.
 * One-time host synchronisation: the FIRST step enqueued on a stream the ctx has not seen before calibrates which of its internal
 * side streams overlaps with that stream (two ~150 us probe kernels and a host wait on an event; up to four caller streams are
 * remembered, PBRE_SIDE_PROBE=0 disables the probe).  A stream that is being captured into a hipGraph is never probed.
 * Returns PBRE_E_ARG when PBRE_F_AUTO_RESET is set and pbre_set_physics changed the scene since the last full pbre_reset (the settled
 * snapshot the in-kernel restart uses is stale). */
#define PBRE_STREAM_LEGACY ((void*)1)
int pbre_step_device(pbre_ctx* ctx, const float* d_actions, float* d_out, void* stream);
int pbre_sync(pbre_ctx* ctx);

/* raw simulator state, host [num_envs][pbre_state_floats()] float32 (parity tests, checkpoint/restore) */
int pbre_get_state(pbre_ctx* ctx, float* state);
int pbre_set_state(pbre_ctx* ctx, const float* state);
/* `count` consecutive floats of every env's state record starting at float `first`, host [num_envs][count] -- the reference
 * attributes `_env_step_counter`, `terminated`, `_hand_pose`, `_target_pose` (R/envs/panda_envs/panda_push_gym_env.py:45-46,142-143)
 * without downloading the whole batch state */
int pbre_get_state_cols(pbre_ctx* ctx, int32_t first, int32_t count, float* out);

/* Page-locked host memory for the buffers handed to pbre_step / pbre_reset / pbre_observe: DMA straight into the caller's
 * arrays instead of the runtime's staging copies (the rows of a 131072-env step are 18 MB).  NULL on failure. */
void* pbre_host_alloc(size_t bytes);
void pbre_host_free(void* p);
/* recompute the observation of the current state (replaces get_extended_observation, :150-187) */
int pbre_observe(pbre_ctx* ctx, float* obs_out);
/* `n` bare physics steps with hold motors (replaces the settle loops `for _ in range(100): p.stepSimulation`,
 * panda_push_gym_env.py:132-133,139-140); flags: PBRE_F_NO_OBJECT or 0 */
int pbre_settle(pbre_ctx* ctx, int32_t n, int32_t flags);

/* Robot-level engines (robot_level = 1): command the persistent POSITION_CONTROL motors of `n` DoF -- replaces the
 * p.setJointMotorControlArray / setJointMotorControl2 calls of iCubHandsEnv.open_hand / pre_grasp / grasp (R/envs/icub_envs/
 * icub_env_with_hands.py:167-244) and pandaEnv.apply_action_fingers (R/envs/panda_envs/panda_env.py:201-225).  dofs: DoF
 * indices; targets: host [n], the same for every selected env; kp: positionGain; max_force: the `force(s)` entry in newtons
 * (grasp: 10), <= 0 keeps PyBullet's default; max_vel: `maxVelocity` (rad/s or m/s), <= 0 none; env_mask: NULL = all envs,
 * else num_envs bytes. */
int pbre_set_motors(pbre_ctx* ctx, int32_t n, const int32_t* dofs, const float* targets, double kp, double max_force,
                    double max_vel, const uint8_t* env_mask);

/* Robot-level engines: the command half of iCubEnv.apply_action / pandaEnv.apply_action alone (R/envs/icub_envs/icub_env.py:
 * 260-361, R/envs/panda_envs/panda_env.py:227-310) -- IK + the setJointMotorControl calls, no stepSimulation: the motors keep
 * the command until the next one, the caller advances the simulation with pbre_settle (the reference's `p.stepSimulation()`
 * loops, examples/helloworlds/helloworld_icub.py:61-125, helloworld_panda.py:89-140).
 * actions: host [num_envs][act_dim], absolute joint targets (joint control) or hand poses x,y,z[,roll,pitch,yaw] (use_ik).
 * max_vel: the `max_vel` argument (maxVelocity of the commanded motors); <= 0 (the reference's -1): none. */
int pbre_apply_action(pbre_ctx* ctx, const float* actions, double max_vel);

/* Robot-level engines: the motor records, host [num_envs][4][W] float32 = target | positionGain | force scale | max velocity per
 * DoF lane (W = 128 iCub with hands, 32 Panda; force scale = force / PyBullet's default force; max velocity 0 = none).  Together
 * with pbre_get_state / pbre_set_state this is the complete simulator state of such an engine (checkpoint / restore, parity tests). */
int pbre_get_motor_state(pbre_ctx* ctx, float* motors);
int pbre_set_motor_state(pbre_ctx* ctx, const float* motors);

/* replace the physics constants of every env of the batch (replaces p.changeDynamics in change_physics_params,
 * R/envs/panda_envs/panda_push_gym_env.py:362-368: object mass / friction / link damping -- domain randomisation between
 * episodes; batch-uniform).  The object must stay a cube (isotropic inertia) for the lane-per-env kernels. */
int pbre_set_physics(pbre_ctx* ctx, const pbre_physics* phys);
int pbre_get_physics(const pbre_ctx* ctx, pbre_physics* phys);

/* The object as a CONVEX HULL (replaces what p.loadURDF / p.loadSDF do with the mesh objects of WorldEnv, YcbWorldEnv and
 * SqWorldEnv -- duck_vhacd, teddy_vhacd, the YCB and superquadric models, R/envs/world_envs/world_env.py:18-25, 61-84, 179-216: Bullet
 * collides the pieces of such a convex decomposition as btConvexHullShape).  verts: host [n_verts][3] float64, the hull's vertices in the
 * object's frame (origin = centre of mass), 4 <= n_verts <= PBRE_HULL_MAXV; they should be the extreme points of their hull (a point
 * inside it is accepted and is then a contact candidate like any other).  The library derives the hull's faces itself (<= PBRE_HULL_MAXF
 * triangles; coplanar vertices form one polygonal face).  Narrow phase: against the table / ground the PBRE_NC_OT = 4 deepest vertices
 * within the contact margin, in vertex order (the box primitive's rule over its 8 vertices: a box given as its 8 vertices in the box's
 * vertex order v = (x > 0) + 2 (y > 0) + 4 (z > 0) reproduces PBRE_SHAPE_BOX); against a robot link's collision sphere the closest
 * point of the hull's surface (exact: nearest face, edge or vertex), or -- centre inside -- the nearest face plane.
 * On success pbre_physics.obj_shape becomes PBRE_SHAPE_HULL and obj_h the half extents of the hull's bounding box; mass, inertia and
 * friction stay what pbre_physics says (obj_inertia: principal inertias about the object's axes).  A scene change like any other: the
 * settled snapshot is stale until the next full pbre_reset.  Hull objects are stepped by the general 16-/32-/64-lane row kernels of
 * every engine (the lane-per-env fast paths are compiled for the primitives).  pbre_set_physics with another obj_shape drops the hull;
 * with PBRE_SHAPE_HULL it keeps it.  PBRE_E_ARG: bad count, non-finite or degenerate (flat) vertex set, more faces than PBRE_HULL_MAXF. */
enum { PBRE_HULL_MAXV = 32, PBRE_HULL_MAXF = 64 };
int pbre_set_object_hull(pbre_ctx* ctx, const double* verts, int32_t n_verts);
/* pbre_physics.solver_residual_threshold > 0: the number of sweeps every env's solver ran in the most recent simulation step
 * (1..solver_iters; solver_iters when the test never fired), host [num_envs] int32.  PBRE_E_UNSUPPORTED while the threshold is 0. */
int pbre_get_sweeps(pbre_ctx* ctx, int32_t* sweeps);

/* Per-env domain randomisation of the object (replaces the per-env, per-episode p.changeDynamics(obj_id, mass=, lateralFriction=,
 * linearDamping=) of change_physics_params, R/envs/panda_envs/panda_push_gym_env.py:362-364, as the reference's Dyn-Rand training
 * calls it once per env).  Arrays are host [num_envs] float32 or NULL (unchanged); env_mask NULL = all envs.  The values live in the
 * env's state record (X[12] mass, X[13] lateral friction, X[15] 1 + linear damping; 0 = the batch value of pbre_physics), survive
 * resets and travel with pbre_get_state / pbre_set_state; the (cube) object's inertia scales with its mass.  robot_lin_damping: the
 * robot links' linear damping (`robot_damping`, :366-367: p.changeDynamics(robot_id, i, linearDamping=...)), per env as well, in
 * V[15] (1 + damping; 0 = the batch value pbre_physics.lin_damping).  Panda task envs only.
 * Restarts from the settled snapshot (pbre_reset_snapshot, PBRE_F_AUTO_RESET) place every env at env 0's settled robot pose and object
 * height: per-env values do not move where the object rests, and the hold motors bring the arm to the same pose whatever its links'
 * velocity damping, so the approximation (< 5e-5 against an explicit reset with uniform values) carries over to heterogeneous damping
 * only approximately -- an env whose robot damping differs from env 0's by a factor settles a few 1e-4 rad elsewhere within the 201
 * settle steps; callers that randomise robot damping and need the exact settled pose use pbre_reset(mask). */
int pbre_set_physics_per_env(pbre_ctx* ctx, const uint8_t* env_mask, const float* obj_mass, const float* obj_mu,
                             const float* obj_lin_damping, const float* robot_lin_damping);

/* observation limits used by the Gym Box space / scale_gym_data (create_gym_spaces, :83-103) */
int pbre_obs_limits(const pbre_ctx* ctx, float* low, float* high);

/* ms of the last pbre_step phases: [0] upload, [1] kernels, [2] download; [3] mean duration of the dominant kernel
 * (k_fast, or k_step on the general path) over the most recent <= 64 steps of any kind, from HIP events recorded on the
 * stream that kernel runs on (synchronises the device) */
int pbre_timing(const pbre_ctx* ctx, double* ms, int32_t n);
/* kernel facts for the bench/roofline report: [0] VGPRs of the fast kernel, [1] VGPRs of the general kernel,
 * [2] fast path enabled, [3..5] envs stepped in the most recent step by the fast kernel / the general row kernel / the
 * lane-per-env robot-contact kernel, [6] VGPRs of the robot-contact kernel, [7] running sum of the complex envs stepped so far
 * (env-steps taken by the robot-contact / limit-row kernels; wraps at 2^31), [8] steps since the last reset whose fast kernel was the
 * variant limited to 3 waves per SIMD (picked when the complex envs' waves would otherwise displace fast-kernel waves), [9] its VGPRs,
 * [10] steps since the last reset whose simple envs were stepped by the pair kernel (robot wave + object wave per 64 envs: the mapping
 * for batches that leave most SIMDs without a wave), [11] its VGPRs, [12] NaN / Inf guard: env-steps since pbre_create whose state was
 * not finite (NaN or +-Inf in a joint angle / velocity or in the object's pose / twist, on input or after the step).  Such an env-step is
 * returned with reward 0 and done 1; with PBRE_F_AUTO_RESET the env restarts from the settled snapshot in the same step, without it the
 * env keeps its NaN state (and is counted again every step) until the caller resets it.  The reference has no such guard.
 * [13] steps since the last reset that were ONE launch (the complex envs' row blocks and the simple envs' waves in one grid, round 5: no
 * fork / join through a second stream; [10] still says whether the simple envs' waves were the pair mapping), [14] its VGPRs, [15] those
 * of [13] in which the last chunks of a machine-filling batch -- the ones the complex envs' waves keep out of the first round -- were
 * stepped as robot wave + object wave pairs (round 6: "tail pairs"). */
int pbre_kernel_info(const pbre_ctx* ctx, int32_t* info, int32_t n);

/* ---- The sharded batch's per-step gather, owned by the context (no counterpart in the reference: it is one env per physics client
 * and process, R/envs/panda_envs/panda_push_gym_env.py:56-62; BASELINE north_star: "shards the env batch across the 8 GPUs of one node
 * with a single RCCL gather over xGMI per step to return stacked observations / rewards"; SURVEY 8(b): "one RCCL communicator per
 * ctx", 8(e): "gather-to-root emulated with ncclGroupStart + ncclSend / ncclRecv").  One process per GPU; the ctx of rank r was created
 * with num_envs = N / G and env_id_base = r N / G.  RCCL is loaded with dlopen on first use (PBRE_RCCL_LIB, else librccl.so.1 /
 * librccl.so): libpbre.so does not link against it.
 *   pbre_comm_probe       any rank: does the RCCL library load and export what the exchanges need (dlopen + dlsym only; no bootstrap
 *                         thread, no socket) -- what the ranks other than 0 call before they agree to build the communicator;
 *   pbre_comm_unique_id   rank 0: the 128-byte ncclUniqueId every rank's pbre_comm_init needs (hand it over by any out-of-band channel);
 *   pbre_comm_init        collective over the G ranks: this ctx's communicator (ncclCommInitRank on the ctx's device) and its
 *                         communication stream; released by pbre_destroy;
 *   pbre_step_gather_device  = pbre_step_device(d_actions -> d_rows_local) on `stream`, then -- on the communication stream, ordered
 *                         behind the step by an event -- ONE grouped exchange: every rank sends its [num_envs, obs_dim + 2] rows to
 *                         rank 0, which receives them into d_rows_all [G * num_envs, obs_dim + 2] (rank order; its own rows by a
 *                         device copy; d_rows_all is ignored on the other ranks).  Asynchronous.  The caller alternates between TWO
 *                         (d_rows_local, d_rows_all) buffer pairs: the exchange of step k overlaps the kernels of step k + 1, and a
 *                         row buffer is stepped into again only after the exchange that last read it (tracked per d_rows_local
 *                         pointer, up to four: the stream waits for that exchange's event).  A step into such a buffer through plain
 *                         pbre_step_device is NOT protected: call pbre_gather_wait(stream) first;
 *   pbre_gather_wait      makes `stream` (and, host_too != 0, the host) wait for every exchange enqueued so far: then rank 0 may
 *                         read d_rows_all;
 *   pbre_scatter_actions_device  the way back of a CLOSED loop (policy on rank 0: a(t + 1) = pi(obs(t)) needs the gathered rows of step
 *                         t before step t + 1 can start, so that gather cannot be overlapped): rank 0 holds d_actions_all
 *                         [G * num_envs, act_dim] (rank order), every rank receives its [num_envs, act_dim] slice into
 *                         d_actions_local -- ONE grouped exchange (ncclSend x (G - 1) on rank 0, one ncclRecv elsewhere; rank 0's own
 *                         slice by a device copy) on the communication stream, ordered behind what `stream` holds on entry (the
 *                         policy) and ahead of what is enqueued on `stream` afterwards (the step).  d_actions_all is ignored on the
 *                         other ranks;
 *   pbre_comm_info        [0] ranks in the communicator (ncclCommCount), [1] this rank, [2] RCCL version code, [3] gathers enqueued,
 *                         [4] action scatters enqueued;
 *   pbre_comm_last_error  message of the last failed call above (borrowed pointer).
 * stream: as for pbre_step_device -- a hipStream_t; PBRE_STREAM_LEGACY (HIP's legacy default stream, torch's default) is a stream like any
 * other here; NULL = the ctx's own stream, ordered through the host (every call then synchronises; pbre_scatter_actions_device refuses it).
 * An exchange that fails inside its ncclGroup closes the group and marks the communicator unusable (every later call returns the error).
 * PBRE_COMM_SELF_P2P=1 (read at pbre_comm_init): rank 0's own rows travel through ncclSend / ncclRecv too (single-GPU tests execute
 * the RCCL point-to-point path that way). */
int pbre_comm_probe(void);
int pbre_comm_unique_id(void* id128);
int pbre_comm_init(pbre_ctx* ctx, const void* id128, int32_t rank, int32_t world);
int pbre_step_gather_device(pbre_ctx* ctx, const float* d_actions, float* d_rows_local, float* d_rows_all, void* stream);
int pbre_gather_wait(pbre_ctx* ctx, void* stream, int32_t host_too);
int pbre_scatter_actions_device(pbre_ctx* ctx, const float* d_actions_all, float* d_actions_local, void* stream);
int pbre_comm_info(const pbre_ctx* ctx, int32_t* info, int32_t n);
const char* pbre_comm_last_error(const pbre_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
