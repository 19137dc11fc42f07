"""pandaEnv -- robot side of the scene (reference pybullet_robot_envs/envs/panda_envs/panda_env.py).

Keeps the reference constructor signature, attributes and methods.  The URDF is parsed once by the engine's own model
compiler (model/urdf.py -> robot_data/franka_panda/panda_model.json) and handed to libpbre as a flat RobotTable; all per-step
work (motors, dynamics, observation) runs on the GPU.

Two ways of use, as in the reference:
  * inside a task env (pandaPushGymEnv, ...): the task env owns the batched engine and fuses the robot's motor commands into its
    step(); the robot object provides model bookkeeping, limits and `get_observation`.
  * alone (examples/helloworlds/helloworld_panda.py): `pandaEnv(cid, use_IK=1)` on a client from `pybullet_robot_envs.connect(n)`.
    The first command or query builds the robot-level engine (pbre_config.robot_level: persistent POSITION_CONTROL motors with
    force / velocity bounds, fingertip contact statistics; scene of the demo -- table at x = 1, a lego-sized box dropped at
    (0.5, 0, 0.8)) and `apply_action` / `pre_grasp` / `grasp` / `check_collision` / `check_contact_fingertips` work on the whole
    batch; `step_simulation(n)` stands for the demo's `for _ in range(n): p.stepSimulation()` loops.

Hand-pose commands may be 3 values (position), 6 (position + Euler angles) or 7 (position + quaternion), `max_vel` is the
motors' `maxVelocity`; `control_eu_or_quat=1` returns the end-effector orientation as a quaternion, `includeVelObs=False` drops
the velocity entries (both assembled on the host from the engine's observation)."""
import math as m

import numpy as np

from pybullet_robot_envs import _capi, _client
from pybullet_robot_envs._gym import seeding
from pybullet_robot_envs.model.table import panda_table, panda_arm_table


class pandaEnv:

    initial_positions = {
        'panda_joint1': 0.0, 'panda_joint2': -0.54, 'panda_joint3': 0.0,
        'panda_joint4': -2.6, 'panda_joint5': -0.30, 'panda_joint6': 2.0,
        'panda_joint7': 1.0, 'panda_finger_joint1': 0.02, 'panda_finger_joint2': 0.02,
    }

    def __init__(self, physicsClientId, use_IK=0, base_position=(0.0, 0, 0.625), control_orientation=1, control_eu_or_quat=0,
                 joint_action_space=9, includeVelObs=True):

        self._physics_client_id = physicsClientId
        self._client = _client.get(physicsClientId)
        self._client.robot = self
        self._use_IK = use_IK
        self._control_orientation = control_orientation
        self._base_position = base_position

        self.joint_action_space = joint_action_space
        self._include_vel_obs = includeVelObs
        self._control_eu_or_quat = control_eu_or_quat

        self._workspace_lim = [[0.3, 0.65], [-0.3, 0.3], [0.65, 1.5]]
        self._eu_lim = [[-m.pi, m.pi], [-m.pi, m.pi], [-m.pi, m.pi]]

        self.end_eff_idx = 11  # 8

        self._home_hand_pose = []

        self._num_dof = 7
        self._joint_name_to_ids = {}
        self.robot_id = 0

        self._own_engine = False       # True once the robot-level engine of the stand-alone use has been built
        self.seed()
        self.reset()

    def reset(self):
        # Load robot model: parsed parameters -> RobotTable (replaces p.loadURDF, panda_env.py:53-56)
        self.robot_table, self._model = panda_table(self._base_position)
        self._joint_name_to_ids = {}
        for i, link in enumerate(self._model["links"]):
            if link["jtype"] != 0:
                assert link["joint_name"] in self.initial_positions.keys()
                self._joint_name_to_ids[link["joint_name"]] = i
        self.ll, self.ul, self.jr, self.rs = self.get_joint_ranges()

        if self._use_IK:
            # panda_env.py:83-91; the IK solve + first stepSimulation happen inside the engine's reset (pbre_reset)
            self._home_hand_pose = [0.2, 0.0, 0.8,
                                    min(m.pi, max(-m.pi, m.pi)),
                                    min(m.pi, max(-m.pi, 0)),
                                    min(m.pi, max(-m.pi, 0))]

    def get_joint_ranges(self):
        lower_limits, upper_limits, joint_ranges, rest_poses = [], [], [], []
        for joint_name in self._joint_name_to_ids.keys():
            link = self._model["links"][self._joint_name_to_ids[joint_name]]
            ll, ul = link["lower"], link["upper"]
            lower_limits.append(ll)
            upper_limits.append(ul)
            joint_ranges.append(ul - ll)
            rest_poses.append(self.initial_positions[joint_name])
        return lower_limits, upper_limits, joint_ranges, rest_poses

    def get_action_dim(self):
        if not self._use_IK:
            return self.joint_action_space
        if self._control_orientation and self._control_eu_or_quat == 0:
            return 6
        elif self._control_orientation and self._control_eu_or_quat == 1:
            return 7
        return 3

    def get_observation_dim(self):
        return 3 + (3 if self._control_eu_or_quat == 0 else 4) + (3 if self._include_vel_obs else 0) + len(self._joint_name_to_ids)

    def get_workspace(self):
        return [i[:] for i in self._workspace_lim]

    def set_workspace(self, ws):
        self._workspace_lim = [i[:] for i in ws]

    def get_rotation_lim(self):
        return [i[:] for i in self._eu_lim]

    def set_rotation_lim(self, eu):
        self._eu_lim = [i[:] for i in eu]

    def get_observation_limits(self):
        lim = []
        lim.extend(list(self._workspace_lim))
        if self._control_eu_or_quat == 0:
            lim.extend(self._eu_lim)
        else:
            lim.extend([[-1, 1], [-1, 1], [-1, 1], [-1, 1]])
        if self._include_vel_obs:
            lim.extend([[-1, 1], [-1, 1], [-1, 1]])
        lim.extend([[self.ll[i], self.ul[i]] for i in range(len(self._joint_name_to_ids))])
        return lim

    # ------------------------------------------------------------------ engine of the stand-alone (robot-level) use
    def _engine_or_build(self):
        """The task env's engine when there is one, else the robot-level engine (built and reset on first use: replaces the
        reference constructor's loadURDF + motors + apply_action(home pose) + stepSimulation, panda_env.py:51-91)."""
        c = self._client
        if c.engine is None:
            tbl, _ = panda_arm_table(self._base_position)
            ws = self._workspace_lim
            c.engine = _capi.make_engine(tbl, devices=c.devices, task=_capi.TASK_REACH, num_envs=c.num_envs, lib=c.lib, robot=_capi.ROBOT_PANDA_ARM,
                                    device_id=c.device_id, env_id_base=c.env_id_base, seed=c.seed,
                                    use_ik=1 if self._use_IK else 0, control_orientation=1 if self._control_orientation else 0,
                                    num_controlled_joints=int(self.joint_action_space), num_joints_ctrl=int(self.joint_action_space),
                                    eu_lim=[-1e9, 1e9] * 3,        # Euler limits are applied in apply_action: a quaternion command bypasses them
                                    robot_ws=[x for lim in ws for x in lim])
            c.engine.reset()
            self._own_engine = True
        return c.engine

    def _robot_level(self):
        eng = self._engine_or_build()
        if not self._own_engine:
            raise RuntimeError("inside a task env the motors are commanded by the env's fused step(); robot-level commands and "
                               "contact queries belong to pandaEnv used alone (see the module docstring)")
        return eng

    @property
    def num_envs(self):
        return self._client.num_envs

    @staticmethod
    def _quat_from_euler(e):
        """pybullet.getQuaternionFromEuler for [N, 3] -> [N, 4] (x, y, z, w)"""
        cr, sr, cp, sp, cy, sy = (f(e[:, k] * 0.5) for k in range(3) for f in (np.cos, np.sin))
        return np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], axis=1)

    @staticmethod
    def _euler_from_quat(q):
        """pybullet.getEulerFromQuaternion for [N, 4] (x, y, z, w)."""
        x, y, z, w = (q[:, k].astype(np.float64) for k in range(4))
        sarg = -2.0 * (x * z - w * y)
        roll = np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z)
        pitch = np.arcsin(np.clip(sarg, -1.0, 1.0))
        yaw = np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)
        lo, hi = sarg <= -0.99999, sarg >= 0.99999
        roll = np.where(lo | hi, 0.0, roll)
        pitch = np.where(lo, -0.5 * m.pi, np.where(hi, 0.5 * m.pi, pitch))
        yaw = np.where(lo, 2 * np.arctan2(x, -y), np.where(hi, 2 * np.arctan2(-x, y), yaw))
        return np.stack([roll, pitch, yaw], axis=1)

    def get_observation(self):
        """EE position (3), orientation (3 Euler angles, or the quaternion with control_eu_or_quat=1), standardised EE linear
        velocity (3, unless includeVelObs=False) and joint positions (9) with their limits (panda_env.py:141-193).  A list for a
        single env, an [N, dim] array for a batch."""
        eng = self._engine_or_build()
        raw = eng.observe()[:, :18].astype(np.float64)
        parts = [raw[:, 0:3]]
        parts.append(raw[:, 3:6] if self._control_eu_or_quat == 0 else self._quat_from_euler(raw[:, 3:6]))
        if self._include_vel_obs:
            parts.append(raw[:, 6:9])
        parts.append(raw[:, 9:18])
        obs = np.concatenate(parts, axis=1)
        if obs.shape[0] == 1:
            return list(obs[0]), self.get_observation_limits()
        return obs, self.get_observation_limits()

    # ------------------------------------------------------------------ robot-level commands (stand-alone use)
    def _batch(self, action):
        a = np.asarray(action, dtype=np.float64)
        if a.ndim == 1:
            a = np.tile(a, (self.num_envs, 1))
        return a

    def pre_grasp(self):
        self.apply_action_fingers([0.04, 0.04])

    def grasp(self, obj_id=None):
        self.apply_action_fingers([0.0, 0.0], obj_id)

    def apply_action_fingers(self, action, obj_id=None):
        """Finger joints in position control with force 10 and maxVelocity 1 (panda_env.py:201-225; PyBullet's default positionGain
        0.1 [EXT-UNVERIFIED]).  With `obj_id` a finger whose contact force on the object has reached 20 N keeps its current
        position instead of closing further -- decided per env."""
        assert len(action) == 2, ('finger joints are 2! The number of actions you passed is ', len(action))
        eng = self._robot_level()
        if obj_id is None:
            eng.set_motors([7, 8], [float(action[0]), float(action[1])], 0.1, 10.0, max_vel=1.0)
            return
        _, forces = self.check_contact_fingertips(obj_id)
        forces = np.atleast_2d(np.asarray(forces, dtype=np.float64))
        q = eng.get_state_cols(7, 2)
        mot = eng.get_motor_state()
        for k in range(2):
            mot[:, 0, 7 + k] = np.where(forces[:, k] >= 20.0, q[:, k], float(action[k]))
            mot[:, 1, 7 + k] = 0.1
            mot[:, 2, 7 + k] = 10.0 * eng.cfg.phys.dt / eng.cfg.phys.max_motor_impulse
            mot[:, 3, 7 + k] = 1.0
        eng.set_motor_state(mot)

    def apply_action(self, action, max_vel=-1):
        """Command the motors (panda_env.py:227-310); the simulation does not advance.  IK: the hand pose -- 3 (position), 6
        (position + roll, pitch, yaw, each clipped to +-pi) or 7 (position + quaternion, used as given) values; z is clipped to
        the workspace; with max_vel != -1 only the 7 arm joints are commanded, with `maxVelocity=max_vel`.  Joint control: one
        absolute target per joint of `joint_action_space`, clipped to the joint limits.  A 1-D action goes to every env."""
        eng = self._robot_level()
        a = self._batch(action)
        if self._use_IK:
            if not (a.shape[1] == 3 or a.shape[1] == 6 or a.shape[1] == 7):
                raise AssertionError('number of action commands must be \n- 3: (dx,dy,dz)'
                                     '\n- 6: (dx,dy,dz,droll,dpitch,dyaw)'
                                     '\n- 7: (dx,dy,dz,qx,qy,qz,w)'
                                     '\ninstead it is: ', a.shape[1])
            if eng.act_dim == 3:                 # orientation not under control: the home orientation is kept (:247-249)
                cmd = a[:, :3]
            else:
                if a.shape[1] == 6:
                    eu = np.minimum(m.pi, np.maximum(-m.pi, a[:, 3:6]))
                elif a.shape[1] == 7:
                    eu = self._euler_from_quat(a[:, 3:7])
                else:                            # `else: use current orientation` (:263-265)
                    eu = eng.observe()[:, 3:6].astype(np.float64)
                cmd = np.concatenate([a[:, :3], eu], axis=1)
            eng.apply_action(cmd.astype(np.float32), max_vel=float(max_vel))
        else:
            assert a.shape[1] == self.joint_action_space, ('number of motor commands differs from number of motor to control', a.shape[1])
            eng.apply_action(a.astype(np.float32), max_vel=float(max_vel))

    def step_simulation(self, n=1):
        """`for _ in range(n): p.stepSimulation()` of the demo script."""
        self._robot_level().settle(int(n))

    def check_contact_fingertips(self, obj_id=None):
        """Fingers (0..2) touching the object and the mean normal force on each (panda_env.py:320-361).  Stand-in geometry: a
        finger is its two collision spheres, so the reference's filter "contact on the internal part of the finger" reduces to
        "contact on the finger"; the reference averages over `[0] + forces` and returns finger 1's mean for both fingers
        (:359: `p1_f_mean = np.mean(p0_f)`) -- here each finger reports its own mean normal force.
        ints / tuples for one env, arrays [N] / [N, 2] for a batch."""
        t = self._robot_level().observe()[:, -7:].astype(np.float64)
        n, f = t[:, 5].astype(int), t[:, :2]
        if self.num_envs == 1:
            return int(n[0]), (float(f[0, 0]), float(f[0, 1]))
        return n, f

    def check_collision(self, obj_id=None):
        # any contact with the object that is not a fingertip contact (:312-318)
        t = self._robot_level().observe()[:, -7:].astype(np.float64)
        c = (t[:, 6] - t[:, 5]) > 0
        return bool(c[0]) if self.num_envs == 1 else c

    def get_object_pose(self):
        """[N, 7] position + quaternion of the object of the demo scene (the demo reads it back through PyBullet)."""
        eng = self._engine_or_build()
        return eng.get_state_cols(eng.obj_off, 7).astype(np.float64)

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def debug_gui(self):
        pass
