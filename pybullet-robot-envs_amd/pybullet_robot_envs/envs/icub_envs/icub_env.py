"""iCubEnv -- robot side of the scene (reference pybullet_robot_envs/envs/icub_envs/icub_env.py).

Keeps the reference constructor signature and the attributes / methods the task envs use.  The SDF is parsed once by the
engine's own model compiler (model/sdf.py -> robot_data/iCub/icub_model.json) and handed to libpbre as a flat RobotTable;
the floating base pinned by `p.createConstraint(JOINT_FIXED)` (icub_env.py:97-103) is a fixed base at the constraint's
rest pose (model/table.py pin_base).  All per-step work (IK, motors, dynamics, observation) runs on the GPU.

Two ways of use, as in the reference:
  * inside a task env (iCubReachGymEnv, ...): the task env owns the batched engine and fuses the robot's motor commands into its
    step(); the robot object provides model bookkeeping, limits and `get_observation`.
  * alone: `iCubEnv(cid, use_IK=1)` on a client from `pybullet_robot_envs.connect(n)`.  The first command or query builds the
    robot-level engine (pbre_config.robot_level, csrc/pbre_icub_arm.hip: persistent POSITION_CONTROL motors with velocity bounds;
    scene of the reach task -- table and object) and `apply_action(action, max_vel)` commands the whole batch;
    `step_simulation(n)` stands for a script's `for _ in range(n): p.stepSimulation()` loops.

Hand-pose commands may be 3 values (position), 6 (position + Euler angles, clipped to the arm's Euler limits as in the reference)
or 7 (position + quaternion, used as given, as in the reference); `max_vel` is the motors' `maxVelocity`."""
import math as m

import numpy as np

from pybullet_robot_envs import _capi, _client
from pybullet_robot_envs._gym import seeding
from pybullet_robot_envs.model.table import icub_table, ICUB_HOME


class iCubEnv:

    initial_positions = dict(ICUB_HOME)

    joint_groups = {'l_leg': ['l_knee', 'l_ankle_pitch', 'l_hip_pitch'],
                    'r_leg': ['r_knee', 'r_ankle_pitch', 'r_hip_pitch'],
                    'head': ['neck_pitch', 'neck_roll', 'neck_yaw'],
                    'torso': ['torso_pitch', 'torso_roll', 'torso_yaw'],
                    'l_arm': ['l_shoulder_pitch', 'l_shoulder_roll', 'l_shoulder_yaw',
                              'l_elbow', 'l_wrist_pitch', 'l_wrist_prosup', 'l_wrist_yaw'],
                    'r_arm': ['r_shoulder_pitch', 'r_shoulder_roll', 'r_shoulder_yaw',
                              'r_elbow', 'r_wrist_pitch', 'r_wrist_prosup', 'r_wrist_yaw'],
                    }

    def __init__(self, physicsClientId, use_IK=0, control_arm='l', control_orientation=1, control_eu_or_quat=0, floating_base=False):
        # floating_base (new, trailing; default: the base rigidly pinned at the constraint's rest pose): the reference's soft-pinned floating
        # base as a dynamic body -- model/table.py: float_base (six virtual joints held by the constraint's equivalent motors, legs lumped)
        self._floating_base = bool(floating_base)

        self._physics_client_id = physicsClientId
        self._client = _client.get(physicsClientId)
        self._client.robot = self
        self._use_IK = use_IK
        self._control_orientation = control_orientation
        self._control_eu_or_quat = control_eu_or_quat
        self._control_arm = control_arm if control_arm == 'r' or control_arm == 'l' else 'l'  # left arm by default

        self.end_eff_idx = []

        self._workspace_lim = [[0.1, 0.45], [-0.3, 0.3], [0.5, 1.0]]
        self._eu_lim = [[-m.pi/2, m.pi/2], [-m.pi/2, m.pi/2], [-m.pi/2, m.pi/2]]

        # set initial hand pose (icub_env.py:66-74)
        if self._control_arm == 'l':
            self._home_hand_pose = [0.3, 0.26, 0.8, 0, 0, 0]  # x, y, z, roll, pitch, yaw
            self._eu_lim = [[-m.pi / 2, m.pi / 2], [-m.pi / 2, m.pi / 2], [-m.pi / 2, m.pi / 2]]
        else:
            self._home_hand_pose = [0.3, -0.26, 0.8, 0, 0, m.pi]
            self._eu_lim = [[-m.pi / 2, m.pi / 2], [-m.pi / 2, m.pi / 2], [m.pi / 2, 3 / 2 * m.pi]]

        self._joints_to_control = []
        self._joints_to_block = []
        self._joint_name_to_ids = {}

        self.robot_id = 0

        self.ll, self.ul, self.jr, self.rs, self.jd = None, None, None, None, None

        self.seed()
        self.reset()

    def reset(self):
        # Load robot model: parsed parameters -> RobotTable (replaces p.loadSDF + p.createConstraint, icub_env.py:91-103)
        # The reference's joint names / indices / limits cover the whole SDF model (38 links, 32 DoF); the engine simulates
        # the model without the legs: limbs rooted at the fixed base are independent dynamical systems, the legs carry no
        # collision geometry and no env observes them, so they cannot change any output (model/table.py prune_base_branches;
        # tests/test_golden_icub.py::test_pruned_legs_are_exact).
        self.robot_table, self._sim_model, self._info = icub_table(self._control_arm, floating_base=self._floating_base)
        _, self._model, self._full_info = icub_table(self._control_arm, full=True)
        self._joint_name_to_ids = {}
        for i, link in enumerate(self._model["links"]):
            if link["jtype"] != 0:
                assert link["joint_name"] in self.initial_positions.keys()
                self._joint_name_to_ids[link["joint_name"]] = i

        # Controlled joints (reference icub_env.py:123-143): the torso and the chosen arm, in link-index order; every other joint is
        # "blocked" (held at its rest pose by the IK branch); the end effector is the chosen arm's wrist-yaw link.
        if not self._joints_to_control:
            side = 'l' if self._control_arm == 'l' else 'r'
            driven = set(self.joint_groups['torso']) | set(self.joint_groups[side + '_arm'])
            ids = self._joint_name_to_ids
            self._joints_to_control = [i for name, i in ids.items() if name in driven]
            self._joints_to_block = [i for name, i in ids.items() if name not in driven]
            self.end_eff_idx = ids[side + '_wrist_yaw']
        assert self.end_eff_idx == self._full_info["ee_link"]

        self.ll, self.ul, self.jr, self.rs, self.jd = self.get_joint_ranges()
        # `if self._use_IK: self.apply_action(self._home_hand_pose)` + `p.stepSimulation()` (icub_env.py:147-151) happen inside
        # the engine's reset (pbre_reset)

    def controlled_dofs(self):
        """DoF indices (engine numbering, i.e. in the simulated model) of _joints_to_control, same order."""
        sim = self._info["dof_names"]
        return [sim.index(self._model["links"][i]["joint_name"]) for i in self._joints_to_control]

    def sim_home(self):
        """initial_positions per DoF of the simulated model"""
        return [self.initial_positions.get(n, 0.0) for n in self._info["dof_names"]]

    def get_joint_ranges(self):
        """lower / upper limits, ranges, rest poses (= initial positions) and IK joint damping (0.1 controlled, 100 blocked) of
        every joint, in link-index order (reference icub_env.py:157-174, from the parsed model instead of p.getJointInfo)"""
        links = [(n, self._model["links"][i], i in self._joints_to_control) for n, i in self._joint_name_to_ids.items()]
        lo = [l["lower"] for _, l, _ in links]
        hi = [l["upper"] for _, l, _ in links]
        return (lo, hi, [u - d for d, u in zip(lo, hi)], [self.initial_positions[n] for n, _, _ in links],
                [0.1 if c else 100. for _, _, c in links])

    def get_workspace(self):
        return [i[:] for i in self._workspace_lim]

    def set_workspace(self, ws):
        self._workspace_lim = [i[:] for i in ws]

    def get_rotation_lim(self):
        return [i[:] for i in self._eu_lim]

    def set_rotation_lim(self, eu):
        self._eu_lim = [i[:] for i in eu]

    def get_action_dim(self):
        if not self._use_IK:
            return len(self._joints_to_control)
        if self._control_orientation and self._control_eu_or_quat == 0:
            return 6  # position x,y,z + roll/pitch/yaw of hand frame
        elif self._control_orientation and self._control_eu_or_quat == 1:
            return 7  # position x,y,z + quat of hand frame
        return 3  # position x,y,z

    def get_observation_dim(self):
        return (9 if self._control_eu_or_quat == 0 else 10) + len(self._joints_to_control)

    def get_observation_limits(self):
        lim = []
        lim.extend(list(self._workspace_lim))
        if self._control_eu_or_quat == 0:
            lim.extend(self._eu_lim)
        else:
            lim.extend([[-1, 1], [-1, 1], [-1, 1], [-1, 1]])
        lim.extend([[-1, 1], [-1, 1], [-1, 1]])
        lim.extend([[self.ll[i], self.ul[i]] for i, idx in enumerate(self._joint_name_to_ids.values())
                    if idx in self._joints_to_control])
        return lim

    def get_observation(self):
        """Hand COM pose (3 + 3 Euler), its linear velocity (3) and the controlled joint positions (10) with their limits
        (icub_env.py:202-249).  List of 19 for a single env, [N, 19] array for a batch."""
        eng = self._engine_or_build()
        obs = eng.observe()[:, :9 + len(self._joints_to_control)].astype(np.float64)
        if self._control_eu_or_quat != 0:      # hand orientation as a quaternion (icub_env.py:219-224): converted from the engine's Euler angles
            from pybullet_robot_envs.envs.panda_envs.panda_env import pandaEnv
            obs = np.concatenate([obs[:, :3], pandaEnv._quat_from_euler(obs[:, 3:6]), obs[:, 6:]], axis=1)
        if obs.shape[0] == 1:
            return list(obs[0]), self.get_observation_limits()
        return obs, self.get_observation_limits()

    def _com_to_link_hand_frame(self):
        if self._control_arm == 'r':
            com_T_link_hand = ((0.064668, -0.0056, -0.022681), (0., 0., 0., 1.))
        else:
            com_T_link_hand = ((-0.064768, -0.00563, -0.02266), (0., 0., 0., 1.))
        return com_T_link_hand

    # ------------------------------------------------------------------ engine of the stand-alone (robot-level) use
    def _engine_or_build(self):
        """The task env's engine when there is one, else the robot-level engine (built and reset on first use: replaces the
        reference reset's loadSDF + motors at the initial positions + apply_action(home hand pose) + stepSimulation,
        icub_env.py:91-151)."""
        c = self._client
        if c.engine is None:
            dofs = self.controlled_dofs()
            home = self.sim_home()
            c.engine = _capi.make_engine(self.robot_table, devices=c.devices, task=_capi.TASK_REACH, num_envs=c.num_envs, lib=c.lib,
                                         robot=_capi.ROBOT_ICUB, robot_level=1, ik_absolute=1, max_steps=1 << 30, target_dist_min=-1.0,
                                         device_id=c.device_id, env_id_base=c.env_id_base, seed=c.seed,
                                         use_ik=1 if self._use_IK else 0, control_orientation=1 if self._control_orientation else 0,
                                         num_controlled_joints=len(dofs), num_joints_ctrl=len(dofs), act_dof=dofs + [-1] * (16 - len(dofs)),
                                         home=home + [0.0] * (40 - len(home)), ik_pos_scale=1.0, ik_rot_scale=1.0,
                                         home_hand_pose=[float(x) for x in self._home_hand_pose],
                                         eu_lim=[-1e9, 1e9] * 3,        # Euler limits are applied in apply_action: a quaternion command bypasses them
                                         ik_link_offset=list(self._com_to_link_hand_frame()[0]),
                                         robot_ws=[x for lim in self._workspace_lim for x in lim])
            c.engine.reset()
            self._own_engine = True
        return c.engine

    def _robot_level(self):
        eng = self._engine_or_build()
        if not getattr(self, "_own_engine", False):
            raise RuntimeError("inside a task env the motors are commanded by the env's fused step(); robot-level commands belong "
                               "to iCubEnv used alone (see the module docstring)")
        return eng

    @property
    def num_envs(self):
        return self._client.num_envs

    def _batch(self, action):
        a = np.asarray(action, dtype=np.float32)
        if a.ndim == 1:
            a = np.tile(a, (self.num_envs, 1))
        return a

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

    def _hand_pose_command(self, a, eng):
        """3 / 6 / 7 command values -> what the engine takes (x, y, z[, roll, pitch, yaw]); icub_env.py:262-300."""
        if not (a.shape[1] == 3 or a.shape[1] == 6 or a.shape[1] == 7):
            raise AssertionError('number of action commands must be \n- 3: (dx,dy,dz)'
                                 '\n- 6: (dx,dy,dz,droll,dpitch,dyaw)'
                                 '\n- 7: (dx,dy,dz,qx,qy,qz,w)'
                                 '\ninstead it is: ', a.shape[1])
        if eng.act_dim == 3:              # orientation not under control: the home orientation is kept (:281-283)
            return np.ascontiguousarray(a[:, :3])
        if a.shape[1] == 6:               # Euler angles, each `min(hi, max(lo, x))` (:289-291)
            eu = a[:, 3:6].astype(np.float64)
            for k in range(3):
                eu[:, k] = np.minimum(self._eu_lim[k][1], np.maximum(self._eu_lim[k][0], eu[:, k]))
        elif a.shape[1] == 7:             # quaternion, used as given (:296-297)
            eu = self._euler_from_quat(a[:, 3:7])
        else:                             # `else: use current orientation` (:299-300)
            eu = self._current_hand_euler(eng, a.shape[0])
        return np.concatenate([a[:, :3], eu.astype(np.float32)], axis=1)

    def _current_hand_euler(self, eng, n):
        return eng.observe()[:, 3:6].astype(np.float64)

    def apply_action(self, action, max_vel=-1):
        """Command the motors (icub_env.py:259-360); the simulation does not advance.  Joint control: one absolute target per
        controlled joint (torso + the chosen arm), clipped to the joint limits, gain 0.5.  IK: the hand pose (x, y, z[, roll, pitch,
        yaw | quaternion]) clipped to the workspace, solved over every joint with the other joints held at their rest poses, gain
        0.2.  `max_vel` is the motors' maxVelocity.  A 1-D action is sent to every env, a [N, k] array per env."""
        eng = self._robot_level()
        a = self._batch(action)
        if self._use_IK:
            a = self._hand_pose_command(a, eng)
        elif a.shape[1] != len(self._joints_to_control):
            raise AssertionError('number of motor commands differs from number of motor to control',
                                 a.shape[1], len(self._joints_to_control))
        eng.apply_action(a, max_vel=float(max_vel))

    def step_simulation(self, n=1):
        """`for _ in range(n): p.stepSimulation()` of a script that drives the robot."""
        self._robot_level().settle(int(n))

    def get_object_pose(self):
        """[N, 7] position + quaternion of the scene's object."""
        eng = self._engine_or_build()
        return eng.get_state_cols(eng.obj_off, 7).astype(np.float64)

    def get_joint_positions(self):
        """[N, ndof] joint positions of the simulated model, `self._info['dof_names']` order."""
        eng = self._engine_or_build()
        return eng.get_state_cols(0, eng.ndof).astype(np.float64)

    def delete_simulated_robot(self):
        pass

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def debug_gui(self):
        pass
