"""pandaEnv -- robot side of the scene (reference pybullet_robot_envs/envs/panda_envs/panda_env.py).

Keeps the reference constructor signature and the methods the task envs use.  The URDF is parsed once by
the engine's own model compiler (model/urdf.py -> robot_data/franka_panda/panda_model.json) and handed
to libpbre as a flat RobotTable; all per-step work (motors, dynamics, observation) runs on the GPU."""
import math as m

import numpy as np

from pybullet_robot_envs import _client
from pybullet_robot_envs._gym import seeding
from pybullet_robot_envs.model.table import panda_table


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

        if use_IK and not control_orientation:
            raise NotImplementedError("use_IK=1 with control_orientation=0 is not implemented (the task envs use the default 1)")
        if control_eu_or_quat != 0:
            raise NotImplementedError("control_eu_or_quat=1 (quaternion observations) is not implemented")
        if not includeVelObs:
            raise NotImplementedError("includeVelObs=False is not implemented (the reference task envs never forward it)")

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
        return 9 + len(self._joint_name_to_ids)

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
        lim.extend(self._eu_lim)
        lim.extend([[-1, 1], [-1, 1], [-1, 1]])
        lim.extend([[self.ll[i], self.ul[i]] for i in range(len(self._joint_name_to_ids))])
        return lim

    def get_observation(self):
        """EE pose (3 + 3 Euler), standardised EE linear velocity (3) and joint positions (9) with their limits
        (panda_env.py:141-193).  List of 18 for a single env, [N, 18] array for a batch."""
        eng = self._client.require_engine()
        obs = eng.observe()[:, :self.get_observation_dim()].astype(np.float64)
        if obs.shape[0] == 1:
            return list(obs[0]), self.get_observation_limits()
        return obs, self.get_observation_limits()

    def apply_action(self, action, max_vel=-1):
        raise NotImplementedError("motor targets are applied inside the fused GPU step; use the task env's step()")

    def pre_grasp(self):
        raise NotImplementedError("finger commands are not implemented by the batched engine")

    def grasp(self, obj_id=None):
        raise NotImplementedError("finger commands are not implemented by the batched engine")

    def check_collision(self, obj_id):
        raise NotImplementedError("contact queries are not exposed by the batched engine")

    def check_contact_fingertips(self, obj_id):
        raise NotImplementedError("contact queries are not exposed by the batched engine")

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def debug_gui(self):
        pass
