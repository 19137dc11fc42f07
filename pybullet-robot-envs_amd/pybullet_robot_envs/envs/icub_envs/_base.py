"""Shared implementation of the three iCub task envs (reach / push / push-goal): the Panda task base with the iCub's
constructor arguments, engine configuration, observation limits and state-record layout (80 floats per env: 20 simulated DoF, one env per half-wave)."""
import math as m

import numpy as np

from pybullet_robot_envs import _capi, _client
from pybullet_robot_envs.envs.panda_envs._base import PandaTaskBase
from pybullet_robot_envs.envs.icub_envs.icub_env import iCubEnv
from pybullet_robot_envs.envs.world_envs.world_env import WorldEnv


class ICubTaskBase(PandaTaskBase):

    def _setup_icub(self, action_repeat, use_IK, control_arm, control_orientation, obj_name, obj_pose_rnd_std, tg_pose_rnd_std,
                    renders, max_steps, reward_type, num_envs, device_id, env_id_base, seed, _lib, auto_reset=False, floating_base=False):
        self._time_step = 1. / 240.
        self._control_arm = control_arm
        self._use_IK = use_IK
        self._control_orientation = control_orientation
        self._action_repeat = action_repeat
        self._observation = []
        self._renders = renders        # accepted for API parity; there is no GUI
        self._max_steps = max_steps
        self._last_frame_time = 0
        self._target_dist_min = 0.03
        self._tg_pose_rnd_std = tg_pose_rnd_std
        self._obj_pose_rnd_std = obj_pose_rnd_std
        self._reward_type = reward_type
        self.num_envs = int(num_envs)
        self._auto_reset = bool(auto_reset)

        self._physics_client_id = _client.connect(num_envs, device_id, env_id_base, seed, _lib)
        self._client = _client.get(self._physics_client_id)

        # Load robot (icub_reach_gym_env.py:68-70)
        self._robot = iCubEnv(self._physics_client_id, use_IK=self._use_IK, control_arm=self._control_arm,
                              control_orientation=self._control_orientation, floating_base=floating_base)

        # Load world environment (:73-75)
        self._world = WorldEnv(self._physics_client_id, obj_name=obj_name, obj_pose_rnd_std=obj_pose_rnd_std,
                               workspace_lim=self._robot.get_workspace())

        # limit iCub workspace to table plane (:78-80)
        workspace = self._robot.get_workspace()
        workspace[2][0] = self._world.get_table_height()
        self._robot.set_workspace(workspace)

        self._build_engine()

        # Define spaces
        self.observation_space, self.action_space = self.create_gym_spaces()
        self.seed()

    def _build_engine(self):
        c = self._client
        if c.engine is not None:
            c.engine.close()
        r = self._robot
        r._own_engine = False            # the engine of this client is the task env's from now on (iCubEnv._robot_level)
        dofs = r.controlled_dofs()
        home = r.sim_home()
        ori = 1 if self._control_orientation else 0
        overrides = dict(device_id=c.device_id, env_id_base=c.env_id_base, seed=c.seed, action_repeat=int(self._action_repeat), max_steps=int(self._max_steps),
                         obj_pose_rnd_std=float(self._obj_pose_rnd_std), tg_pose_rnd_std=float(self._tg_pose_rnd_std),
                         target_dist_min=float(self._target_dist_min), h_table=float(self._world.get_table_height()),
                         flags=_capi.F_AUTO_RESET if self._auto_reset else 0,
                         use_ik=1 if self._use_IK else 0, control_orientation=ori, reward_type=int(self._reward_type),
                         num_controlled_joints=len(dofs), num_joints_ctrl=len(dofs), act_dof=dofs + [-1] * (16 - len(dofs)),
                         home=home + [0.0] * (40 - len(home)),
                         # icub_reach_gym_env.py:206-212: 0.005 (position only) or 0.01 / 0.02 (position / rotation)
                         ik_pos_scale=0.01 if ori else 0.005, ik_rot_scale=0.02,
                         home_hand_pose=[float(x) for x in r._home_hand_pose],
                         eu_lim=[x for lim in r.get_rotation_lim() for x in lim],
                         ik_link_offset=list(r._com_to_link_hand_frame()[0]),
                         ws_lim=[x for lim in self._world.get_workspace() for x in lim],
                         robot_ws=[x for lim in r.get_workspace() for x in lim])
        c.engine = _capi.make_engine(r.robot_table, devices=c.devices, task=self._TASK, num_envs=c.num_envs, lib=c.lib, robot=_capi.ROBOT_ICUB,
                                      phys=self._world.object_physics(), **overrides)
        assert c.engine.act_dim == r.get_action_dim() and c.engine.state_floats == (144 if r._floating_base else 80)
        self._engine = c.engine

    def _exact_limits(self, lim32):
        # the reference builds the limits as Python floats (icub_reach_gym_env.py:150-180, icub_push_gym_env.py:165-203)
        lim = []
        lim.extend(self._robot.get_observation_limits())
        wl = self._world.get_workspace()
        lim.extend(wl)
        lim.extend([[-m.pi, m.pi]] * 3)
        lim.extend([[-0.5, 0.5]] * 3)
        lim.extend([[0, 2 * m.pi]] * 3)
        if self._TASK != _capi.TASK_REACH:
            lim.extend(wl[:3])
        assert len(lim) == len(lim32)
        assert np.allclose(np.array(lim, dtype=np.float32), np.array(lim32, dtype=np.float32), atol=0, rtol=0), \
            "engine observation limits differ from the Python-side limits"
        return lim

    @property
    def _tg_pose(self):
        return self._target_pose
