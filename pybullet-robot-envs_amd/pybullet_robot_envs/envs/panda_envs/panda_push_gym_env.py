"""pandaPushGymEnv (reference pybullet_robot_envs/envs/panda_envs/panda_push_gym_env.py)."""
import numpy as np

from pybullet_robot_envs import _capi
from pybullet_robot_envs.envs.panda_envs._base import PandaTaskBase
from pybullet_robot_envs.envs.world_envs.world_env import get_objects_list
from pybullet_robot_envs.envs.utils import goal_distance


class pandaPushGymEnv(PandaTaskBase):
    _TASK = _capi.TASK_PUSH

    def __init__(self,
                 numControlledJoints=7,
                 use_IK=0,
                 action_repeat=1,
                 obj_name=get_objects_list()[1],
                 renders=False,
                 max_steps=1000,
                 obj_pose_rnd_std=0.0,
                 tg_pose_rnd_std=0.0,
                 includeVelObs=True,
                 num_envs=1, device_id=0, env_id_base=0, seed=1234, auto_reset=False, _lib=None, devices=None):
        device_id = devices if devices is not None else device_id
        self._target_dist_max = 0.3
        self._setup(numControlledJoints, use_IK, action_repeat, obj_name, renders, max_steps, obj_pose_rnd_std,
                    tg_pose_rnd_std, includeVelObs, 0.1, num_envs, device_id, env_id_base, seed, _lib, auto_reset)

    def change_physics_params(self, obj_mass, obj_friction, obj_dumping, robot_damping):
        """Domain randomisation hook of the reference (panda_push_gym_env.py:362-368: p.changeDynamics on the object's mass /
        lateral friction / linear damping, then on the arm links' linear damping).  Scalars apply to every env of the batch, [N]
        arrays give every env its own object and its own arm damping (the reference calls this per env and episode); the values persist
        across resets.  The reference's own loop over the arm links reads a non-existent attribute (`_num_dof_no_fingers`, :366) and
        raises after the object was changed; here both parts are applied (every robot link gets `robot_damping`).  The cube's inertia is
        rescaled with its mass."""
        n = self._engine.num_envs
        rd = np.broadcast_to(np.asarray(robot_damping, np.float32), (n,))
        self._engine.set_physics_per_env(obj_mass=obj_mass, obj_mu=obj_friction, obj_lin_damping=obj_dumping, robot_lin_damping=rd)
        return 0

    # host-side restatements of the reference helpers on the current state (the GPU step already returns them)
    def _distances(self):
        st = self._engine.get_state().astype(np.float64)
        ee = self._engine.observe()[:, :3].astype(np.float64)
        return goal_distance(ee, st[:, 9:12]), goal_distance(st[:, 9:12], st[:, 32:35]), st

    def _termination(self):
        d1, d2, st = self._distances()
        done = (d2 <= self._target_dist_min) | (st[:, 36] != 0) | (st[:, 35] > self._max_steps)
        return self._squeeze(done.astype(np.float32))

    def _compute_reward(self):
        d1, d2, _ = self._distances()
        reward = np.where(d2 <= self._target_dist_min, np.float32(1000.0) + (100 - d2 * 80), -d1 - d2)
        return self._squeeze(reward)
