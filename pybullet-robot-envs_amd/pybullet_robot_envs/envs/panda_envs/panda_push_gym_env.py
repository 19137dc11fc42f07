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
                 num_envs=1, device_id=0, env_id_base=0, seed=1234, auto_reset=False, _lib=None):
        self._target_dist_max = 0.3
        self._setup(numControlledJoints, use_IK, action_repeat, obj_name, renders, max_steps, obj_pose_rnd_std,
                    tg_pose_rnd_std, includeVelObs, 0.1, num_envs, device_id, env_id_base, seed, _lib, auto_reset)

    def change_physics_params(self, obj_mass, obj_friction, obj_dumping, robot_damping):
        """Domain randomisation hook of the reference (panda_push_gym_env.py:362-368: p.changeDynamics on the object's
        mass / lateral friction / linear damping and on the arm links' linear damping), applied to every env of the
        batch.  The engine has one linear-damping constant for all bodies (Bullet's default 0.04 each), so
        `obj_dumping` and `robot_damping` must agree.  The cube's inertia is rescaled with its mass."""
        if obj_dumping != robot_damping:
            raise NotImplementedError("separate object / robot linear damping is not supported (one batch-uniform constant)")
        ph = self._engine.get_physics()
        scale = float(obj_mass) / ph.obj_mass
        self._engine.set_physics(obj_mass=float(obj_mass), obj_mu=float(obj_friction), lin_damping=float(obj_dumping),
                                 obj_inertia=[ph.obj_inertia[i] * scale for i in range(3)])
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
