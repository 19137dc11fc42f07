"""WorldEnv -- table + object side of the scene (reference pybullet_robot_envs/envs/world_envs/world_env.py).

Same constructor signature and method surface as the reference class; the simulation itself is the
batched HIP engine owned by the task env.  The scene constants the reference reads from `pybullet_data`
(absent from the reference checkout) are the engine's documented stand-ins: table top at h = 0.625
(world_env.py:68-69), `cube_small` = 5 cm / 0.1 kg box.  Other `obj_name`s of get_objects_list() are
accepted for API parity but simulated with the same box (their meshes are not available)."""
import math as m

import numpy as np

from pybullet_robot_envs import _client
from pybullet_robot_envs._gym import seeding


def get_objects_list():
    return ['duck_vhacd', 'cube_small', 'teddy_vhacd', 'domino/domino']


def get_ycb_objects_list():
    raise NotImplementedError("pybullet_object_models (YCB meshes) is not available to this engine")


def euler_from_quat(q):
    """pybullet.getEulerFromQuaternion on [..., 4] arrays (x, y, z, w)."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    sarg = -2.0 * (x * z - w * y)
    roll = np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z)
    pitch = np.arcsin(np.clip(sarg, -1, 1))
    yaw = np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)
    lo, hi = sarg <= -0.99999, sarg >= 0.99999
    roll = np.where(lo | hi, 0.0, roll)
    pitch = np.where(lo, -0.5 * m.pi, np.where(hi, 0.5 * m.pi, pitch))
    yaw = np.where(lo, 2 * np.arctan2(x, -y), np.where(hi, 2 * np.arctan2(-x, y), yaw))
    return np.stack([roll, pitch, yaw], axis=-1)


class WorldEnv:

    def __init__(self,
                 physicsClientId,
                 obj_name='duck_vhacd',
                 obj_pose_rnd_std=0.05,
                 workspace_lim=None,
                 control_eu_or_quat=0):

        if workspace_lim is None:
            workspace_lim = [[0.25, 0.52], [-0.3, 0.3], [0.5, 1.0]]
        if control_eu_or_quat != 0:
            raise NotImplementedError("control_eu_or_quat=1 (quaternion observations) is not implemented")

        self._physics_client_id = physicsClientId
        self._client = _client.get(physicsClientId)
        self._client.world = self
        self._ws_lim = tuple([list(i) for i in workspace_lim])
        self._h_table = 0.625
        self._obj_name = obj_name
        self._obj_pose_rnd_std = obj_pose_rnd_std
        self._obj_init_pose = []
        self._control_eu_or_quat = control_eu_or_quat
        self.obj_id = 2
        self.table_id = 1

        self.seed()
        self.reset()

    def reset(self):
        # set ws limit on z according to table height (world_env.py:72); the object itself is (re)loaded by the
        # engine reset (pbre_reset), including _sample_pose
        self._ws_lim[2][:] = [self._h_table, self._h_table + 0.3]

    def get_table_height(self):
        return self._h_table

    def get_workspace(self):
        return [i[:] for i in self._ws_lim]

    def get_observation_dimension(self):
        return 6

    def get_object_pose(self):
        """Batched object position [N,3] and quaternion [N,4]."""
        eng = self._client.require_engine()
        st, o = eng.get_state(), eng.ndof
        return st[:, o:o + 3].astype(np.float64), st[:, o + 3:o + 7].astype(np.float64)

    def get_observation(self):
        """Object position + Euler angles and their limits (world_env.py:109-126).  A list of 6 floats for a
        single env, an [N, 6] array for a batch."""
        observation_lim = []
        observation_lim.extend(self._ws_lim)
        observation_lim.extend([[-m.pi, m.pi], [-m.pi, m.pi], [-m.pi, m.pi]])
        if self._client.engine is None:       # before the task env built the engine: initial pose
            x_min, x_max = self._ws_lim[0][0] + 0.05, self._ws_lim[0][1] - 0.1          # _sample_pose, world_env.py:147-160
            y_min, y_max = self._ws_lim[1][0] + 0.05, self._ws_lim[1][1] - 0.05
            pos = np.array([[x_min + 0.5 * (x_max - x_min), y_min + 0.5 * (y_max - y_min), self._h_table + 0.07]])
            quat = np.array([[0.0, 0.0, m.sin(m.pi / 8), m.cos(m.pi / 8)]])
        else:
            pos, quat = self.get_object_pose()
        obs = np.concatenate([pos, euler_from_quat(quat)], axis=1)
        if obs.shape[0] == 1:
            return list(obs[0]), observation_lim
        return obs, observation_lim

    def check_contact(self, body_id, obj_id=None):
        raise NotImplementedError("contact queries are not exposed by the batched engine")

    def debug_gui(self):
        pass

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]


class YcbWorldEnv(WorldEnv):
    def __init__(self, *a, **kw):
        raise NotImplementedError("YcbWorldEnv needs pybullet_object_models meshes, which this engine does not have")


class SqWorldEnv(WorldEnv):
    def __init__(self, *a, **kw):
        raise NotImplementedError("SqWorldEnv needs pybullet_object_models meshes, which this engine does not have")
