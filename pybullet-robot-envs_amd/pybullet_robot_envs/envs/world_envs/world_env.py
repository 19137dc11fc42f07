"""WorldEnv -- table + object side of the scene (reference pybullet_robot_envs/envs/world_envs/world_env.py).

Same constructor signature and method surface as the reference class; the simulation itself is the
batched HIP engine owned by the task env.  The scene constants the reference reads from `pybullet_data`
(absent from the reference checkout) are the engine's documented stand-ins: table top at h = 0.625
(world_env.py:68-69), `cube_small` = 5 cm / 0.1 kg box.  Every other `obj_name` of get_objects_list() / get_ycb_objects_list() is a box
with that object's approximate bounding dimensions, mass and friction (model/objects.py: the meshes are not available), so the name
changes the dynamics; `YcbWorldEnv` / `SqWorldEnv` select from the YCB table."""
import math as m

import numpy as np

from pybullet_robot_envs import _client
from pybullet_robot_envs._gym import seeding


def get_objects_list():
    return ['duck_vhacd', 'cube_small', 'teddy_vhacd', 'domino/domino']


def get_ycb_objects_list():
    """names of pybullet_object_models.ycb_objects (reference world_env.py:28-32 lists the package's folders)"""
    from pybullet_robot_envs.model.objects import YCB_OBJECTS
    return sorted(YCB_OBJECTS)


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
        self._physics_client_id = physicsClientId
        self._client = _client.get(physicsClientId)
        self._client.world = self
        self._ws_lim = tuple([list(i) for i in workspace_lim])
        self._h_table = 0.625
        self._obj_name = obj_name
        self.object_physics()           # unknown names fail here, as p.loadURDF would
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

    def object_physics(self):
        """pbre_physics fields of this world's object (box stand-in of `obj_name`, model/objects.py): the task env hands them to the
        engine -- the counterpart of load_object's p.loadURDF (world_env.py:76-84)."""
        from pybullet_robot_envs.model.objects import object_physics
        return object_physics(self._obj_name)

    def load_object(self, obj_name):
        """(world_env.py:76-84) switch the object; takes effect in the engine at once when the task env has built it.  In the reference
        load_object is part of WorldEnv.reset(): the new object only exists after a reset.  Here the engine refuses snapshot restarts
        (pbre_reset_snapshot, in-kernel auto-reset) until the next full reset() has settled the new object (its rest height differs)."""
        self._obj_name = obj_name
        ph = self.object_physics()
        if self._client.engine is not None:
            self._client.engine.set_physics(**ph)

    def get_object_shape_info(self):
        """(world_env.py:95-98) one p.getCollisionShapeData row of the stand-in: geometry type 3 = GEOM_BOX with the full extents, 2 =
        GEOM_SPHERE with the radius (x3), 4 = GEOM_CYLINDER with [height, radius, 0] (pybullet's conventions)"""
        ph = self.object_physics()
        h, shape = ph["obj_h"], ph.get("obj_shape", 0)
        if shape == 3:
            geom, dims = 5, [1.0, 1.0, 1.0]          # GEOM_MESH with the mesh scale (the object is the convex hull of its mesh, model/objects.py)
        elif shape == 1:
            geom, dims = 2, [h[0]] * 3
        elif shape == 2:
            geom, dims = 4, [2 * h[2], h[0], 0.0]
        else:
            geom, dims = 3, [2 * x for x in h]
        return [self.obj_id, -1, geom, dims, "", [0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]]

    def get_table_height(self):
        return self._h_table

    def get_workspace(self):
        return [i[:] for i in self._ws_lim]

    def get_observation_dimension(self):
        return 6 if self._control_eu_or_quat == 0 else 7

    def get_object_pose(self):
        """Batched object position [N,3] and quaternion [N,4]."""
        eng = self._client.require_engine()
        st, o = eng.get_state(), eng.obj_off
        return st[:, o:o + 3].astype(np.float64), st[:, o + 3:o + 7].astype(np.float64)

    def get_observation(self):
        """Object position + Euler angles (or, with control_eu_or_quat=1, the quaternion) and their limits (world_env.py:109-126).
        A list of 6 (7) floats for a single env, an [N, 6 (7)] array for a batch."""
        observation_lim = []
        observation_lim.extend(self._ws_lim)
        if self._control_eu_or_quat == 0:
            observation_lim.extend([[-m.pi, m.pi], [-m.pi, m.pi], [-m.pi, m.pi]])
        else:
            observation_lim.extend([[-1, 1], [-1, 1], [-1, 1], [-1, 1]])
        if self._client.engine is None:       # before the task env built the engine: initial pose
            x_min, x_max = self._ws_lim[0][0] + 0.05, self._ws_lim[0][1] - 0.1          # _sample_pose, world_env.py:147-160
            y_min, y_max = self._ws_lim[1][0] + 0.05, self._ws_lim[1][1] - 0.05
            pos = np.array([[x_min + 0.5 * (x_max - x_min), y_min + 0.5 * (y_max - y_min), self._h_table + 0.07]])
            quat = np.array([[0.0, 0.0, m.sin(m.pi / 8), m.cos(m.pi / 8)]])
        else:
            pos, quat = self.get_object_pose()
        obs = np.concatenate([pos, euler_from_quat(quat) if self._control_eu_or_quat == 0 else quat], axis=1)
        if obs.shape[0] == 1:
            return list(obs[0]), observation_lim
        return obs, observation_lim

    def check_contact(self, body_id, obj_id=None):
        """Is the object in contact with `body_id` (world_env.py:128-134: `len(p.getContactPoints(obj_id, body_id)) > 0`)?  `body_id` is
        this world's `table_id` or the robot's `robot_id`.  A bool for a single env, an [N] bool array for a batch.  A query on the
        downloaded state with the step's own detection rule (distance below the contact margin): model/contacts.py."""
        from pybullet_robot_envs.model import contacts
        c = self._client
        eng = c.require_engine()
        robot = c.robot
        if body_id == self.table_id:
            bit = contacts.OBJECT_TABLE
        elif robot is not None and body_id == robot.robot_id:
            bit = contacts.ROBOT_OBJECT
        else:
            raise ValueError("check_contact: body_id %r is neither this world's table_id (%d) nor the robot's robot_id" % (body_id, self.table_id))
        no_obj = bool(eng.cfg.flags & 1)
        f = contacts.contact_flags(robot.robot_table, eng.get_state(), eng.obj_off, eng.get_physics(), no_obj)
        hit = (f & bit) != 0
        return bool(hit[0]) if hit.shape[0] == 1 else hit

    def debug_gui(self):
        pass

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]


class YcbWorldEnv(WorldEnv):
    """(world_env.py:179-196) a world whose object comes from the YCB set: box stand-ins of model/objects.py"""

    def __init__(self, physicsClientId, obj_name='YcbMustardBottle', obj_pose_rnd_std=0.05, workspace_lim=None, control_eu_or_quat=0):
        super(YcbWorldEnv, self).__init__(physicsClientId, obj_name, obj_pose_rnd_std, workspace_lim, control_eu_or_quat)


class SqWorldEnv(WorldEnv):
    """(world_env.py:199-216) superquadric approximations of the YCB objects: the same box stand-ins"""

    def __init__(self, physicsClientId, obj_name='YcbMustardBottle', obj_pose_rnd_std=0.05, workspace_lim=None, control_eu_or_quat=0):
        super(SqWorldEnv, self).__init__(physicsClientId, obj_name, obj_pose_rnd_std, workspace_lim, control_eu_or_quat)
