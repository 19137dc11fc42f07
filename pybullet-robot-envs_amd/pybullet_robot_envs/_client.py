"""Physics "clients".  In the reference a client id comes from `p.connect(...)` and selects a PyBullet world
(reference panda_push_gym_env.py:56-62).  Here a client id selects a batched-engine session: the robot and
world wrappers register their configuration on it and the task env builds one `_capi.Engine` from both."""
import itertools

_clients = {}
_ids = itertools.count()


class Client(object):
    def __init__(self, num_envs=1, device_id=0, env_id_base=0, seed=1234, lib=None):
        self.num_envs = int(num_envs)
        # `device_id` may be one HIP device ordinal or a list of them (the Gym classes' `devices=[...]`): the batch is then sharded
        # over those GPUs from this process (_capi.MultiEngine)
        self.devices = [int(d) for d in device_id] if isinstance(device_id, (list, tuple)) else None
        self.device_id = self.devices[0] if self.devices else int(device_id)
        self.env_id_base = int(env_id_base)
        self.seed = int(seed)
        self.lib = lib
        self.robot = None
        self.world = None
        self.engine = None

    def require_engine(self):
        if self.engine is None:
            raise RuntimeError("the batched engine of this client has not been built yet "
                               "(construct a task env, e.g. pandaPushGymEnv, which owns the simulation)")
        return self.engine


def connect(num_envs=1, device_id=0, env_id_base=0, seed=1234, lib=None):
    """Counterpart of `p.connect(p.DIRECT)`: returns an integer physicsClientId."""
    cid = next(_ids)
    _clients[cid] = Client(num_envs, device_id, env_id_base, seed, lib)
    return cid


def get(cid):
    try:
        return _clients[cid]
    except KeyError:
        raise RuntimeError("unknown physicsClientId %r (use pybullet_robot_envs.connect())" % (cid,))


def disconnect(cid):
    c = _clients.pop(cid, None)
    if c is not None and c.engine is not None:
        c.engine.close()
