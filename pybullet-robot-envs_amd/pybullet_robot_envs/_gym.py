"""gym compatibility layer.  The reference depends on gym==0.12.5 (reference requirements.txt:1) for
`gym.Env`, `gym.GoalEnv`, `spaces.Box/Dict`, `seeding.np_random` and `register`.  gym is used when it is
importable; otherwise this minimal stand-in provides the same names so that the env classes keep the
reference's surface (observation_space.low/high/shape/dtype, action_space.sample(), seed() -> [seed])."""
import numpy as np

try:  # pragma: no cover - depends on the installation
    import gym as _gym
    from gym import spaces
    from gym.utils import seeding
    Env, GoalEnv = _gym.Env, getattr(_gym, "GoalEnv", _gym.Env)
    register = _gym.envs.registration.register
    HAVE_GYM = True
except Exception:  # gym absent (this image): internal shim
    HAVE_GYM = False

    class Env(object):
        metadata = {'render.modes': []}
        reward_range = (-float('inf'), float('inf'))
        spec = None
        action_space = None
        observation_space = None

        def step(self, action):
            raise NotImplementedError

        def reset(self):
            raise NotImplementedError

        def render(self, mode='human'):
            raise NotImplementedError

        def close(self):
            pass

        def seed(self, seed=None):
            return

        @property
        def unwrapped(self):
            return self

    class GoalEnv(Env):
        def compute_reward(self, achieved_goal, desired_goal, info):
            raise NotImplementedError

    class _Space(object):
        def __init__(self, shape=None, dtype=None):
            self.shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self.np_random = np.random.RandomState()

        def seed(self, seed=None):
            self.np_random.seed(seed)

    class Box(_Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            dtype = np.float32 if dtype is None else dtype
            if shape is None:
                assert np.shape(low) == np.shape(high)
                shape = np.shape(low)
            else:
                assert np.isscalar(low) and np.isscalar(high)
                low = low + np.zeros(shape)
                high = high + np.zeros(shape)
            super(Box, self).__init__(shape, dtype)
            self.low = np.asarray(low).astype(self.dtype)
            self.high = np.asarray(high).astype(self.dtype)

        def sample(self):
            return self.np_random.uniform(low=self.low, high=self.high, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high)

        def __repr__(self):
            return "Box" + str(self.shape)

    class Dict(_Space):
        def __init__(self, spaces_=None, **kw):
            super(Dict, self).__init__(None, None)
            self.spaces = dict(spaces_ or {}, **kw)

        def __getitem__(self, k):
            return self.spaces[k]

        def sample(self):
            return dict((k, s.sample()) for k, s in self.spaces.items())

    class spaces(object):
        Box = Box
        Dict = Dict

    class seeding(object):
        @staticmethod
        def np_random(seed=None):
            if seed is None:
                seed = int(np.random.SeedSequence().entropy % (2 ** 32))
            return np.random.RandomState(int(seed) % (2 ** 32)), seed

    _REGISTRY = {}

    def register(id, entry_point=None, max_episode_steps=None, kwargs=None, **_):
        _REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=dict(kwargs or {}))

    def make(id, **kw):
        import importlib
        spec = _REGISTRY[id.split(':')[-1]]
        mod, cls = spec['entry_point'].split(':')
        args = dict(spec['kwargs'])
        args.update(kw)
        return getattr(importlib.import_module(mod), cls)(**args)
