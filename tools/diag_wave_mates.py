"""Diagnostic: does a simple env's result depend on the other envs of its k_fast wave (a tilted cube with fewer than four table contacts / a
sliding cube in the same wave switch the wave to the other copy of the solver loop)?  Prints the largest difference of the untouched
envs between two engines over 40 steps (expected 0).  GPU box."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "pybullet-robot-envs_amd")
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
lib = _capi.load()
n = 64
kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=lib)
a = _capi.Engine(tbl, **kw); b = _capi.Engine(tbl, **kw)
a.reset(); b.reset()
st = a.get_state()
sb = st.copy()
# env 1 of engine b: the cube tilted by 20 degrees about x and lifted a little: fewer than four object-table contacts
ang = np.deg2rad(20.0)
sb[1, 12:16] = [np.sin(ang / 2), 0, 0, np.cos(ang / 2)]
sb[1, 11] += 0.01
# env 2 of engine b: the cube moving (sliding: clamped friction rows)
sb[2, 25:28] = [0.3, -0.2, 0.0]
b.set_state(sb); a.set_state(st)
rng = np.random.default_rng(0)
worst = 0.0
for k in range(40):
    act = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    a.step(act); b.step(act)
    sa_, sb_ = a.get_state(), b.get_state()
    others = np.ones(n, bool); others[[1, 2]] = False
    d = np.abs(sa_[others] - sb_[others]).max()
    worst = max(worst, float(d))
print("max difference of the untouched envs between the two engines over 40 steps:", worst)
