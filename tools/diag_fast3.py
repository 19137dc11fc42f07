"""Diagnostic: PBRE_FAST3=1 against =0 in IK mode, where do the two engines first differ (env, step, class of the env)?  GPU box."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "pybullet-robot-envs_amd")
import numpy as np, parity
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table, PANDA_SPHERES
tbl, model = panda_table()
panda = {"table": tbl, "model": model, "spheres": PANDA_SPHERES}
lib = _capi.load()
n = 4096
for trial in range(6):
    kw = dict(task=1, num_envs=n, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, lib=lib, flags=_capi.F_AUTO_RESET | int(os.environ.get("EXTRA_FLAGS", "0")), max_steps=40, use_ik=1)
    os.environ["PBRE_FAST3"] = os.environ.get("A_FAST3", "1"); a = _capi.Engine(tbl, **kw)
    os.environ["PBRE_FAST3"] = "0"; b = _capi.Engine(tbl, **kw)
    a.reset(); b.reset()
    _, ora = parity.make_pair(_capi.Engine, lib, tbl, 1)
    base, _ = ora.batch_reset(1)
    S = parity.contact_states(ora, panda, base[0], np.random.default_rng(1), 8, 8).astype(np.float32)
    st = a.get_state(); st[:len(S), :S.shape[1]] = S
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(9 + trial)
    first = None
    for k in range(60):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        sa0 = a.get_state()
        ra, rb = a.step(act), b.step(act)
        sa, sb = a.get_state(), b.get_state()
        d = np.abs(sa - sb).max(1)
        if d.max() > 0:
            bad = np.nonzero(d)[0]
            from pybullet_robot_envs.model import contacts
            print("trial", trial, "step", k, "envs", bad[:8], "max diff", d.max(), "cols", np.nonzero(np.abs(sa[bad[0]] - sb[bad[0]]))[0][:12])
            first = k
            break
    print("trial", trial, "first mismatch:", first, "fast3 steps", a.kernel_info()[8], "complex now", a.kernel_info()[5])
    a.close(); b.close()
