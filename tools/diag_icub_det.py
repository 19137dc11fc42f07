"""Diagnostic: two identical iCub push engines side by side (crafted contact states among the batch, random actions, auto-reset): do they stay
bit-identical?  PBRE_ICUB_LANE=1 / 0 selects the pipeline / the lane-group kernel.  GPU box."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "pybullet-robot-envs_amd")
import numpy as np, parity
from pybullet_robot_envs import _capi
lib = _capi.load()
for use_ik in (0, 1):
    n = 4096
    eng0, ora, info = parity.make_icub_pair(_capi.Engine, lib, 1, task=1, use_ik=0, obj_std=0.0, tg_std=0.2)
    base, _ = ora.batch_reset(1)
    S, kinds = parity.icub_contact_states(ora, info, base[0], np.random.default_rng(21), 12, 12, 12, 12, "l")
    kw = dict(task=1, use_ik=use_ik, obj_std=0.05, tg_std=0.2, max_steps=60, flags=_capi.F_AUTO_RESET)
    a, _, _ = parity.make_icub_pair(_capi.Engine, lib, n, **kw)
    b, _, _ = parity.make_icub_pair(_capi.Engine, lib, n, **kw)
    a.reset(); b.reset()
    st = a.get_state(); st[:len(S), :S.shape[1]] = S.astype(np.float32)
    a.set_state(st); b.set_state(st)
    rng = np.random.default_rng(3)
    first = None
    for k in range(120):
        act = rng.uniform(-1, 1, (n, a.act_dim)).astype(np.float32)
        ra, rb = a.step(act), b.step(act)
        d = np.abs(a.get_state() - b.get_state()).max(1)
        if d.max() > 0 and first is None:
            first = (k, np.nonzero(d)[0][:6].tolist(), float(d.max()))
            break
    print("use_ik", use_ik, "lane", os.environ.get("PBRE_ICUB_LANE"), "first mismatch:", first, "complex now", a.kernel_info()[5])
