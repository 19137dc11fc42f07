#!/usr/bin/env python
"""Step time vs number of complex envs (robot in contact with the object / the table) for the two complex-env kernels
(k_row_list: rows, k_fast_rc: lanes), 131072 Panda-push envs on one MI355X.  Contact states are templates found with the
oracle (tests/scenarios.py) and tiled over the chosen fraction of the batch; every step restarts from the same states."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orc, parity
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table, PANDA_SPHERES
tbl, model = panda_table()
panda = {"table": tbl, "model": model, "spheres": PANDA_SPHERES}
ora = orc.Oracle(tbl, task=1)
base, _ = ora.batch_reset(1)
rng = np.random.default_rng(1)
tmpl = parity.contact_states(ora, panda, base[0], rng, 8, 8).astype(np.float32)
n = 131072
dev = torch.device("cuda", 0)
act = torch.rand((n, 7), device=dev) * 2 - 1
for frac in (0.0, 0.0002, 0.002, 0.01, 0.03, 0.08, 0.2):
    row = []
    for name, fl in (("rows", _capi.F_COMPLEX_ROWS), ("lanes", _capi.F_COMPLEX_LANES)):
        eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, flags=fl, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
        eng.reset()
        S = eng.get_state()
        k = int(round(frac * n))
        idx = np.random.default_rng(2).choice(n, k, replace=False)
        S[idx] = tmpl[np.arange(k) % len(tmpl)]
        out = torch.zeros((n, eng.obs_dim + 2), device=dev)
        s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
        ts = []
        for rep in range(6):
            eng.set_state(S)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.step_device(act.data_ptr(), out.data_ptr(), s.cuda_stream); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        row.append("%s %.3f ms (complex %d)" % (name, float(np.median(ts[1:])), eng.kernel_info()[5] if False else k))
    print("complex frac %.4f: " % frac + " | ".join(row))
