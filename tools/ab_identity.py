"""A/B of two builds of libpbre.so in one process: are the rows bit-identical, and how long does the stationary / fresh step take?
   usage: python tools/ab_identity.py <libA.so> <libB.so> [envs] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table

la, lb = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 1200
dev = torch.device("cuda", 0)
tbl, _ = panda_table()
libs = [_capi.load(la), _capi.load(lb)]
engs = [_capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, lib=l, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET) for l in libs]
stream = torch.cuda.Stream(device=dev)
outs = [torch.zeros((n, engs[0].obs_dim + 2), device=dev) for _ in engs]
for e in engs:
    e.reset()
    st = e.get_state()
    st[:, e.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)
    e.set_state(st)
gen = torch.Generator(device=dev); gen.manual_seed(7)
act = torch.empty((n, 7), device=dev)
torch.cuda.synchronize()
same = True
fresh = [0.0, 0.0]
for k in range(steps):
    act.uniform_(-1, 1, generator=gen)
    torch.cuda.synchronize()
    for i, e in enumerate(engs):
        t0 = time.perf_counter()
        e.step_device(act.data_ptr(), outs[i].data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        if 5 <= k < 25:
            fresh[i] += (time.perf_counter() - t0) / 20
    if k % 50 == 0 or k == steps - 1:
        same = same and bool(torch.equal(outs[0], outs[1]))
print("rows bit-identical over %d steps: %s   (states equal: %s)" % (steps, same, bool(np.array_equal(engs[0].get_state(), engs[1].get_state()))))
# timing: alternate blocks of 50 steps
pool = torch.rand((50, n, 7), device=dev, generator=gen) * 2 - 1
res = [[], []]
for rep in range(6):
    for i, e in enumerate(engs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(50):
            e.step_device(pool[k].data_ptr(), outs[i].data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / 50 * 1e3)
for i, nm in enumerate((la, lb)):
    print("%s: stationary ms per step median %.4f (min %.4f), fresh (steps 5..24, host-synchronised) %.4f" % (os.path.basename(nm), float(np.median(res[i])), min(res[i]), fresh[i] * 1e3))
