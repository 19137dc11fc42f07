"""Sharding invariance at the headline size: one engine with 131072 Panda-push envs (k_fast) against two engines with 65536 envs each and
env_id_base 0 / 65536 (k_fast_pair steps their simple envs), the same action stream, stationary protocol, auto-reset: are the output rows
bit-identical step by step?  GPU box.     python tools/diag_sharding_at_scale.py [steps]"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "pybullet-robot-envs_amd")
import numpy as np, torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
tbl, _ = panda_table()
n, h = 131072, 65536
dev = torch.device("cuda", 0)
kw = dict(task=1, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET, seed=1234)
whole = _capi.Engine(tbl, num_envs=n, **kw)
parts = [_capi.Engine(tbl, num_envs=h, env_id_base=k * h, **kw) for k in range(2)]
clocks = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)
for e, sl in [(whole, slice(0, n))] + [(parts[k], slice(k * h, (k + 1) * h)) for k in range(2)]:
    e.reset()
    st = e.get_state(); st[:, e.x_off + 3] = clocks[sl]; e.set_state(st)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ow = whole.obs_dim + 2
out_w = torch.zeros((n, ow), device=dev); out_p = torch.zeros((n, ow), device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(5)
first = None
for k in range(steps):
    act = torch.rand((n, 7), device=dev, generator=gen) * 2 - 1
    whole.step_device(act.data_ptr(), out_w.data_ptr(), s.cuda_stream)
    for j in range(2):
        parts[j].step_device(act[j * h:(j + 1) * h].data_ptr(), out_p[j * h:(j + 1) * h].data_ptr(), s.cuda_stream)
    if k % 50 == 49 or k == steps - 1:
        torch.cuda.synchronize()
        if first is None and not torch.equal(out_w, out_p):
            first = k
st_w = whole.get_state(); st_p = np.concatenate([p.get_state() for p in parts])
print("first differing output rows at (checked every 50 steps):", first, "| states equal:", bool(np.array_equal(st_w, st_p)),
      "| pair-kernel steps of a shard:", parts[0].kernel_info()[10], "| 3-wave steps of the whole batch:", whole.kernel_info()[8],
      "| complex env-steps (whole):", whole.kernel_info()[7])
