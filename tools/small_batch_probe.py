"""Which kernel steps a SMALL batch fastest?  BASELINE config 2 (Panda reach, object frozen, 4096 envs) and small push batches: the lane-per-env
path (k_fast / pair: one env per lane, 64 envs per wave -- 64 waves at 4096 envs, one wave's latency) against the general 16-lane row kernel
(PBRE_F_FORCE_GENERAL: 4 envs per wave, an env's rows spread over 16 lanes).   usage: python tools/small_batch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table

dev = torch.device("cuda", 0)
tbl, _ = panda_table()
stream = torch.cuda.Stream(device=dev)
for task, name, base in ((_capi.TASK_REACH, "reach, object frozen (config 2)", _capi.F_NO_OBJECT), (_capi.TASK_PUSH, "push", 0)):
    for n in (1024, 4096, 8192, 16384):
        for extra, kn in ((0, "lane-per-env"), (_capi.F_FORCE_GENERAL, "general row kernel")):
            eng = _capi.Engine(tbl, task=task, num_envs=n, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=base | _capi.F_AUTO_RESET | extra)
            eng.reset()
            st = eng.get_state()
            st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)
            eng.set_state(st)
            gen = torch.Generator(device=dev); gen.manual_seed(1)
            out = torch.zeros((n, eng.obs_dim + 2), device=dev)
            act = torch.empty((n, eng.act_dim), device=dev)
            torch.cuda.synchronize()
            for k in range(600):
                act.uniform_(-1, 1, generator=gen)
                eng.step_device(act.data_ptr(), out.data_ptr(), stream.cuda_stream)
            pool = torch.rand((200, n, eng.act_dim), device=dev, generator=gen) * 2 - 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(200):
                eng.step_device(pool[k].data_ptr(), out.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 200 * 1e3
            print("%-32s %6d envs  %-20s %.4f ms per step = %6.1f M env-steps/s" % (name, n, kn, ms, n / ms / 1e3))
            eng.close()
