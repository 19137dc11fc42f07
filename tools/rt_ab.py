"""Residual-exit step, two builds of libpbre.so side by side: stationary step time (one launch and two kernels) and a checksum of rows and state
after the same 300 steps.   usage: python tools/rt_ab.py <envs> <lib> [<lib> ...]"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table

n = int(sys.argv[1])
dev = torch.device("cuda", 0)
tbl, _ = panda_table()
stream = torch.cuda.Stream(device=dev)
gen = torch.Generator(device=dev); gen.manual_seed(1)
act = torch.rand((300, n, 7), device=dev, generator=gen) * 2 - 1
for path in sys.argv[2:]:
    lib = _capi.load(path)
    for fused in ("1", "0"):
        os.environ["PBRE_FUSED"] = fused
        eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET, lib=lib)
        eng.reset()
        st = eng.get_state()
        st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)
        eng.set_state(st)
        out = torch.zeros((n, eng.obs_dim + 2), device=dev)
        torch.cuda.synchronize()
        for k in range(600):
            eng.step_device(act[k % 300].data_ptr(), out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        eng.set_physics(solver_residual_threshold=1e-7)
        for k in range(100):
            eng.step_device(act[k].data_ptr(), out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(100, 300):
            eng.step_device(act[k].data_ptr(), out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 200 * 1e3
        crc = zlib.crc32(out.cpu().numpy().tobytes()) ^ zlib.crc32(eng.get_state().tobytes())
        sw = eng.get_sweeps()
        print("%-28s envs %6d fused %s: %.4f ms per RT step, checksum %08x, sweeps median %.0f cap %.4f, complex %d" % (os.path.basename(path), n, fused, ms, crc, np.median(sw), float((sw >= 150).mean()), eng.kernel_info()[5]))
        eng.close()
