"""Stationary-mix step time of the Panda push batch as a function of the batch size around the machine-filling size (131072 envs =
2048 k_fast waves = 1024 SIMDs x 2 waves): is the 0.19 -> 0.25 ms step of the stationary mix the second round that the complex envs'
row waves force on a grid that fills every wave slot?   usage: python tools/tail_probe.py [--sizes 131072,129024,...] [--preroll 1000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="131072,130048,129024,126976,122880,98304,65536")
    ap.add_argument("--preroll", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--iters", type=int, default=0, help="solver iterations during the timed stationary steps only (0: unchanged, 150): what the solver loops cost")
    a = ap.parse_args()
    import torch
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table
    tbl, _ = panda_table()
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    rows = []
    for n in [int(x) for x in a.sizes.split(",")]:
        eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
        eng.reset()
        st = eng.get_state()
        st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)      # episodes of every age (bench.py)
        eng.set_state(st)
        gen = torch.Generator(device=dev); gen.manual_seed(1234)
        out = torch.zeros((n, eng.obs_dim + 2), device=dev)
        act = torch.empty((n, eng.act_dim), device=dev)
        res = {"envs": n, "k_fast_waves": (n + 63) // 64}
        for phase, cnt in (("fresh", 0), ("stationary", a.preroll)):
            for _ in range(cnt):
                act.uniform_(-1, 1, generator=gen)
                eng.step_device(act.data_ptr(), out.data_ptr(), side.cuda_stream)
            # i.i.d. actions resident in HBM, one slice per timed step (a short recycled pool biases every env's random walk and drives
            # the joints into their limits within a few hundred steps: bench.py)
            ns = a.steps if cnt else 20
            if cnt and a.iters:
                eng.set_physics(solver_iters=a.iters)
                res["timed_solver_iters"] = a.iters
            pool = torch.rand((ns + 5, n, eng.act_dim), device=dev, generator=gen) * 2 - 1
            for k in range(5):
                eng.step_device(pool[k].data_ptr(), out.data_ptr(), side.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(ns):
                eng.step_device(pool[5 + k].data_ptr(), out.data_ptr(), side.cuda_stream)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / ns * 1e3
            del pool
            res[phase] = {"ms_per_step": round(ms, 4), "M_env_steps_per_s": round(n / ms / 1e3, 1), "complex_envs": int(eng.kernel_info()[5])}
        res["finite"] = bool(torch.isfinite(out).all())
        print(json.dumps(res), flush=True)
        rows.append(res)
        eng.close()


if __name__ == "__main__":
    main()
