#!/usr/bin/env python
"""Where a lone wave's time goes: cycles per phase of lane 0 of block 0 of the Panda step kernels (Core::step inside k_row_list, the robot
and object waves of k_fast_pair), summed over the timed launches of a stationary batch.  Needs the probe build:
    tools/build_variant.sh probe "-DPBRE_PHASE_PROBE"
    PBRE_LIB=$(pwd)/pybullet-robot-envs_amd/csrc/libpbre_probe.so python tools/phase_probe.py --envs 16384
(s_memtime ticks; the tool calibrates them against a spin kernel-free wall clock: ticks of the whole step / HIP-event time.)"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))

NAMES = {0: "row: loads, motor targets", 1: "row: forward kinematics", 2: "row: velocities / bias forces / inertias", 3: "row: subtree sums",
         4: "row: CRBA", 5: "row: M^-1 (Gauss-Jordan)", 6: "row: v*, object dynamics", 7: "row: collision detection", 8: "row: constraint rows",
         9: "row: the 150 sweeps", 10: "row: integration + store", 11: "row: Fast::finish on lane 0",
         16: "robot wave: forward sweep", 17: "robot wave: CRBA + M^-1", 18: "robot wave: motor block (closed form)", 19: "robot wave: (object setup: none)",
         20: "robot wave: object tests, observation, row", 21: "robot wave: integration", 22: "robot wave: kinematics of the new state",
         23: "robot wave: waiting at the barrier", 24: "object wave: (robot sweep: none)", 25: "object wave: -", 26: "object wave: -",
         27: "object wave: unconstrained velocity, candidates, rows", 28: "object wave: the 150 sweeps", 29: "object wave: integration"}


def main():
    # the probes sample block 0 of the two-kernel step's kernels (in k_fused's 64-thread grid block 0 is an object wave): probe that step
    os.environ.setdefault("PBRE_FUSED", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--preroll", type=int, default=1100)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    import torch
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table
    lib = _capi.load()
    lib.pbre_debug_probe.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    tbl, _ = panda_table()
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    n = a.envs
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
    eng.reset()
    st = eng.get_state()
    st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)
    eng.set_state(st)
    out = torch.zeros((n, eng.obs_dim + 2), device=dev)
    act = torch.empty((n, eng.act_dim), device=dev)
    for _ in range(a.preroll):
        act.uniform_(-1, 1)
        eng.step_device(act.data_ptr(), out.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.pbre_debug_probe(buf, 1)
    pool = torch.rand((a.steps, n, eng.act_dim), device=dev) * 2 - 1
    t0 = time.perf_counter()
    for k in range(a.steps):
        eng.step_device(pool[k].data_ptr(), out.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    lib.pbre_debug_probe(buf, 0)
    v = [int(x) for x in buf]
    res = {"envs": n, "ms_per_step": round(ms, 4), "complex_envs": int(eng.kernel_info()[5]), "steps": a.steps,
           "ticks_per_step": {NAMES.get(i, str(i)): round(v[i] / a.steps, 1) for i in range(32) if v[i]}}
    names = {11: "clamp-free motor stages left their bound: restart", 12: "clamp-free", 13: "robot-only chain", 14: "two zipped chains", 15: "loops with per-slot tests"}
    res["row_wave_paths"] = {names[k]: {"waves_per_step": round(v[32 + k] / a.steps, 3), "sweep_ticks_per_wave": round(v[48 + k] / max(1, v[32 + k]))} for k in names if v[32 + k]}
    row = sum(v[0:12]) / a.steps
    res["row_wave_ticks_per_step"] = round(row, 1)
    res["object_wave_ticks_per_step"] = round(sum(v[24:30]) / a.steps, 1)
    res["robot_wave_ticks_per_step"] = round(sum(v[16:24]) / a.steps, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
