#!/usr/bin/env python
"""How many explicit sweeps does the object block need before its closed form (Fast::obj_closed) passes its validity bound?  CPU emulation
builds with -DPBRE_OC_K=K, the bench's stationary protocol in small (i.i.d. actions, auto-reset, de-synchronised episode clocks): fraction of
the simple-class lane-steps whose bound fails (those run the explicit rows for the remaining sweeps -- and on the GPU so does their wave).
    python tools/oc_accept_probe.py [--ks 22,16,12,8,4] [--envs 256] [--preroll 300] [--steps 60]"""
import argparse, ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="22,16,12,8,4")
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--preroll", type=int, default=300)
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table
    tbl, _ = panda_table()
    emu = os.path.join(ROOT, "tests", "host_emu")
    csrc = os.path.join(ROOT, "pybullet-robot-envs_amd", "csrc")
    for k in [int(x) for x in a.ks.split(",")]:
        so = os.path.join(emu, "build", "libpbre_emu_ock%d.so" % k)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DPBRE_OC_K=%d" % k, "-o", so, os.path.join(emu, "emu_capi.cpp"), "-ldl"], cwd=emu)
        lib = _capi.load(so)
        lib.pbre_emu_oc_stats.argtypes = [C.POINTER(C.c_long), C.c_int]
        eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=a.envs, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET, lib=lib)
        eng.reset()
        st = eng.get_state()
        st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, a.envs).astype(np.float32)
        eng.set_state(st)
        rng = np.random.default_rng(5)
        out = (C.c_long * 2)()
        for i in range(a.preroll + a.steps):
            if i == a.preroll:
                lib.pbre_emu_oc_stats(out, 1)
            eng.step(rng.uniform(-1, 1, (a.envs, eng.act_dim)).astype(np.float32))
        lib.pbre_emu_oc_stats(out, 1)
        fail, ok = out[0], out[1]
        waves = (1.0 - (1.0 - fail / max(1, fail + ok)) ** 64)
        print(json.dumps({"explicit_sweeps": k, "lane_steps": fail + ok, "failed": fail, "failed_frac": fail / max(1, fail + ok),
                          "waves_with_a_failing_lane_if_independent": waves}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
