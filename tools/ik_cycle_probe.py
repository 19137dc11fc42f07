#!/usr/bin/env python
"""At which iteration does an iCub env's damped-least-squares IK sequence become periodic?

q_{k+1} = F(q_k) is a deterministic function of the joint angles alone (target and model are fixed during the call), so once
q_{k+1} == q_k (fixed point) or q_{k+1} == q_{k-1} (two-cycle: a last-bit flip-flop) bit for bit, the rest of the <= 100 iterations is known
without running it, and a wave whose 64 envs have all converged or become periodic can leave the loop with the exact result.  This probe
runs the CPU lane-emulation build with -DPBRE_IK_PROBE (tests/host_emu: build/libpbre_emu_ikprobe.so) on an iCub-push batch in the
stationary protocol (de-synchronised episode clocks, i.i.d. U(-1,1) Cartesian actions, in-kernel auto-reset) and prints the histogram of
the iteration at which each IK call converged (residual < 1 mm) / hit a fixed point / hit a two-cycle / did none of these within the cap.
Test infrastructure: the product never loads this library.
    python tools/ik_cycle_probe.py [--envs 128] [--steps 600] [--out profiles/r04_ik_cycle_hist.json]"""
import argparse, ctypes as C, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=128); ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--max-steps", type=int, default=500); ap.add_argument("--skip", type=int, default=300, help="steps before the histogram starts (stationary mix)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    os.environ["PBRE_ICUB_LANE"] = "1"
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "host_emu"), "build/libpbre_emu_ikprobe.so"])
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import icub_table
    import parity
    path = os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu_ikprobe.so")
    lib = _capi.load(path); raw = C.CDLL(path)
    tbl, model, info = icub_table("l")
    ov = parity.icub_overrides(info, "l", 1, 0, 1)
    n = a.envs
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, robot=_capi.ROBOT_ICUB, lib=lib, flags=_capi.F_AUTO_RESET, obj_pose_rnd_std=0.05,
                       tg_pose_rnd_std=0.2, max_steps=a.max_steps, **ov)
    eng.reset()
    st = eng.get_state()
    st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, a.max_steps, n).astype(np.float32)
    eng.set_state(st)
    h = (C.c_long * 512)()
    rng = np.random.default_rng(7)
    for k in range(a.steps):
        if k == a.skip:
            raw.pbre_ik_probe_hist(h, 1)
        eng.step(rng.uniform(-1, 1, (n, eng.act_dim)).astype(np.float32))
    raw.pbre_ik_probe_hist(h, 0)
    H = np.array(list(h), dtype=np.int64).reshape(4, 128)
    tot = int(H.sum())
    names = ["converged (residual < 1 mm)", "fixed point", "two-cycle", "none within the cap"]
    res = {"envs": n, "steps_counted": a.steps - a.skip, "ik_calls": tot}
    for k, nm in enumerate(names):
        cnt = int(H[k].sum())
        its = np.repeat(np.arange(128), H[k])
        res[nm] = {"calls": cnt, "frac": cnt / max(tot, 1), "iteration_percentiles_50_90_99_max": [int(np.percentile(its, p)) for p in (50, 90, 99, 100)] if cnt else None}
    # what a wave-uniform exit could use: an env is "done" at its recorded iteration (kinds 0-2) or never (kind 3)
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
