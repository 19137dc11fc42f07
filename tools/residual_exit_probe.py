#!/usr/bin/env python
"""After how many PGS sweeps would Bullet's residual test end the solver loop on this workload?

PyBullet documents `setPhysicsEngineParameter(solverResidualThreshold=...)` ("velocity threshold, if the maximum velocity-level error for
each constraint is below this threshold the solver will terminate (unless the solver hits the numSolverIterations); default 1e-7"), and
Bullet's loop compares the largest SQUARED velocity-level row change of a sweep with it [EXT-UNVERIFIED: PyBullet is absent here].  The
reference sets numSolverIterations=150 and leaves the threshold alone.  The engine and the oracle run all 150 sweeps (threshold 0: the
stricter reading of the workload); this probe rolls the fp64 oracle out in bench.py's protocol (Panda push, i.i.d. U(-1,1) joint actions,
de-synchronised episode clocks) and reports, per env-step, the sweep after which the test with 1e-7 would have fired
(orc_step_info.sweeps_to_1e7), and how far the states of an oracle that does leave the loop there drift from the 150-sweep oracle.
Test infrastructure (oracle only).     python tools/residual_exit_probe.py [--envs 64] [--steps 300] [--out profiles/r04_residual_exit_hist.json]"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64); ap.add_argument("--steps", type=int, default=300); ap.add_argument("--out", default="")
    a = ap.parse_args()
    import orc
    ora, _tbl = orc.panda_oracle()
    ora.task.obj_pose_rnd_std = 0.05; ora.task.tg_pose_rnd_std = 0.2
    n = a.envs
    st, _ = ora.batch_reset(n)
    rng = np.random.default_rng(11)
    hist = np.zeros(ora.params.solver_iters + 2, np.int64)
    drift = {"q": 0.0, "qd": 0.0, "obj_pos": 0.0, "obj_v": 0.0}
    nd = ora.ndof
    for k in range(a.steps):
        act = rng.uniform(-1, 1, (n, 7))
        # one step of an oracle that leaves the loop as PyBullet presumably does, from the same state: how far apart are the results?
        ora.params.solver_residual_threshold = 1e-7
        s2, _ = ora.batch_step(st, act)
        ora.params.solver_residual_threshold = 0.0
        st_new, out, used, to7 = ora.batch_step_sweeps(st, act)
        assert (used == ora.params.solver_iters).all() or (used <= ora.params.solver_iters).all()
        np.add.at(hist, np.minimum(to7, ora.params.solver_iters + 1), 1)
        vo = 16      # velocity record of the 16-lane layout
        drift["q"] = max(drift["q"], float(np.abs(s2[:, :nd] - st_new[:, :nd]).max()))
        drift["qd"] = max(drift["qd"], float(np.abs(s2[:, vo:vo + nd] - st_new[:, vo:vo + nd]).max()))
        drift["obj_pos"] = max(drift["obj_pos"], float(np.abs(s2[:, nd:nd + 3] - st_new[:, nd:nd + 3]).max()))
        drift["obj_v"] = max(drift["obj_v"], float(np.abs(s2[:, vo + nd:vo + nd + 6] - st_new[:, vo + nd:vo + nd + 6]).max()))
        st = st_new
        # episodes end and restart as in the engine's auto-reset: re-sample the finished envs
        done = out[:, -1] > 0.5
        if done.any():
            fresh, _ = ora.batch_reset(int(done.sum()), env_id0=1000 * (k + 1))
            st[done] = fresh
    its = np.repeat(np.arange(hist.size), hist)
    never = int(hist[-1])
    res = {"workload": "pandaPushGymEnv joint control, fp64 oracle, %d envs x %d steps, i.i.d. U(-1,1) actions" % (n, a.steps),
           "threshold_on_the_squared_residual": 1e-7, "solver_iters": int(ora.params.solver_iters), "env_steps": int(hist.sum()),
           "sweeps_until_the_test_fires": {"median": int(np.percentile(its, 50)), "p90": int(np.percentile(its, 90)), "p99": int(np.percentile(its, 99)),
                                           "max": int(its.max()), "never_within_the_cap": never},
           "hist": {str(i): int(c) for i, c in enumerate(hist) if c},
           "one_step_difference_between_leaving_there_and_running_all_sweeps": drift,
           "note": "EXT-UNVERIFIED: PyBullet is absent; the test and its default are restated from Bullet's solver loop and PyBullet's documentation"}
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
