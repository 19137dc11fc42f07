#!/usr/bin/env python
"""Which of the [EXT-UNVERIFIED] constants would matter?  

Every Bullet-internal constant the oracle and the engine share (oracle/pbre_oracle.c: orc_default_params, tagged [EXT-UNVERIFIED]) is
replaced, one at a time, by a plausible alternative, and the fp64 ORACLE re-runs three pieces of the bench workload:
  (a) one step from identical post-reset states with i.i.d. U(-1,1) actions  (the parity tests' protocol),
  (b) a 60-step free-running rollout of the same actions,
  (c) the scripted push of tests/scenarios.py (approach + sweep through the cube, closed loop), where contacts decide the outcome.
Reported: the largest change of each observed quantity against the default constants -- to be read against the parity bounds of
tests/parity.py (TOL: q 1.5e-6 rad, qd 1.5e-4 rad/s, object position 3e-7 m, EE position 1.5e-6 m).  A constant whose alternative moves
the result by less than those bounds cannot be told apart by a PyBullet-pinned run at that protocol; the others say where to look first.
Test infrastructure (oracle only; no GPU).     python tools/ext_sensitivity.py [--envs 16] [--out profiles/r05_ext_sensitivity.json]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import orc          # noqa: E402
import scenarios    # noqa: E402
from pybullet_robot_envs.model.table import panda_table      # noqa: E402

ALTERNATIVES = [     # (name, setter, what the alternative stands for)
    ("erp 0.2 -> 0.1", lambda o: setattr(o.params, "erp", 0.1), "contact / limit ERP: btContactSolverInfo::m_erp2 is 0.2 for contacts, m_erp 0.2; 0.1 is the other value in the source tree"),
    ("erp 0.2 -> 0.8", lambda o: setattr(o.params, "erp", 0.8), "PyBullet's `erp` / `contactERP` parameters are documented as defaults 0.2; btMultiBody joint feedback uses up to 0.8"),
    ("linear_slop 1e-5 -> 0", lambda o: setattr(o.params, "linear_slop", 0.0), "btContactSolverInfo::m_linearSlop: -0.0 in btContactSolverInfoData's constructor of newer trees, 1e-5 after PyBullet's init"),
    ("linear_slop 1e-5 -> 1e-4", lambda o: setattr(o.params, "linear_slop", 1e-4), "allowedCcdPenetration-sized slop"),
    ("contact_margin 1e-3 -> 2e-3", lambda o: setattr(o.params, "contact_margin", 2e-3), "contact breaking threshold / URDF collision margin"),
    ("contact_margin 1e-3 -> 2e-2", lambda o: setattr(o.params, "contact_margin", 2e-2), "gContactBreakingThreshold 0.02 (contacts created that early carry positive distances)"),
    ("lin/ang damping 0.04 -> 0", lambda o: (setattr(o.params, "lin_damping", 0.0), setattr(o.params, "ang_damping", 0.0)), "btMultiBody m_linearDamping / m_angularDamping 0.04 (URDF importer) vs none"),
    ("max_coord_vel 100 -> 1e3", lambda o: setattr(o.params, "max_coord_vel", 1e3), "btMultiBody::m_maxCoordinateVelocity"),
    ("motor force 1e5 N -> 240 N (Panda effort)", lambda o: setattr(o.params, "max_motor_impulse", 240.0 / 240.0 * 1.0), "setJointMotorControl2 without `force`: PyBullet's default vs the URDF effort limit (87 N m arm joints; here 240 N dt as a stand-in)"),
    ("limit_max_impulse 100 -> 1e10", lambda o: setattr(o.params, "limit_max_impulse", 1e10), "btMultiBodyJointLimitConstraint max impulse"),
    ("solver_residual_threshold 0 -> 1e-7", lambda o: setattr(o.params, "solver_residual_threshold", 1e-7), "PyBullet's documented solverResidualThreshold default (engine option since round 5)"),
    ("solver_iters 150 -> 50", lambda o: setattr(o.params, "solver_iters", 50), "PyBullet's default numSolverIterations if the reference's setPhysicsEngineParameter did not take effect"),
    ("table_mu 0.5 -> 1.0", lambda o: setattr(o.params, "table_mu", 1.0), "table.urdf lateral_friction (pybullet_data; absent here): 0.5 is Bullet's default, some table URDFs set 1.0"),
    ("obj_mu 1.0 -> 0.5", lambda o: setattr(o.params, "obj_mu", 0.5), "cube_small.urdf lateral_friction 1.0 vs Bullet's default 0.5"),
    ("obj_mass 0.1 -> 0.05", lambda o: (setattr(o.params, "obj_mass", 0.05), [o.params.obj_inertia.__setitem__(k, o.params.obj_inertia[k] * 0.5) for k in range(3)]), "cube_small.urdf mass"),
    ("ik_damping 0.1 -> 0.01 (IK control only)", None, "BussIK's DLS damping; not on the joint-control bench path -- see tools/ik_cycle_probe.py"),
]


def quantities(st, ref):
    d = np.abs(st - ref)
    return {"q_rad": float(d[:, :9].max()), "qd_rad_s": float(d[:, 16:25].max()), "obj_pos_m": float(d[:, 9:12].max()),
            "obj_v_m_s": float(d[:, 25:28].max())}


def run(o, n, seed=3):
    rng = np.random.default_rng(seed)
    st0, _ = o.batch_reset(n)
    acts = rng.uniform(-1, 1, (60, n, 7))
    one, _ = o.batch_step(st0, acts[0])
    st = st0
    for k in range(60):
        st, _ = o.batch_step(st, acts[k])
    # scripted push, closed loop
    sp = st0.copy()
    sp[:, 32:35] = [0.9, 0.9, 0.65]
    o.task.max_steps = 10 ** 6
    plans = [scenarios.push_actions(o, sp[e]) for e in range(n)]
    n_app, n_push = plans[0][2], plans[0][3]
    for k in range(n_app + n_push):
        goal = [p[0] if k < n_app else p[1] for p in plans]
        a = np.array([np.append(scenarios.track(sp[e], goal[e], 1.0 if k < n_app else 0.35), 0.0)[:7] for e in range(n)])
        sp, _ = o.batch_step(sp, a)
    return st0, one, st, sp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_ext_sensitivity.json"))
    a = ap.parse_args()
    tbl, _ = panda_table()

    def make():
        o = orc.Oracle(tbl, task=1)
        o.task.obj_pose_rnd_std = 0.03
        o.task.tg_pose_rnd_std = 0.0
        return o
    base = run(make(), a.envs)
    push0 = np.linalg.norm(base[3][:, 9:11] - base[0][:, 9:11], axis=1)
    rows = []
    for name, setter, why in ALTERNATIVES:
        if setter is None:
            rows.append({"constant": name, "note": why})
            continue
        o = make()
        setter(o)
        r = run(o, a.envs)
        push = np.linalg.norm(r[3][:, 9:11] - r[0][:, 9:11], axis=1)
        rows.append({"constant": name, "stands_for": why,
                     "reset_state": quantities(r[0], base[0]), "one_step": quantities(r[1], base[1]) if np.abs(r[0] - base[0]).max() < 1e-9 else "n/a (the reset state already differs)",
                     "rollout_60_steps": quantities(r[2], base[2]),
                     "scripted_push": {"final_obj_pos_diff_m_max": float(np.linalg.norm(r[3][:, 9:11] - base[3][:, 9:11], axis=1).max()),
                                       "push_length_m_median": float(np.median(push)), "push_length_m_median_default": float(np.median(push0))}})
        print(json.dumps(rows[-1]))
    out = {"tool": "tools/ext_sensitivity.py", "engine": "fp64 oracle (oracle/pbre_oracle.c), Panda push, joint control", "envs": a.envs,
           "parity_bounds_for_scale": {"q_rad": 1.5e-6, "qd_rad_s": 1.5e-4, "obj_pos_m": 3e-7, "obj_v_m_s": 5e-5}, "rows": rows}
    json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
