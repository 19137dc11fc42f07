#!/usr/bin/env python
"""Measured per-quantity errors of the engine against the oracle (the numbers tests/parity.py: TOL is calibrated on).
    python tools/parity_report.py [--lib emu|hip] [--n 64]
`emu`: the CPU lane emulation of the device sources (no GPU needed); `hip`: libpbre.so on the GPU."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="emu")
ap.add_argument("--n", type=int, default=48)
args = ap.parse_args()

import parity
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table, PANDA_SPHERES

lib = _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so")) if args.lib == "emu" else _capi.load()
tbl, model = panda_table()
panda = {"table": tbl, "model": model, "spheres": PANDA_SPHERES}
parity.TOL = dict((k, 1e9) for k in parity.TOL)          # measure, do not assert
parity.TOL_CONTACT = dict(parity.TOL)
res = {}
n = args.n
for task in (0, 1):
    eng, ora = parity.make_pair(_capi.Engine, lib, tbl, n, task=task)
    st = parity.check_reset(eng, ora, n)
    res["single_steps_task%d" % task] = parity.check_single_steps(eng, ora, st, np.random.default_rng(0), steps=6)
_, ora = parity.make_pair(_capi.Engine, lib, tbl, 1)
base, _ = ora.batch_reset(1)
rng = np.random.default_rng(1)
S = parity.contact_states(ora, panda, base[0], rng, 24, 24)
for flags, name in ((0, "auto"), (_capi.F_COMPLEX_ROWS, "rows"), (_capi.F_COMPLEX_LANES, "lanes")):
    eng, ora = parity.make_pair(_capi.Engine, lib, tbl, len(S), flags=flags)
    res["contact_rich_" + name] = parity.check_single_steps(eng, ora, S, np.random.default_rng(1), steps=1, skip_ambiguous=True)
eng, ora = parity.make_pair(_capi.Engine, lib, tbl, 4, flags=_capi.F_COMPLEX_ROWS)
st, _ = ora.batch_reset(4)
st[0, 3] = 0.02; st[1, 5] = -0.12; st[2, 7] = 0.045; st[3, 1] = -1.9
res["joint_limit_rows"] = parity.check_single_steps(eng, ora, st, np.random.default_rng(2), steps=2)
print(json.dumps(res, indent=1, default=float))
