"""GPU diagnostic: single-step error of the engine against the oracle from the engine's own states along a free-running episode."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl, _ = panda_table()
n, steps = 24, 1000
fl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
A, ora = parity.make_pair(_capi.Engine, None, tbl, n, max_steps=steps + 10, flags=fl)
st = parity.check_reset(A, ora, n)
st[:, 32:35] = [0.6, 0.3, 0.65]
st = st.astype(np.float32).astype(np.float64)
A.set_state(st.astype(np.float32))
rng = np.random.default_rng(3)
nsaved = 0
for k in range(steps):
    a = rng.uniform(-1, 1, (n, 7)).astype(np.float32)
    s0 = A.get_state().copy()
    ncx = A.kernel_info()[5]
    A.step(a)
    st, out = ora.batch_step(st, a)
    sA = A.get_state().astype(np.float64)
    so, _ = ora.batch_step(s0.astype(np.float64), a)
    e = np.abs(sA[:, 16:25] - so[:, 16:25]).max(1)
    if e.max() > 2e-4:
        i = int(e.argmax())
        print("step", k, "env", i, "single-step qd err %.3e" % e.max(), "joint", int(np.abs(sA[i, 16:25] - so[i, 16:25]).argmax()), "complex now", ncx,
              "q", np.round(s0[i, :9], 4), flush=True)
        if nsaved < 5:
            np.savez(os.path.join(ROOT, "gpurun_out", "diverge%d.npz" % nsaved), state=s0, action=a, eng=sA, ora=so, step=k, env=i)
            nsaved += 1
    if k % 100 == 99:
        print(k, "free-running vs oracle q %.2e qd %.2e" % (np.abs(sA[:, :9] - st[:, :9]).max(), np.abs(sA[:, 16:25] - st[:, 16:25]).max()), "complex", ncx, flush=True)
