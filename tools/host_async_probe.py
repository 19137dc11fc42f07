"""Where does a step of the pipelined host path (Engine.step_async / step_wait) spend its time?  Host-side phases (copy of the actions into
the page-locked slot, the enqueueing C call, the wait) and, from torch, the PCIe rates of this box for the two transfer sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table

dev = torch.device("cuda", 0)
n = 131072
for mb, name in (((n * 7 * 4, "actions"), (n * 35 * 4, "rows")) if os.environ.get("PROBE_TORCH", "1") == "1" else ()):
    h = torch.empty(mb // 4, dtype=torch.float32).pin_memory()
    d = torch.empty(mb // 4, dtype=torch.float32, device=dev)
    for direction in ("h2d", "d2h"):
        for _ in range(3):
            (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            (d.copy_(h, non_blocking=True) if direction == "h2d" else h.copy_(d, non_blocking=True))
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 20
        print("torch %s %s: %.3f ms = %.1f GB/s" % (name, direction, el * 1e3, mb / el / 1e9))
tbl, _ = panda_table()
eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
eng.reset()
rng = np.random.default_rng(0)
NA = 61      # (a pool this size, walked with a stride: i.i.d. enough for a few hundred steps -- a pool of 4 drives half the batch into joint limits)
acts = [rng.uniform(-1, 1, (n, 7)).astype(np.float32) for _ in range(NA)]
tw = []
for k in range(int(os.environ.get("PROBE_WARM", "300"))):
    t0 = time.perf_counter()
    eng.step(acts[(7 * k) % NA], copy=False)
    tw.append(time.perf_counter() - t0)
if tw:
    print("synchronous Engine.step during the warm-up, ms: first 8", [round(x * 1e3, 2) for x in tw[:8]], " steps 50..57", [round(x * 1e3, 2) for x in tw[50:58]], " last 8", [round(x * 1e3, 2) for x in tw[-8:]],
          " timing() h2d / kernels / d2h", [round(x, 3) for x in eng.timing()[:3]], " complex envs", eng.kernel_info()[5])
K = int(os.environ.get("PROBE_STEPS", "60"))
eng.step_async(acts[0])
for k in range(1, 6):
    eng.step_async(acts[(7 * k) % NA]); eng.step_wait()
ph = np.zeros(3)
t_all = time.perf_counter()
for k in range(K):
    A = eng._async
    t0 = time.perf_counter()
    slot = A["issued"] % 3
    np.copyto(A["act"][slot], acts[(7 * k) % NA])
    t1 = time.perf_counter()
    eng._chk(eng.lib.pbre_step_async(eng._ctx, _capi._fp(A["act"][slot]), _capi._fp(A["out"][slot]))); A["issued"] += 1
    t2 = time.perf_counter()
    eng.step_wait()
    t3 = time.perf_counter()
    ph += [t1 - t0, t2 - t1, t3 - t2]
el = (time.perf_counter() - t_all) / K
eng.step_wait()
print("pipelined: %.3f ms per step; host phases (ms): copy actions %.3f, enqueue %.3f, wait %.3f   [PBRE_ASYNC_D2H=%s]" % (el * 1e3, *(ph / K * 1e3), os.environ.get("PBRE_ASYNC_D2H", "0")))
# without the host copy of the actions (the caller writes them into the page-locked slot itself)
t_all = time.perf_counter()
for k in range(K):
    A = eng._async
    slot = A["issued"] % 3
    eng._chk(eng.lib.pbre_step_async(eng._ctx, _capi._fp(A["act"][slot]), _capi._fp(A["out"][slot]))); A["issued"] += 1
    eng.step_wait()
el = (time.perf_counter() - t_all) / K
print("pipelined, actions already in the slot: %.3f ms per step" % (el * 1e3))
t_all = time.perf_counter()
for k in range(K):
    eng.step(acts[(7 * k) % NA], copy=False)
print("synchronous Engine.step: %.3f ms per step" % ((time.perf_counter() - t_all) / K * 1e3))
# Engine.step_pipelined: copy the actions into the slot, THEN wait for the step before last, THEN enqueue (the host's copy is off the critical path)
for k in range(4):
    eng.step_pipelined(acts[(7 * k) % NA])
tp = []
t_all = time.perf_counter()
for k in range(K):
    t0 = time.perf_counter()
    eng.step_pipelined(acts[(7 * k + 3) % NA])
    tp.append(time.perf_counter() - t0)
el = (time.perf_counter() - t_all) / K
eng.step_wait(); eng.step_wait()
print("Engine.step_pipelined: %.3f ms per step  (per call, ms: first 6 %s, last 6 %s)  [PBRE_ASYNC_D2H=%s]" % (el * 1e3, [round(x * 1e3, 2) for x in tp[:6]], [round(x * 1e3, 2) for x in tp[-6:]], os.environ.get("PBRE_ASYNC_D2H", "0")))
