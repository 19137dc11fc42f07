"""RCCL for real: a `nccl` (= RCCL on ROCm) process group is initialised on the box's GPU and the sharded data path that
`bench.py --gpus N` times -- ShardedEngine + GatherPipeline: step kernels on torch's current stream, one asynchronous gather of the
[obs | reward | done] rows per step, double-buffered output rows reused only after their gather completed -- runs against an
unsharded engine stepping the same envs: the gathered rows must be bit-identical.  The boxes have one GPU, so the world size is 1
(RCCL refuses two ranks on one device); what this exercises is the RCCL communicator, the stream ordering between the engine's
kernels and the collective, and the async_op / buffer-reuse loop.  World size 2 is covered on CPU by tests/test_sharding_gloo.py.
Runs in a subprocess: the process group and RCCL's threads must not leak into the pytest process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, socket
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import torch
import torch.distributed as dist
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
from pybullet_robot_envs.sharding import ShardedEngine, GatherPipeline

s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
# a collective that must go through RCCL even with one rank
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(t)
torch.cuda.synchronize()
assert t.tolist() == list(range(8))

tbl, _ = panda_table()
n, steps = 4096, 12
kw = dict(task=_capi.TASK_PUSH, seed=77, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)                       # as bench.py: kernels and the collective are ordered on this stream
sh = ShardedEngine(tbl, n, device_id=0, **kw)
assert sh.distributed and sh.world == 1 and sh.n_local == n and sh.env_id_base == 0
ref = _capi.Engine(tbl, num_envs=n, device_id=0, **kw)
o1 = sh.reset(); o2 = ref.reset()
assert np.array_equal(o1, o2)
pipe = GatherPipeline(sh, dev, gather=True, host_staged=False)
gen = torch.Generator(device=dev); gen.manual_seed(5)
acts = torch.rand((steps, n, sh.act_dim), device=dev, generator=gen) * 2 - 1
got = []
for k in range(steps):
    b = pipe.step(acts[k], side.cuda_stream)
    if k >= 1:                                    # read step k-1's rows while step k is in flight (the overlap bench.py relies on)
        got.append(pipe.rows(b ^ 1).clone())
pipe.drain()
got.append(pipe.rows((steps - 1) & 1).clone())
torch.cuda.synchronize()
a_host = acts.cpu().numpy()
for k in range(steps):
    ob, rw, dn = ref.step(a_host[k])
    want = np.concatenate([ob, rw[:, None], dn[:, None]], 1)
    g = got[k].cpu().numpy()
    assert g.shape == want.shape
    assert np.array_equal(g, want), "step %d: gathered rows differ from the unsharded engine (max %g)" % (k, np.abs(g - want).max())
# ShardedEngine.step_device (synchronous gather) on top
out = torch.zeros((n, sh.obs_dim + 2), device=dev)
gl = [torch.zeros_like(out)]
sh.step_device(acts[0], out, gl, side.cuda_stream)
torch.cuda.synchronize()
ob, rw, dn = ref.step(a_host[0])
assert np.array_equal(gl[0].cpu().numpy(), np.concatenate([ob, rw[:, None], dn[:, None]], 1))
ver = ".".join(str(v) for v in torch.cuda.nccl.version())
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK version", ver, "steps", steps, "envs", n)
'''


@pytest.mark.gpu
def test_rccl_gather_pipeline_world_size_1(hip_lib):
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT], capture_output=True, text=True, timeout=600, env=env)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, tail
    print(r.stdout.strip().splitlines()[-1])


CTX_WORKER = r'''
import os, sys, socket
import numpy as np
ROOT, self_p2p, with_torch_nccl = sys.argv[1], sys.argv[2], sys.argv[3]
os.environ["PBRE_COMM_SELF_P2P"] = self_p2p
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
from pybullet_robot_envs.sharding import ShardedEngine, GatherPipeline, CtxGatherPipeline

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if with_torch_nccl == "1":      # torch.distributed's RCCL and the context's communicator in ONE process (what bench.py --gpus N has)
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
tbl, _ = panda_table()
n, steps = 4096, 12
kw = dict(task=_capi.TASK_PUSH, seed=77, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
side = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(side)
sh = ShardedEngine(tbl, n, device_id=0, **kw)
ref = _capi.Engine(tbl, num_envs=n, device_id=0, **kw)
sh.engine.reset(); ref.reset()
pipe = CtxGatherPipeline(sh, dev)
info = pipe.info()
assert info["ranks_seen"] == 1 and info["rank"] == 0 and info["rccl_version_code"] > 0, info
gen = torch.Generator(device=dev); gen.manual_seed(5)
acts = torch.rand((steps, n, sh.act_dim), device=dev, generator=gen) * 2 - 1
got = []
for k in range(steps):
    b = pipe.step(acts[k], side.cuda_stream)
    if k >= 1:                                    # step k-1's stacked rows, read while step k is in flight
        got.append(pipe.rows(b ^ 1).clone())
pipe.drain()
got.append(pipe.rows((steps - 1) & 1).clone())
torch.cuda.synchronize()
a_host = acts.cpu().numpy()
for k in range(steps):
    ob, rw, dn = ref.step(a_host[k])
    want = np.concatenate([ob, rw[:, None], dn[:, None]], 1)
    assert np.array_equal(got[k].cpu().numpy(), want), "step %d: the context's gather differs from the unsharded engine" % k
assert pipe.info()["exchanges"] == steps
if with_torch_nccl == "1":
    dist.barrier(); dist.destroy_process_group()
print("CTX_RCCL_OK", pipe.info(), "self_p2p", self_p2p)
'''


@pytest.mark.gpu
@pytest.mark.parametrize("self_p2p,with_torch_nccl", [("0", "0"), ("1", "0"), ("1", "1")])
def test_context_owned_rccl_gather_world_size_1(hip_lib, self_p2p, with_torch_nccl):
    """pbre_comm_init / pbre_step_gather_device (csrc/pbre_comm.hip): the communicator is created by the C-ABI (ncclCommInitRank through
    the dlopen'ed RCCL), the step + its exchange are enqueued by ONE C call, double buffered, and the stacked rows equal the unsharded
    engine's bit for bit.  self_p2p = 1: rank 0's own rows go through ncclSend / ncclRecv (the point-to-point kernels really run on the
    box's one GPU); with_torch_nccl = 1: torch.distributed's "nccl" group lives in the same process (one shared librccl)."""
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-c", CTX_WORKER, ROOT, self_p2p, with_torch_nccl], capture_output=True, text=True, timeout=600, env=env)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0 and "CTX_RCCL_OK" in r.stdout, tail
    print(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["closed", "open", "open-legacy"])
def test_context_owned_exchanges_two_ranks_on_one_gpu(hip_lib, mode):
    """world = 2 of csrc/pbre_comm.hip on the GPU: two processes, both on device 0, run pbre_comm_init -> pbre_scatter_actions_device ->
    pbre_step_gather_device -> pbre_gather_wait with tests/fake_rccl as PBRE_RCCL_LIB (device buffers staged through shared memory; the real
    RCCL refuses two ranks on one device).  What executes is the product's grouped ncclRecv x world / ncclSend branch, its event
    ordering between the step stream and the communication stream, and the per-buffer reuse protection; rank 0 finds the stacked rows
    bit-identical to an unsharded engine's, in a closed loop (actions computed from the gathered observations) and in the overlapped one."""
    import test_comm_fake_rccl
    test_comm_fake_rccl.run_world("hip", 2, 2048, 10, mode, timeout=600)
