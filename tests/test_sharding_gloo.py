"""N > 1 path on CPU: two processes (gloo), each owning half of the env batch through the host lane emulation,
one gather per step.  The stacked result on rank 0 must equal the single-process batch bit for bit
(envs are independent and RNG streams are keyed by global env id)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, steps, q):
    sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
    import torch.distributed as dist
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table
    from pybullet_robot_envs.sharding import ShardedEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so"))
    tbl, _ = panda_table()
    se = ShardedEngine(tbl, total, lib=lib, device_id=0, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    assert (se.env_id_base, se.n_local) == (rank * total // world, total // world)
    acts = np.random.default_rng(11).uniform(-1, 1, (steps, total, 7)).astype(np.float32)
    res = [se.reset()]
    for k in range(steps):
        r = se.step(acts[k, se.env_id_base:se.env_id_base + se.n_local])
        res.append(None if r is None else np.concatenate([r[0], r[1][:, None], r[2][:, None]], 1))
    # the asynchronous, double-buffered gather loop bench.py --gpus N times (sharding.GatherPipeline), on CPU tensors over gloo: the
    # host emulation's pbre_step_device takes host pointers, the `stream` argument is ignored
    import torch
    from pybullet_robot_envs.sharding import GatherPipeline
    pipe = GatherPipeline(se, torch.device("cpu"), gather=True, host_staged=False)
    got = []
    for k in range(steps):
        a = torch.from_numpy(np.ascontiguousarray(acts[k, se.env_id_base:se.env_id_base + se.n_local]))
        b = pipe.step(a, stream=1)
        if k >= 1:
            r = pipe.rows(b ^ 1)
            got.append(None if r is None else r.clone().numpy())
    pipe.drain()
    r = pipe.rows((steps - 1) & 1)
    got.append(None if r is None else r.clone().numpy())
    dist.barrier()
    if rank == 0:
        q.put(res + got)
    else:
        assert all(r is None for r in res) and all(g is None for g in got)
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process(panda, emu_lib):
    import torch.multiprocessing as mp
    from pybullet_robot_envs import _capi
    total, steps, world = 8, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    eng = _capi.Engine(panda["table"], num_envs=total, lib=emu_lib, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    acts = np.random.default_rng(11).uniform(-1, 1, (steps, total, 7)).astype(np.float32)
    assert np.array_equal(res[0], eng.reset())
    for k in range(steps):
        o, r, d = eng.step(acts[k])
        assert np.array_equal(res[k + 1], np.concatenate([o, r[:, None], d[:, None]], 1))
    for k in range(steps):                 # the pipelined loop continued from the same state with the same actions again
        o, r, d = eng.step(acts[k])
        assert np.array_equal(res[steps + 1 + k], np.concatenate([o, r[:, None], d[:, None]], 1))


def _hands_worker(rank, world, port, total, steps, q):
    sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import parity
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import icub_hands_table, GRASP_POS
    from pybullet_robot_envs.sharding import ShardedEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so"))
    tbl, _, info = icub_hands_table("r")
    se = ShardedEngine(tbl, total, task=_capi.TASK_REACH, lib=lib, device_id=0, robot=_capi.ROBOT_ICUB_HANDS, obj_pose_rnd_std=0.04,
                       **parity.hands_overrides(info, "r", 0))
    home = np.asarray(info["home"], np.float32)[info["controlled"]]
    acts = home[None, None, :] + np.random.default_rng(12).uniform(-0.2, 0.2, (steps, total, len(home))).astype(np.float32)
    res = [se.reset()]
    se.engine.set_motors(info["fingers"], GRASP_POS, 0.1, 10.0)          # finger command on every shard
    for k in range(steps):
        r = se.step(acts[k, se.env_id_base:se.env_id_base + se.n_local])
        res.append(None if r is None else np.concatenate([r[0], r[1][:, None], r[2][:, None]], 1))
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


def test_two_rank_gather_icub_hands(emu_lib):
    """BASELINE config 5's sharding: the iCub-with-hands batch split over two ranks equals the single-process batch bit for bit."""
    import torch.multiprocessing as mp
    import parity
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import icub_hands_table, GRASP_POS
    total, steps, world = 2, 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hands_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tbl, _, info = icub_hands_table("r")
    eng = _capi.Engine(tbl, task=_capi.TASK_REACH, num_envs=total, lib=emu_lib, robot=_capi.ROBOT_ICUB_HANDS, obj_pose_rnd_std=0.04,
                       **parity.hands_overrides(info, "r", 0))
    home = np.asarray(info["home"], np.float32)[info["controlled"]]
    acts = home[None, None, :] + np.random.default_rng(12).uniform(-0.2, 0.2, (steps, total, len(home))).astype(np.float32)
    assert np.array_equal(res[0], eng.reset())
    assert np.ptp(res[0][:, 46]) > 1e-4                                  # the two envs got different object poses
    eng.set_motors(info["fingers"], GRASP_POS, 0.1, 10.0)
    for k in range(steps):
        o, r, d = eng.step(acts[k])
        assert np.array_equal(res[k + 1], np.concatenate([o, r[:, None], d[:, None]], 1))


def _icub_worker(rank, world, port, total, steps, q):
    sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("PBRE_ICUB_LANE", "1")
    import torch.distributed as dist
    import parity
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import icub_table
    from pybullet_robot_envs.sharding import ShardedEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so"))
    tbl, _, info = icub_table("l")
    se = ShardedEngine(tbl, total, lib=lib, device_id=0, task=_capi.TASK_PUSH, robot=_capi.ROBOT_ICUB, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2,
                       **parity.icub_overrides(info, "l", 1, 0, 1))
    acts = np.random.default_rng(13).uniform(-1, 1, (steps, total, 3)).astype(np.float32)
    res = [se.reset()]
    for k in range(steps):
        r = se.step(acts[k, se.env_id_base:se.env_id_base + se.n_local])
        res.append(None if r is None else np.concatenate([r[0], r[1][:, None], r[2][:, None]], 1))
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


def test_two_rank_gather_icub_push(emu_lib):
    """The iCub push env (IK control; the lane-per-env pipeline's host build) split over two ranks equals the single-process batch bit for
    bit: per-env object / target poses are drawn from streams keyed by the global env id."""
    import torch.multiprocessing as mp
    import parity
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import icub_table
    total, steps, world = 4, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_icub_worker, args=(r, world, port, total, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tbl, _, info = icub_table("l")
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=total, lib=emu_lib, robot=_capi.ROBOT_ICUB, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2,
                       **parity.icub_overrides(info, "l", 1, 0, 1))
    acts = np.random.default_rng(13).uniform(-1, 1, (steps, total, 3)).astype(np.float32)
    assert np.array_equal(res[0], eng.reset())
    for k in range(steps):
        o, r, d = eng.step(acts[k])
        assert np.array_equal(res[k + 1], np.concatenate([o, r[:, None], d[:, None]], 1))


def test_shard_range():
    from pybullet_robot_envs.sharding import shard_range
    assert shard_range(131072, 3, 8) == (49152, 16384)
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
