"""The cross-block spin-wait paths under contention.  Three hot paths wait, inside a kernel, for data another wave of the
same or of another kernel publishes: k_fused<., false>'s row waves for the object wave's global side record (csrc/pbre_panda.hpp:
PBRE_OBJV_SYNC), its tail pairs' robot waves for their object waves' records (PBRE_PAIR_SYNC; the first test covers both) and the iCub pipeline's quads for kw_lane_ik's per-env marks (csrc/pbre_lane.hip: PBRE_IK_WAIT).  Both rest on "the producer
waits for nothing and is dispatched first"; both are bounded, and a wait that runs out poisons the env-step, which the NaN / Inf guard counts
(pbre_kernel_info[12]).  Here the engine under test steps while a SECOND context keeps the GPU busy from another stream -- a 131072-env Panda
batch, one launch of 2300 one-wave blocks after the other: every wave slot of the chip is contended for -- and must (a) never reach a bound
(guard counter 0) and (b) produce bit for bit the rows it produces alone."""
import numpy as np
import pytest

from pybullet_robot_envs import _capi

pytestmark = pytest.mark.gpu


def _busy_engine(panda, hip_lib, torch, dev):
    """the competing context: 131072 Panda-push envs in their stationary mix, and a closure that enqueues `k` steps on its own stream"""
    n = 131072
    eng = _capi.Engine(panda["table"], task=1, num_envs=n, lib=hip_lib, seed=77, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
    eng.reset()
    stream = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    act = torch.rand((4, n, eng.act_dim), device=dev, generator=gen) * 2 - 1
    out = torch.zeros((n, eng.obs_dim + 2), device=dev)

    def enqueue(k):
        for i in range(k):
            eng.step_device(act[i & 3].data_ptr(), out.data_ptr(), stream.cuda_stream)
    return eng, enqueue, stream


def _run(eng, torch, dev, steps, acts, stream, desync_state=None):
    if desync_state is not None:
        eng.set_state(desync_state)
    out = torch.zeros((steps, eng.num_envs, eng.obs_dim + 2), device=dev)
    torch.cuda.synchronize()          # (the fill runs on torch's current stream, the steps on `stream`: under contention the fill came LAST and wiped step 0's rows)
    for k in range(steps):
        eng.step_device(acts[k].data_ptr(), out[k].data_ptr(), stream.cuda_stream)
    return out


def test_one_launch_panda_step_beside_a_second_context(panda, hip_lib, monkeypatch):
    """k_fused<7, false> (131072 envs: 64-thread blocks, the row waves spin on the object waves' global side records) beside a second
    131072-env context: rows bit-identical to the solo run, no bounded wait ran out"""
    import torch
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("PBRE_FUSED", "1")
    # (tail pairs in every launch: by the hint they start once the device has REPORTED complex envs -- and these 320 launches are all enqueued before the first has run)
    monkeypatch.setenv("PBRE_TAIL_PAIR", "236")
    n, warm, steps = 131072, 260, 60
    gen = torch.Generator(device=dev); gen.manual_seed(11)
    acts = torch.rand((warm + steps, n, 7), device=dev, generator=gen) * 2 - 1
    s_a = torch.cuda.Stream(device=dev)
    kw = dict(task=1, num_envs=n, lib=hip_lib, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)

    def fresh():
        e = _capi.Engine(panda["table"], **kw)
        e.reset()
        st = e.get_state()
        st[:, e.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)      # episodes of every age: envs finish in every step
        e.set_state(st)
        return e
    solo = fresh()
    ref = _run(solo, torch, dev, warm + steps, acts, s_a)
    torch.cuda.synchronize()
    assert solo.kernel_info()[13] >= warm + steps and solo.kernel_info()[10] == 0, "the solo run was not the 64-thread one-launch step"
    assert solo.kernel_info()[5] > 0, "no complex env: the row waves had nothing to wait for"
    assert solo.kernel_info()[15] > 0, "no step had tail pairs: the robot waves' wait for their object waves' global records was not exercised"
    assert solo.kernel_info()[12] == 0
    busy, enqueue, s_b = _busy_engine(panda, hip_lib, torch, dev)
    test = fresh()
    enqueue(400)                                      # ~60 ms of launches ahead of the engine under test ...
    got = _run(test, torch, dev, warm + steps, acts, s_a)
    enqueue(400)                                      # ... and more behind it, so that it never runs alone
    torch.cuda.synchronize()
    assert test.kernel_info()[12] == 0 and busy.kernel_info()[12] == 0, (test.kernel_info()[12], busy.kernel_info()[12])
    diff = (ref != got).any(dim=2)                      # [steps, envs]
    first = int(diff.any(dim=1).float().argmax()) if bool(diff.any()) else -1
    assert not bool(diff.any()), "rows differ beside a second context: first at step %d, %d envs there, %d env-steps in all" % (first, int(diff[first].sum()), int(diff.sum()))
    assert np.array_equal(solo.get_state(), test.get_state())
    for e in (solo, test, busy):
        e.close()


def test_icub_ik_hand_over_beside_a_second_context(panda, hip_lib, monkeypatch):
    """the iCub pipeline under Cartesian control with the per-env IK hand-over (kw_lane_ik publishes, the quads wait for their env's mark)
    beside a 131072-env Panda context: bit-identical to the solo run, guard counter 0"""
    import torch
    import parity
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("PBRE_ICUB_LANE", "1")
    n, steps = 16384, 120
    s_a = torch.cuda.Stream(device=dev)

    def fresh():
        e, _, _ = parity.make_icub_pair(_capi.Engine, hip_lib, n, task=1, use_ik=1, obj_std=0.05, tg_std=0.2, max_steps=60, flags=_capi.F_AUTO_RESET)
        e.reset()
        return e
    solo = fresh()
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    acts = torch.rand((steps, n, solo.act_dim), device=dev, generator=gen) * 2 - 1
    ref = _run(solo, torch, dev, steps, acts, s_a)
    torch.cuda.synchronize()
    assert solo.kernel_info()[12] == 0 and solo.kernel_info()[2] == 1, solo.kernel_info()
    busy, enqueue, s_b = _busy_engine(panda, hip_lib, torch, dev)
    test = fresh()
    enqueue(600)
    got = _run(test, torch, dev, steps, acts, s_a)
    enqueue(300)
    torch.cuda.synchronize()
    assert test.kernel_info()[12] == 0 and busy.kernel_info()[12] == 0, (test.kernel_info()[12], busy.kernel_info()[12])
    diff = (ref != got).any(dim=2)                      # [steps, envs]
    first = int(diff.any(dim=1).float().argmax()) if bool(diff.any()) else -1
    assert not bool(diff.any()), "rows differ beside a second context: first at step %d, %d envs there, %d env-steps in all" % (first, int(diff[first].sum()), int(diff.sum()))
    assert np.array_equal(solo.get_state(), test.get_state())
    for e in (solo, test, busy):
        e.close()
