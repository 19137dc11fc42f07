"""Worker of tests/test_comm_fake_rccl.py (and of the GPU variant in tests/test_gpu_rccl.py): one rank of a `world`-rank sharded batch that
runs the CONTEXT-OWNED exchanges of include/pbre.h -- pbre_comm_init, pbre_scatter_actions_device, pbre_step_gather_device,
pbre_gather_wait -- with tests/fake_rccl as the RCCL library, so that the world > 1 branches of csrc/pbre_comm_impl.hpp execute.
Rank 0 checks the stacked rows of every step against an unsharded engine stepping the same envs: bit for bit.

    python tests/comm_worker.py <lib: emu | hip> <rank> <world> <rendezvous file> <total envs> <steps> [closed|open]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))


def main():
    kind, rank, world, rdv, total, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    mode = sys.argv[7] if len(sys.argv) > 7 else "closed"       # closed | open | closed-legacy | open-legacy (the latter two: GPU, torch's default stream)
    fake = os.path.join(ROOT, "tests", "fake_rccl", "build", "libfake_rccl.so")
    os.environ["PBRE_RCCL_LIB"] = fake
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table
    tbl, _ = panda_table()
    dev = kind == "hip"
    if dev:
        os.environ["FAKE_RCCL_DEVICE"] = "1"
        import torch
        torch.cuda.set_device(0)
        lib = _capi.load()
        if mode.endswith("-legacy"):      # torch's default stream = HIP's legacy null stream: PBRE_STREAM_LEGACY, a stream like any other for the exchanges
            stream = _capi.STREAM_LEGACY
        else:
            side = torch.cuda.Stream()
            torch.cuda.set_stream(side)
            stream = side.cuda_stream
        def buf(shape):
            return torch.zeros(shape, device="cuda", dtype=torch.float32)
        def ptr(t):
            return t.data_ptr()
        def host(t):
            torch.cuda.synchronize()
            return t.cpu().numpy()
        def put(t, a):
            t.copy_(torch.from_numpy(np.ascontiguousarray(a, np.float32)))
    else:
        lib = _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so"))
        stream = 1          # (any non-null "stream": the emulation is synchronous)
        def buf(shape):
            return np.zeros(shape, np.float32)
        def ptr(t):
            return t.ctypes.data
        def host(t):
            return t
        def put(t, a):
            t[...] = a
    n = total // world
    kw = dict(task=_capi.TASK_PUSH, seed=77, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET, max_steps=6, lib=lib)
    eng = _capi.Engine(tbl, num_envs=n, env_id_base=rank * n, **kw)
    eng.reset()
    # rendezvous: rank 0 publishes the 128-byte id in a file (any out-of-band channel will do, include/pbre.h)
    if rank == 0:
        uid = _capi.Engine.comm_unique_id(lib, fake)
        with open(rdv + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(rdv + ".tmp", rdv)
    else:
        t0 = time.time()
        while not os.path.exists(rdv):
            assert time.time() - t0 < 60, "rank 0 never published the id"
            time.sleep(0.01)
        uid = open(rdv, "rb").read()
    eng.comm_init(uid, rank, world, fake)
    info = eng.comm_info()
    assert info["ranks_seen"] == world and info["rank"] == rank and info["rccl_version_code"] == 99999, info      # the shim is what ran
    ow = eng.obs_dim + 2
    rows_local = [buf((n, ow)) for _ in range(2)]
    rows_all = [buf((total, ow)) if rank == 0 else None for _ in range(2)]
    act_local = buf((n, eng.act_dim))
    act_all = buf((total, eng.act_dim)) if rank == 0 else None
    ref = None
    if rank == 0:
        ref = _capi.Engine(tbl, num_envs=total, env_id_base=0, **kw)
        ref.reset()
    rng = np.random.default_rng(5)
    prev = None
    for k in range(steps):
        b = k & 1
        if rank == 0:
            if mode.startswith("closed") and prev is not None:
                # a policy that READS the gathered rows of the previous step: a(t + 1) = f(obs(t)) -- the closed loop
                a_all = np.tanh(3.0 * prev[:, 9:16]).astype(np.float32) * np.float32(0.7) + rng.uniform(-0.3, 0.3, (total, eng.act_dim)).astype(np.float32)
            else:
                a_all = rng.uniform(-1, 1, (total, eng.act_dim)).astype(np.float32)
            put(act_all, a_all)
        eng.scatter_actions_device(ptr(act_all) if rank == 0 else 0, ptr(act_local), stream)
        eng.step_gather_device(ptr(act_local), ptr(rows_local[b]), ptr(rows_all[b]) if rank == 0 else 0, stream)
        if mode.startswith("closed") or k == steps - 1:
            eng.gather_wait(stream, host=True)
        if rank == 0:
            ob, rw, dn = ref.step(a_all)
            want = np.concatenate([ob, rw[:, None], dn[:, None]], 1)
            if mode.startswith("closed") or k == steps - 1:
                got = np.array(host(rows_all[b]))
                assert got.shape == want.shape
                assert np.array_equal(got, want), "step %d: stacked rows differ from the unsharded engine (max %g)" % (k, np.abs(got - want).max())
                prev = got
    info = eng.comm_info()
    assert info["exchanges"] == steps and info["action_scatters"] == steps, info
    eng.close()
    if ref is not None:
        ref.close()
    print("COMM_OK rank %d of %d, %d envs, %d steps, %s loop" % (rank, world, total, steps, mode))


if __name__ == "__main__":
    main()
