"""Multi-GPU sharding of the env batch: one process per GPU (torch.distributed; backend "nccl" is RCCL over
xGMI on ROCm), env i lives on rank i // (N / G).  Envs are independent, so the data path has no collective
except the single gather per step that returns the stacked [obs | reward | done] rows to rank 0.
RNG streams are keyed by the global env id (pbre_config.env_id_base), so results are bitwise independent of G."""
import os

import numpy as np

from pybullet_robot_envs import _capi


def shard_range(total_envs, rank, world):
    if total_envs % world != 0:
        raise ValueError("total_envs (%d) must be divisible by the number of ranks (%d)" % (total_envs, world))
    n = total_envs // world
    return rank * n, n


class ShardedEngine(object):
    """The local shard of a `total_envs`-wide batch plus the per-step gather."""

    def __init__(self, robot_table, total_envs, task=_capi.TASK_PUSH, lib=None, device_id=None, **cfg):
        import torch.distributed as dist
        self.dist = dist
        self.distributed = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.distributed else 0
        self.world = dist.get_world_size() if self.distributed else 1
        self.total_envs = int(total_envs)
        base, n = shard_range(self.total_envs, self.rank, self.world)
        if device_id is None:
            device_id = int(os.environ.get("LOCAL_RANK", "0"))
        self.engine = _capi.Engine(robot_table, task=task, num_envs=n, lib=lib, device_id=device_id,
                                   env_id_base=base, **cfg)
        self.n_local, self.env_id_base = n, base
        self.obs_dim, self.act_dim = self.engine.obs_dim, self.engine.act_dim
        self._gather_list = None

    # ---- host-buffer path (numpy in, numpy out on rank 0) ----
    def _gather_np(self, local):
        import torch
        if not self.distributed:
            return local
        t = torch.from_numpy(np.ascontiguousarray(local))
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        outs = [torch.empty_like(t) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(t, outs, dst=0)
        if self.rank != 0:
            return None
        return torch.cat(outs, 0).cpu().numpy()

    def reset(self):
        return self._gather_np(self.engine.reset())

    def step(self, actions_local):
        """actions_local: [n_local, act_dim] for this rank's envs.  Returns (obs, reward, done) stacked over all ranks
        on rank 0, None elsewhere."""
        obs, rew, done = self.engine.step(actions_local)
        out = self._gather_np(np.concatenate([obs, rew[:, None], done[:, None]], 1))
        if out is None:
            return None
        return out[:, :self.obs_dim], out[:, self.obs_dim], out[:, self.obs_dim + 1]

    # ---- device-resident path (torch CUDA tensors; one RCCL gather per step) ----
    def step_device(self, actions, out, gathered=None, stream=None):
        """actions [n_local, act_dim] and out [n_local, obs_dim + 2]: CUDA float32 tensors on this rank's GPU;
        gathered: list of `world` tensors like `out` on rank 0 (None elsewhere).  Asynchronous.  The step kernels go to torch's
        current stream (stream=None) -- the stream RCCL orders the gather against -- or to the given hipStream_t handle, which then
        must be the stream torch.distributed sees as current (torch.cuda.set_stream / with torch.cuda.stream(...))."""
        if stream is None:
            stream = _capi.torch_stream(actions.device)
        self.engine.step_device(actions.data_ptr(), out.data_ptr(), stream)
        if self.distributed:
            self.dist.gather(out, gathered if self.rank == 0 else None, dst=0)


class GatherPipeline(object):
    """The per-step data path of a sharded rollout, device resident: step kernels on torch's current stream, then ONE gather of
    the [n_local, obs_dim + 2] rows to rank 0 (RCCL over xGMI with the "nccl" backend), asynchronous and double buffered -- the
    gather of step k overlaps with the kernels of step k + 1, and an output buffer is reused only after the gather that read it
    has completed.  `bench.py --gpus N` times exactly this loop; tests/test_gpu_rccl.py runs it against an unsharded engine.
      gather=False     every rank's consumer reads its own rows (no collective at all);
      host_staged=True the rows go through host memory (gloo; single-GPU test boxes where RCCL refuses two ranks on one device)."""

    def __init__(self, sharded, device, gather=True, host_staged=False):
        import torch
        self.torch, self.sh, self.dev = torch, sharded, device
        self.gather, self.host_staged = gather, host_staged
        eng = sharded.engine
        n = sharded.n_local
        self.out = [torch.zeros((n, eng.obs_dim + 2), device=device, dtype=torch.float32) for _ in range(2)]
        self.gathered = None
        if sharded.distributed and sharded.rank == 0:
            self.gathered = [[torch.zeros_like(self.out[0]) for _ in range(sharded.world)] for _ in range(2)]
        self.pending = [None, None]
        self.k = 0

    def step(self, actions, stream=None, timing_events=None):
        """enqueue one step of the local shard on `actions` [n_local, act_dim] (CUDA tensor) and its gather; returns the buffer index b:
        rank 0 finds the stacked rows of this step in self.gathered[b] after wait(b) (self.out[b] with one rank).  timing_events: a pair
        of torch events; the second is recorded between the step kernels and the gather (bench.py's kernel-only figure)"""
        sh, b = self.sh, self.k & 1
        self.wait(b)                                   # the gather that read this buffer two steps ago
        if stream is None:
            stream = _capi.torch_stream(actions.device)
        sh.engine.step_device(actions.data_ptr(), self.out[b].data_ptr(), stream)
        if timing_events is not None:
            timing_events[1].record()
        if sh.distributed and self.gather:
            if self.host_staged:
                self.torch.cuda.current_stream(self.dev).synchronize()
                h = self.out[b].cpu()
                outs = [self.torch.empty_like(h) for _ in range(sh.world)] if sh.rank == 0 else None
                sh.dist.gather(h, outs, dst=0)
                if sh.rank == 0:
                    for g, o in zip(self.gathered[b], outs):
                        g.copy_(o)
            else:
                self.pending[b] = sh.dist.gather(self.out[b], self.gathered[b] if sh.rank == 0 else None, dst=0, async_op=True)
        self.k += 1
        return b

    def wait(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def drain(self):
        self.wait(0)
        self.wait(1)

    def rows(self, b):
        """stacked [total_envs, obs_dim + 2] rows of the step that returned b (rank 0; None elsewhere)"""
        self.wait(b)
        if not self.sh.distributed or not self.gather:
            return self.out[b]
        return self.torch.cat(self.gathered[b], 0) if self.sh.rank == 0 else None


class CtxGatherPipeline(object):
    """GatherPipeline's loop with the gather owned by the engine's context (include/pbre.h: pbre_comm_init / pbre_step_gather_device;
    csrc/pbre_comm.hip): step kernels on the caller's stream, then ONE grouped RCCL exchange (ncclSend of every rank's rows, matching
    ncclRecv's on rank 0) on the context's communication stream, double buffered -- all enqueued by one C call per step, no torch
    collective and no Python between the kernels and the exchange.  torch.distributed (any backend) is used once, to hand rank 0's
    ncclUniqueId to the other ranks.  Same interface as GatherPipeline (step / wait / drain / rows / out / k)."""

    def __init__(self, sharded, device, rccl_lib=None):
        import torch
        self.torch, self.sh, self.dev = torch, sharded, device
        eng = sharded.engine
        n, w = sharded.n_local, sharded.world
        # Setup is failure-SYMMETRIC: every rank probes that the RCCL library loads (rank 0: pbre_comm_unique_id, the others: pbre_comm_probe) and the ranks agree on
        # the outcome before anything blocking -- a rank that failed alone (rank 0 before its broadcast, any rank before the collective
        # ncclCommInitRank) would leave the others waiting in a collective it never joins.  Either every rank has a communicator on
        # return or every rank raises (and the caller falls back to torch.distributed's gather on all of them).
        uid, err = None, None
        try:
            if sharded.rank == 0:
                uid = _capi.Engine.comm_unique_id(eng.lib, rccl_lib)
            else:
                _capi.Engine.comm_probe(eng.lib, rccl_lib)             # (load probe only: ncclGetUniqueId starts a bootstrap thread + socket)
        except Exception as e:
            err = repr(e)
        if not self._all_ok(err is None):
            raise RuntimeError("CtxGatherPipeline: RCCL did not load on every rank (this rank: %s)" % (err or "ok"))
        box = [uid if sharded.rank == 0 else None]
        if sharded.distributed:
            sharded.dist.broadcast_object_list(box, src=0)
        try:
            eng.comm_init(box[0], sharded.rank, w, rccl_lib)
        except Exception as e:
            err = repr(e)
        if not self._all_ok(err is None):      # (ncclCommInitRank itself is collective: it fails or succeeds on all ranks; this catches what precedes it)
            raise RuntimeError("CtxGatherPipeline: pbre_comm_init failed on some rank (this rank: %s)" % (err or "ok"))
        self.out = [torch.zeros((n, eng.obs_dim + 2), device=device, dtype=torch.float32) for _ in range(2)]
        self.all = [torch.zeros((n * w, eng.obs_dim + 2), device=device, dtype=torch.float32) for _ in range(2)] if sharded.rank == 0 else [None, None]
        self.act_local = torch.zeros((n, eng.act_dim), device=device, dtype=torch.float32)
        self.gather = True
        self.k = 0

    def _all_ok(self, ok):
        """logical AND of `ok` over the ranks (a MIN all-reduce through the process group the bootstrap uses)"""
        sh = self.sh
        if not sh.distributed:
            return bool(ok)
        t = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float32)
        if sh.dist.get_backend() == "nccl":
            t = t.to(self.dev)
        sh.dist.all_reduce(t, op=sh.dist.ReduceOp.MIN)
        return bool(float(t.cpu()[0]) > 0.5)

    def step(self, actions, stream=None, timing_events=None):
        b = self.k & 1
        if stream is None:
            stream = _capi.torch_stream(actions.device)
        eng = self.sh.engine
        if self.gather:
            eng.step_gather_device(actions.data_ptr(), self.out[b].data_ptr(), self.all[b].data_ptr() if self.all[b] is not None else 0, stream)
        else:
            # (a plain step writes out[b] without the exchange bookkeeping: an exchange that may still be reading it is waited for first)
            eng.gather_wait(stream, host=False)
            eng.step_device(actions.data_ptr(), self.out[b].data_ptr(), stream)
        if timing_events is not None:
            timing_events[1].record()
        self.k += 1
        return b

    def closed_loop_step(self, actions_all, stream=None):
        """One step of a CLOSED loop: rank 0's `actions_all` [total_envs, act_dim] (None elsewhere) are scattered to the ranks, every rank
        steps its shard, the rows are gathered, and `stream` waits for that gather -- rank 0's policy can read rows(b) of THIS step on
        `stream` to produce the next actions.  Nothing overlaps: each step needs the previous one's observations."""
        b = self.k & 1
        if stream is None:
            stream = _capi.torch_stream(self.dev)
        eng = self.sh.engine
        eng.scatter_actions_device(actions_all.data_ptr() if actions_all is not None else 0, self.act_local.data_ptr(), stream)
        eng.step_gather_device(self.act_local.data_ptr(), self.out[b].data_ptr(), self.all[b].data_ptr() if self.all[b] is not None else 0, stream)
        eng.gather_wait(stream, host=False)
        self.k += 1
        return b

    def wait(self, b=None):
        self.sh.engine.gather_wait(_capi.torch_stream(self.dev), host=False)

    def drain(self):
        self.sh.engine.gather_wait(_capi.torch_stream(self.dev), host=True)

    def rows(self, b):
        """stacked [total_envs, obs_dim + 2] rows of the step that returned b (rank 0; None elsewhere)"""
        self.wait(b)
        return self.all[b] if self.gather else self.out[b]

    def info(self):
        return self.sh.engine.comm_info()
