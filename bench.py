#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched Panda-push step() hot path on MI355X.

Workload (BASELINE.json `metric`): pandaPushGymEnv, joint control,
131072 envs in total sharded evenly over the N GPUs of one node (strong scaling: total fixed),
obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, 150 PGS iterations, dt=1/240, actions U(-1,1).
A "step" = one batched env.step() of every env: pbre_step_device (actions and outputs resident
in HBM) and, for N>1, one RCCL gather of the stacked [obs|reward|done] rows to rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs TOTAL]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALG_BYTES_PER_ENV_STEP = 444.0      # read 172 B + write 272 B (Panda push, joint control)
# FLOPs actually required per env-step by the sparse formulation the fast kernel uses: per PGS iteration
# 9 motor rows x 23 + 4 normal rows x 18 + 8 friction rows x 20 = 439 flop, x150, + ~8 k for kinematics/dynamics/obs.
# (SURVEY 8(d)'s 0.4 MFLOP assumed 33 dense rows of 70 flop; the dense figure is what the general row kernel does.)
ALG_FLOP_PER_ENV_STEP = 150 * 439 + 8000.0
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector


def _cpu_worker(args):
    n, steps, seed = args
    import numpy as np
    import orc
    from pybullet_robot_envs.model.table import panda_table
    tbl, _ = panda_table()
    o = orc.Oracle(tbl, task=1)
    o.task.obj_pose_rnd_std = 0.05
    o.task.tg_pose_rnd_std = 0.2
    # one settled state replicated with the per-env object/target randomisation applied analytically would not
    # be the oracle's own reset; resetting n envs costs 201 steps each, so reset a few and tile them.
    st0, _ = o.batch_reset(min(n, 8), env_id0=seed * 1000)
    st = np.tile(st0, (n // st0.shape[0] + 1, 1))[:n].copy()
    rng = np.random.default_rng(seed)
    acts = rng.uniform(-1, 1, (steps, n, 7))
    t0 = time.perf_counter()
    for k in range(steps):
        st, _ = o.batch_step(st, acts[k])
    return n * steps, time.perf_counter() - t0


def cpu_baseline(target_cpu_seconds=20.0):
    """Oracle (CPU port of the same step, double precision) on this box's host cores, bounded sample."""
    import concurrent.futures as cf
    cores = len(os.sched_getaffinity(0))
    # calibrate: ~1e-4 s per env-step per core
    n, steps = 256, 8
    work, dt = _cpu_worker((n, steps, 0))
    per = dt / work
    total = max(cores * n * steps, int(target_cpu_seconds / per))
    steps = 16
    n = max(16, total // (cores * steps))
    t0 = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(_cpu_worker, [(n, steps, i + 1) for i in range(cores)]))
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": done / busy, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps per core on %d cores (oracle/pbre_oracle.c, fp64, 150 PGS iters), "
                      "%.1f s wall incl. process start" % (n, steps, cores, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=131072, help="total envs over all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steady-preroll", type=int, default=300,
                    help="untimed steps before the extra mid-episode measurement (0 disables it)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    assert args.envs % world == 0
    n_local = args.envs // world
    tbl, _ = panda_table()
    from pybullet_robot_envs.sharding import ShardedEngine
    sh = ShardedEngine(tbl, args.envs, task=_capi.TASK_PUSH, device_id=local_rank, seed=1234,
                       obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
    eng = sh.engine
    assert (sh.n_local, sh.env_id_base) == (n_local, rank * n_local)
    ow = eng.obs_dim + 2
    eng.reset()

    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    pool = [torch.rand((n_local, eng.act_dim), device=dev, generator=gen) * 2 - 1 for _ in range(8)]
    out = torch.zeros((n_local, ow), device=dev, dtype=torch.float32)
    gathered = [torch.zeros_like(out) for _ in range(world)] if (world > 1 and rank == 0) else None
    # a dedicated non-null stream: the kernel, the HIP timing events and the RCCL gather all run on it
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0

    def one_step(k, ev=None):
        a = pool[k % len(pool)]
        if ev is not None:
            ev[0].record()
        eng.step_device(a.data_ptr(), out.data_ptr(), stream)
        if ev is not None:
            ev[1].record()
        if world > 1:
            dist.gather(out, gathered, dst=0)       # the one collective of the data path (RCCL over xGMI)

    for k in range(args.warmup):
        one_step(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k, evs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))     # whole launch pair (fork/join incl.), HIP events on the caller's stream
    t = torch.tensor([elapsed, kern_ms], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, pair_ms = float(t[0]), float(t[1])
    finite = bool(torch.isfinite(out).all())
    complex_after = eng.kernel_info()[5]
    # duration of the dominant kernel (k_fast) alone: mean over the last <= 64 timed steps of the HIP event pairs the library
    # records around that kernel on the stream it is launched on (pbre_timing[3])
    kern_ms = float(eng.timing()[3])

    # extra, reported separately: the same K steps measured mid-episode (after an untimed pre-roll), when a few per cent
    # of the envs have robot contacts / joints at a limit and take the heavier k_fast_rc kernel
    steady = None
    if args.steady_preroll > 0:
        for k in range(args.steady_preroll):
            one_step(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.steps):
            one_step(k)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        e2 = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(e2, op=dist.ReduceOp.MAX)
        steady = {"preroll_steps": args.steady_preroll, "value": args.envs * args.steps / float(e2[0]), "unit": "env-steps/s",
                  "ms_per_step": float(e2[0]) / args.steps * 1e3, "complex_env_frac_rank0": eng.kernel_info()[5] / n_local}

    if rank == 0:
        value = args.envs * args.steps / elapsed
        kern_s = kern_ms * 1e-3
        ach_gbs = ALG_BYTES_PER_ENV_STEP * n_local / kern_s / 1e9
        ach_tf = ALG_FLOP_PER_ENV_STEP * n_local / kern_s / 1e12
        info = eng.kernel_info()
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, corrected as
        # calibrated in profiles/r01_pmc_hbm.json); counters cannot be read from inside this process, so the committed
        # per-env figure of the profiled run is scaled to this launch's env count.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")))["k_fast<7>"]
            traffic = pmc["hbm_bytes_per_env_step"] * n_local
        except Exception:
            pass
        res = {
            "metric": "env-steps/sec (whole node), Panda-push 128k envs",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "pandaPushGymEnv joint-control step, %d envs total (%d per GPU), 150 PGS iters, "
                                   "dt 1/240, obj_pose_rnd_std 0.05, tg_pose_rnd_std 0.2, actions U(-1,1) resident in HBM"
                                   % (args.envs, n_local),
                       "envs_total": args.envs, "envs_per_gpu": n_local, "parallelism": "dp%d" % world,
                       "collective": "rccl gather to rank 0 per step" if world > 1 else "none",
                       "outputs_finite": finite, "start_state": "fresh reset() of every env (reference reset_simulation)",
                       "complex_env_frac_after_timed_steps_rank0": complex_after / n_local},
            "steady_state": steady,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_fast<7>", "kernel_ms": kern_ms, "step_launch_pair_ms": pair_ms, "algorithmic_bytes_per_launch": ALG_BYTES_PER_ENV_STEP * n_local,
                         "note": "path is fp32-VALU/dependency bound (AI ~900 FLOP/B >> 25 FLOP/B machine balance); "
                                 "HBM fraction is small by construction, see valu"},
            "valu": {"achieved": ach_tf, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP32_VALU_PEAK_TFLOPS,
                     "flop_per_env_step": ALG_FLOP_PER_ENV_STEP, "vgprs_k_fast": info[0], "vgprs_k_fast_rc": info[6], "vgprs_k_step": info[1]},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is informative; never lose the GPU line over it
                res["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
