#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched Panda-push step() hot path on MI355X.

Workload (BASELINE.json `metric`): pandaPushGymEnv, joint control,
obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, 150 PGS iterations, dt=1/240, actions U(-1,1).
N=1 runs the 131072-env batch the metric is quoted on.  Envs are independent, so the batch shards
with no data dependence between ranks: for N>1 every GPU keeps a 131072-env shard (weak scaling,
131072*N envs in total) and, per step, one RCCL gather returns the stacked [obs|reward|done] rows
to rank 0, overlapped with the next step's kernels (double-buffered output).  The literal config 4
(131072 envs in TOTAL over N GPUs, same gather) is measured in the same run and reported under
"strong_scaling_128k_total".
A "step" = one batched env.step() of every env: pbre_step_device, actions and outputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs PER_GPU]
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


def other_configs(torch, dev, steps=10):
    """Extra, informative lines for the other BASELINE configs on one GPU (never the headline value): the batched
    iCubReach-v0 (config 1's env, 32768 envs) and the iCub with hands (config 5 stand-in: 60 simulated DoF, the palm pressing
    on the object with closing fingers, 8192 envs = 65536 / 8)."""
    import math as m
    out = {}

    def timed(eng, acts):
        o = torch.zeros((eng.num_envs, eng.obs_dim + 2), device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        for k in range(2):
            eng.step_device(acts[k % len(acts)].data_ptr(), o.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            eng.step_device(acts[k % len(acts)].data_ptr(), o.data_ptr(), s)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return {"envs": eng.num_envs, "value": eng.num_envs * steps / el, "unit": "env-steps/s", "ms_per_step": el / steps * 1e3,
                "outputs_finite": bool(torch.isfinite(o).all())}, o
    try:
        from pybullet_robot_envs.envs import iCubReachGymEnv
        env = iCubReachGymEnv(use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, num_envs=32768)   # iCubReach-v0 kwargs
        env.reset()
        r, _ = timed(env._engine, [torch.rand((32768, 3), device=dev) * 2 - 1 for _ in range(4)])
        r["workload"] = "iCubReach-v0 (IK position control of the left hand), one env per half-wave"
        out["icub_reach"] = r
        env.close()
    except Exception as e:
        out["icub_reach"] = {"error": repr(e)}
    try:
        from pybullet_robot_envs import _client
        from pybullet_robot_envs.envs.icub_envs.icub_env_with_hands import iCubHandsEnv
        cid = _client.connect(8192)
        robot = iCubHandsEnv(cid, use_IK=1, control_arm='r')
        q1 = [0.0, 0.0, m.sin(m.pi / 4), m.cos(m.pi / 4)]           # yaw pi/2: palm down above the object
        robot.pre_grasp(); robot.step_simulation(10)
        robot.apply_action([0.5, -0.03, 0.72] + q1); robot.pre_grasp(); robot.step_simulation(60)
        robot.apply_action([0.5, -0.03, 0.69] + q1); robot.pre_grasp(); robot.step_simulation(40)
        robot.grasp(); robot.step_simulation(20)
        base = torch.tensor([0.5, -0.03, 0.69, 0.0, 0.0, m.pi / 2], dtype=torch.float32, device=dev)
        jit = torch.tensor([0.004, 0.004, 0.002, 0.01, 0.01, 0.01], device=dev)
        r, o = timed(robot._engine, [base + (torch.rand((8192, 6), device=dev) - 0.5) * jit for _ in range(4)])
        r["workload"] = ("iCubHandsEnv (60 simulated DoF, one env per wavefront): palm pressing on the object, fingers closing "
                         "(grasp, force 10), IK hand-pose control")
        r["mean_robot_object_contact_points"] = float(o[:, -3].mean())
        out["icub_hands_config5_standin"] = r
        _client.disconnect(cid)
    except Exception as e:
        out["icub_hands_config5_standin"] = {"error": repr(e)}
    return out


EV_EVERY = int(os.environ.get("PBRE_BENCH_EV_EVERY", "8"))     # torch event pairs around every EV_EVERY-th step of the timed region


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=131072, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steady-preroll", type=int, default=300,
                    help="untimed steps before the extra mid-episode measurement (0 disables it)")
    ap.add_argument("--no-strong", action="store_true", help="skip the extra fixed-total measurement at N>1")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the informative iCub / iCub-with-hands lines (N=1 only)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table
    from pybullet_robot_envs.sharding import ShardedEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # test hooks (single-GPU box): PBRE_BENCH_ONE_DEVICE=1 puts every rank on GPU 0, PBRE_BENCH_BACKEND=gloo stages the
    # gather through host memory.  The driver's runs use neither.
    if os.environ.get("PBRE_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("PBRE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    tbl, _ = panda_table()
    # a dedicated non-null stream: the kernels and the HIP timing events run on it; RCCL orders itself after it
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        t = torch.tensor(x, device=dev, dtype=torch.float64)
        if world > 1:
            if backend == "nccl":
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            else:
                tc = t.cpu()
                dist.all_reduce(tc, op=dist.ReduceOp.MAX)
                t = tc
        return [float(v) for v in t]

    class Job(object):
        """One sharded batch: engine, resident action pool, double-buffered output rows, pipelined gather."""

        def __init__(self, total_envs, n_actions):
            self.sh = ShardedEngine(tbl, total_envs, task=_capi.TASK_PUSH, device_id=local_rank, seed=1234,
                                    obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2)
            self.eng = eng = self.sh.engine
            self.n_local = n = self.sh.n_local
            assert self.sh.env_id_base == rank * n
            eng.reset()
            # actions U(-1,1), i.i.d. per (step, env), generated up front and resident in HBM: one slice per
            # step of the run (a short recycled pool would add a constant bias to every env's random walk and drive the
            # joints into their limits within a few hundred steps)
            gen = torch.Generator(device=dev)
            gen.manual_seed(1234 + rank)
            self.pool = torch.rand((n_actions, n, eng.act_dim), device=dev, generator=gen) * 2 - 1
            self.out = [torch.zeros((n, eng.obs_dim + 2), device=dev, dtype=torch.float32) for _ in range(2)]
            self.gathered = [torch.zeros_like(self.out[0]) for _ in range(world)] if (world > 1 and rank == 0) else None
            self.pending = [None, None]
            self.k = 0

        gather = True

        def step(self, ev=None):
            b = self.k & 1
            if self.pending[b] is not None:            # the gather that read this buffer two steps ago
                self.pending[b].wait()
                self.pending[b] = None
            if ev is not None:
                ev[0].record()
            self.eng.step_device(self.pool[self.k % self.pool.shape[0]].data_ptr(), self.out[b].data_ptr(), stream)
            if ev is not None:
                ev[1].record()
            if world > 1 and self.gather:              # the one collective of the data path (RCCL over xGMI)
                if backend == "nccl":
                    self.pending[b] = dist.gather(self.out[b], self.gathered, dst=0, async_op=True)
                else:
                    side.synchronize()
                    h = self.out[b].cpu()
                    dist.gather(h, [torch.empty_like(h) for _ in range(world)] if rank == 0 else None, dst=0)
            self.k += 1

        def drain(self):
            for b in range(2):
                if self.pending[b] is not None:
                    self.pending[b].wait()
                    self.pending[b] = None

        def timed(self, steps, events=False):
            barrier()
            # event pairs around every EV_EVERY-th (8th) step only: an event record is a barrier packet on the stream, and a pair per step
            # costs ~10% of the 0.2 ms kernel it brackets
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if events else None
            t0 = time.perf_counter()
            for k in range(steps):
                self.step(evs[k] if (events and k % EV_EVERY == 0) else None)
            self.drain()
            barrier()
            elapsed = time.perf_counter() - t0
            pair = float(np.mean([a.elapsed_time(b) for a, b in evs[::EV_EVERY]])) if events else 0.0
            return max_over_ranks([elapsed, pair])

    job = Job(args.envs * world, args.warmup + 3 * args.steps + args.steady_preroll)
    eng, n_local, total = job.eng, job.n_local, args.envs * world
    for k in range(args.warmup):
        job.step()
    elapsed, pair_ms = job.timed(args.steps, events=True)   # pair_ms: whole launch pair (fork/join incl.), HIP events on the caller's stream
    finite = bool(torch.isfinite(job.out[0]).all() and torch.isfinite(job.out[1]).all())
    complex_after = eng.kernel_info()[5]
    # duration of the dominant kernel (k_fast) alone: mean over the last <= 64 timed steps of the HIP event pairs the library
    # records around that kernel on the stream it is launched on (pbre_timing[3])
    kern_ms = float(eng.timing()[3])

    # extra, reported separately: the same K steps measured mid-episode (after an untimed pre-roll), when a few per cent
    # of the envs have robot contacts / joints at a limit and take the heavier k_fast_rc kernel
    steady = None
    if args.steady_preroll > 0:
        for k in range(args.steady_preroll):
            job.step()
        e2 = job.timed(args.steps)[0]
        steady = {"preroll_steps": args.steady_preroll, "value": total * args.steps / e2, "unit": "env-steps/s",
                  "ms_per_step": e2 / args.steps * 1e3, "complex_env_frac_rank0": eng.kernel_info()[5] / n_local}
    info = eng.kernel_info()
    # extra at N>1: the same steps without the gather (every rank's consumer reads its own shard's rows): what the engine
    # itself scales to when the stacked rows are not funnelled into one GPU (7 x 18.4 MB per step into rank 0 otherwise)
    no_gather = None
    if world > 1:
        try:
            job.gather = False
            e4 = job.timed(args.steps)[0]
            no_gather = {"value": total * args.steps / e4, "unit": "env-steps/s", "ms_per_step": e4 / args.steps * 1e3}
        except Exception as e:
            no_gather = {"error": repr(e)}
    del job

    # extra at N>1: BASELINE config 4 taken literally -- 131072 envs in TOTAL over the N GPUs (strong scaling), same gather
    strong = None
    if world > 1 and not args.no_strong:
        try:
            j2 = Job(131072, args.warmup + args.steps)
            for k in range(args.warmup):
                j2.step()
            e3 = j2.timed(args.steps)[0]
            strong = {"envs_total": 131072, "envs_per_gpu": j2.n_local, "value": 131072 * args.steps / e3, "unit": "env-steps/s",
                      "ms_per_step": e3 / args.steps * 1e3}
            del j2
        except Exception as e:   # informative only
            strong = {"error": repr(e)}

    if rank == 0:
        value = total * args.steps / elapsed
        kern_s = kern_ms * 1e-3
        ach_gbs = ALG_BYTES_PER_ENV_STEP * n_local / kern_s / 1e9
        ach_tf = ALG_FLOP_PER_ENV_STEP * n_local / kern_s / 1e12
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, corrected as
        # calibrated in profiles/r01_pmc_hbm.json); counters cannot be read from inside this process, so the committed
        # per-env figure of the profiled run is scaled to this launch's env count.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")))["k_fast<7>"]
            traffic = pmc["hbm_bytes_per_env_step"] * n_local
        except Exception:
            pass
        sq = None
        try:   # SQ issue counters of the same kernel (rocprofv3 --pmc pass, tools/pmc_sq.sh), committed summary
            d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_sq.json")))["k_fast<7>"]
            sq = {"valu_insts_per_wave": d["valu_insts_per_wave"], "valu_active_over_wave_cycles": d["valu_active_over_wave_cycles"],
                  "wait_any_over_wave_cycles": d["wait_any_over_wave_cycles"], "waves_per_simd": 2,
                  "note": "2 waves per SIMD x this per-wave VALU-active fraction = VALU pipe ~95% busy while the waves are resident"}
        except Exception:
            pass
        res = {
            "metric": "env-steps/sec (whole node), Panda-push 128k envs",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "pandaPushGymEnv joint-control step, %d envs per GPU (%d in total), 150 PGS iters, "
                                   "dt 1/240, obj_pose_rnd_std 0.05, tg_pose_rnd_std 0.2, actions U(-1,1) resident in HBM"
                                   % (n_local, total),
                       "envs_total": total, "envs_per_gpu": n_local, "parallelism": "dp%d" % world,
                       "collective": "rccl gather of [obs|reward|done] rows to rank 0 per step, overlapped with the next step"
                                     if world > 1 else "none",
                       "outputs_finite": finite, "start_state": "fresh reset() of every env (reference reset_simulation)",
                       "complex_env_frac_after_timed_steps_rank0": complex_after / n_local},
            "steady_state": steady,
            "strong_scaling_128k_total": strong,
            "sharded_consumers_no_gather": no_gather,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_fast<7>", "kernel_ms": kern_ms, "step_launch_pair_ms": pair_ms, "algorithmic_bytes_per_launch": ALG_BYTES_PER_ENV_STEP * n_local,
                         "note": "path is fp32-VALU/dependency bound (AI ~900 FLOP/B >> 25 FLOP/B machine balance); "
                                 "HBM fraction is small by construction, see valu"},
            "valu": {"achieved": ach_tf, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP32_VALU_PEAK_TFLOPS,
                     # the 157.3 TF peak assumes v_pk_fma_f32 at full rate; measured on this chip (profiles/r01_ubench_pkfma.txt) a
                     # packed FMA takes ~2 passes, so the scalar-FMA peak (78.6 TF) is the practical ceiling of fp32 FMA code
                     "peak_scalar_fma": FP32_VALU_PEAK_TFLOPS / 2, "frac_scalar_fma": ach_tf / (FP32_VALU_PEAK_TFLOPS / 2),
                     "flop_per_env_step": ALG_FLOP_PER_ENV_STEP, "sq_counters": sq, "vgprs_k_fast": info[0], "vgprs_k_fast_rc": info[6], "vgprs_k_step": info[1]},
        }
        if world == 1 and not args.no_other_configs:
            res["other_configs"] = other_configs(torch, dev)
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
