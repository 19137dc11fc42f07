#!/usr/bin/env python
"""bench.py -- env-steps/sec of the batched Panda-push step() hot path on MI355X.

Workload (BASELINE.json `metric`): pandaPushGymEnv, joint control, obj_pose_rnd_std=0.05,
tg_pose_rnd_std=0.2, 150 PGS iterations, dt=1/240, actions U(-1,1) i.i.d. per (step, env), **131072 envs in total**, done
envs re-initialised inside the step that finishes them (PBRE_F_AUTO_RESET) and counted as steps.
A "step" = one batched env.step() of every env: pbre_step_device, actions and [obs|reward|done] rows resident in HBM.

The headline (`value`) is the STEADY STATE: the timed K steps follow an untimed pre-roll of >= 1000 steps, so the batch holds
episodes of every age, a fraction of the envs has a joint at its limit or a robot contact ("complex" envs) and envs finish and
restart inside the timed region.  The figure right after a fresh reset() (no complex env yet) is reported as `fresh_reset`.

N > 1 (one process per GPU, RCCL over xGMI): the SAME 131072 envs are sharded over the N GPUs (strong scaling, BASELINE
config 4: 16384 envs per GPU at N = 8) and one RCCL gather per step returns the stacked rows to rank 0, overlapped with the next
step's kernels (an OPEN loop: `value`).  Extra keys: `closed_loop` (action scatter -> step -> gather -> policy on rank 0, nothing
overlapped: what a policy that needs obs(t) for a(t + 1) gets), `weak_scaling_128k_per_gpu` (every GPU keeps a 131072-env shard) and
`sharded_consumers_no_gather`.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without a launcher: starts its own N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALG_BYTES_PER_ENV_STEP = 444.0      # read 172 B + write 272 B (Panda push, joint control)
# FLOPs actually required per env-step by the sparse formulation the fast kernel uses: rounds 1-2, per PGS iteration
# 9 motor rows x 23 + 4 normal rows x 18 + 8 friction rows x 20 = 439 flop, x150, + ~8 k for kinematics/dynamics/obs.
# (SURVEY 8(d)'s 0.4 MFLOP assumed 33 dense rows of 70 flop; the dense figure is what the general row kernel does.)
# Round 3: the 9 motor rows of a contact-free env are evaluated in closed form (matrix power, ~4.2 k FMA) instead of 150 sweeps.
# Round 5: so is the tail of the object's rows -- 22 explicit sweeps x (4 x 18 + 8 x 20), then (S, s)^128 by binary powering and its
# validity bound (~3.2 k FMA).  The figure counts what the kernel now has to execute: 22 x 232 + 6400 + 8400 + ~8000.
ALG_FLOP_PER_ENV_STEP = 22 * 232 + 6400.0 + 8400.0 + 8000.0
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector
TOTAL_ENVS = 131072                 # BASELINE.json: "Panda-push 128k envs"
PROFILE_TAG = "r06"                 # profiles/<tag>_pmc_*.json: counter summaries of this command (tools/pmc.sh, tools/pmc_sq.sh)


def _profile(name, key=None):
    for tag in (PROFILE_TAG,):             # this round's counter passes only (round-2 verdict: no stale profiles behind the line)
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, name))))
            return (d[key] if key else d), "profiles/%s_%s.json" % (tag, name)
        except Exception:
            continue
    return None, None


# ------------------------------------------------------------------------------------------------ CPU baselines
def _cpu_worker(args):
    n, steps, seed = args
    import numpy as np
    import orc
    from pybullet_robot_envs.model.table import panda_table
    tbl, _ = panda_table()
    o = orc.Oracle(tbl, task=1)
    o.task.obj_pose_rnd_std = 0.05
    o.task.tg_pose_rnd_std = 0.2
    # resetting n envs costs 201 steps each, so reset a few and tile them
    st0, _ = o.batch_reset(min(n, 8), env_id0=seed * 1000)
    st = np.tile(st0, (n // st0.shape[0] + 1, 1))[:n].copy()
    rng = np.random.default_rng(seed)
    acts = rng.uniform(-1, 1, (steps, n, 7))
    t0 = time.perf_counter()
    for k in range(steps):
        st, _ = o.batch_step(st, acts[k])
    return n * steps, time.perf_counter() - t0


def _pybullet_worker(args):
    """The reference-equivalent loop on real PyBullet (tests/live_pybullet.py), without the reference's time.sleep(1/240)
    (panda_push_gym_env.py:237: the sleep is GUI pacing, not simulation work).  Only runs where `import pybullet` works."""
    steps, seed = args
    import numpy as np
    import live_pybullet
    env = live_pybullet.LivePandaPush(seed=seed)
    env.reset()
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    for k in range(steps):
        _, _, done = env.step(rng.uniform(-1, 1, 7))
        if done:
            env.reset()
    return steps, time.perf_counter() - t0


def cpu_baseline(target_cpu_seconds=20.0):
    """CPU baseline on this box's host cores, bounded sample.  Real PyBullet (kind "reference") when it is importable --
    BASELINE.md 3.1 -- otherwise the oracle (CPU port of the same step, double precision; kind "port")."""
    import concurrent.futures as cf
    cores = len(os.sched_getaffinity(0))
    try:
        import pybullet  # noqa: F401
        have_pb = True
    except Exception:
        have_pb = False
    if have_pb:
        w1, t1 = _pybullet_worker((200, 0))
        steps = max(200, int(target_cpu_seconds / (t1 / w1)))
        with cf.ProcessPoolExecutor(max_workers=cores) as ex:
            res = list(ex.map(_pybullet_worker, [(steps, i + 1) for i in range(cores)]))
        return {"value": sum(r[0] for r in res) / max(r[1] for r in res), "unit": "env-steps/s", "cores": cores, "kind": "reference",
                "single_core": w1 / t1,
                "sample": "PyBullet DIRECT, one process per core, %d steps each, time.sleep removed" % steps}
    n, steps = 256, 8
    work, dt = _cpu_worker((n, steps, 0))
    per = dt / work
    single_core = work / dt                 # the single-thread figure (this process, one core)
    total = max(cores * n * steps, int(target_cpu_seconds / per))
    steps = 16
    n = max(16, total // (cores * steps))
    t0 = time.perf_counter()
    with cf.ProcessPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(_cpu_worker, [(n, steps, i + 1) for i in range(cores)]))
    wall = time.perf_counter() - t0
    done = sum(r[0] for r in res)
    busy = max(r[1] for r in res)
    return {"value": done / busy, "unit": "env-steps/s", "cores": cores, "kind": "port", "single_core": single_core,
            "pybullet": "absent on this box (import pybullet fails): PyBullet baseline unavailable, the oracle is the stated substitute",
            "sample": "%d envs x %d steps per core on %d cores (oracle/pbre_oracle.c, fp64, 150 PGS iters), "
                      "%.1f s wall incl. process start" % (n, steps, cores, wall)}


# ------------------------------------------------------------------------------------------------ other configs (N = 1)
def other_configs(torch, dev, steps=10):
    """Extra, informative lines for the other BASELINE configs on one GPU (never the headline value): the batched
    iCubReach-v0 (config 1's env, 32768 envs) and the iCub with hands (config 5: 60 simulated DoF, the reference demo's scripted
    grasp holding the brick in the air, 8192 envs = 65536 / 8), each with its own roofline / valu objects."""
    import math as m
    import numpy as np
    from pybullet_robot_envs import _capi
    out = {}

    def timed(eng, acts, nsteps=None):
        nsteps = nsteps or steps
        o = torch.zeros((eng.num_envs, eng.obs_dim + 2), device=dev)
        s = _capi.torch_stream(dev)
        for k in range(2):
            eng.step_device(acts[k % len(acts)].data_ptr(), o.data_ptr(), s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nsteps):
            eng.step_device(acts[k % len(acts)].data_ptr(), o.data_ptr(), s)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return {"envs": eng.num_envs, "value": eng.num_envs * nsteps / el, "unit": "env-steps/s", "ms_per_step": el / nsteps * 1e3,
                "outputs_finite": bool(torch.isfinite(o).all())}, o

    def roof(r, eng, alg_bytes, pmc_name, valu_key):
        kms = float(eng.timing()[3])
        if kms > 0:
            gbs = alg_bytes * eng.num_envs / (kms * 1e-3) / 1e9
            r["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                             "kernel_ms": kms, "algorithmic_bytes_per_env_step": alg_bytes,
                             "note": "fp32-VALU / dependency bound like the Panda path; see valu"}
        d, src = _profile(pmc_name)
        if d and valu_key in d:      # (a counter pass that failed leaves a summary without the key: the timing must not be lost over it)
            envs_per_wave = r.pop("_envs_per_wave")
            r["valu"] = {"valu_insts_per_env_step": d[valu_key] / envs_per_wave,
                         "valu_active_over_wave_cycles": d.get("valu_active_over_wave_cycles", d.get("sq_active_inst_valu_over_wave_cycles")),
                         "source": src}
        r.pop("_envs_per_wave", None)

    try:
        # BASELINE configs 2 and 3 (parity-test cases, reported here for completeness): Panda reach, 4096 envs, object frozen and
        # contact-free; Panda push, 32768 envs, joint control.  Both in their stationary mix (600 untimed steps, auto-reset on).
        from pybullet_robot_envs.model.table import panda_table
        tblp, _ = panda_table()
        for key, n_, task, fl, label in (("panda_reach_config2", 4096, _capi.TASK_REACH, _capi.F_NO_OBJECT | _capi.F_AUTO_RESET,
                                          "pandaReachGymEnv-style reach, 4096 envs, object frozen and contact-free (free-space dynamics only)"),
                                         ("panda_push_config3", 32768, _capi.TASK_PUSH, _capi.F_AUTO_RESET,
                                          "pandaPushGymEnv joint-control step, 32768 envs, box-on-table contact + 150 PGS sweeps")):
            eng = _capi.Engine(tblp, task=task, num_envs=n_, flags=fl, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, device_id=dev.index or 0)
            eng.reset()
            op = torch.zeros((n_, eng.obs_dim + 2), device=dev)
            sp = _capi.torch_stream(dev)
            fresh = torch.empty((n_, 7), device=dev)
            for k in range(600):                   # i.i.d. actions per (step, env): a fresh draw per pre-roll step (a short recycled pool
                fresh.uniform_(-1.0, 1.0)          # biases every env's random walk and drives the joints into their limits)
                eng.step_device(fresh.data_ptr(), op.data_ptr(), sp)
            actp = [torch.rand((n_, 7), device=dev) * 2 - 1 for _ in range(100)]
            rp, _ = timed(eng, actp, 100)
            rp["workload"] = label + "; stationary mix (600 untimed steps first)"
            rp["complex_envs"] = int(eng.kernel_info()[5])
            out[key] = rp
            eng.close()
    except Exception as e:
        out["panda_configs_2_3"] = {"error": repr(e)}
    try:
        from pybullet_robot_envs.envs import iCubReachGymEnv
        acts = [torch.rand((32768, 3), device=dev) * 2 - 1 for _ in range(8)]
        # (1) the env in its stationary mix under random actions, the headline's protocol: auto-reset, episode clocks de-synchronised (step
        # counters U{0..max_steps-1}), max_steps + 500 untimed steps with fresh i.i.d. actions, then the timed ones.  (Rounds 1-2 recycled a pool
        # of 8 action tensors -- a constant mean action per env, the commanded hand pose drifts into a workspace corner -- and kept the
        # clocks synchronised, which measures one phase of the episode.)  With IK control a wave's IK runs until its slowest env has
        # converged or given up after 100 iterations; far into an episode some env of nearly every wave is at an out-of-reach target.
        # This leg runs first: for about a second after the Panda legs every kernel of this latency-bound, half-empty workload runs
        # ~1.6x slower (PBRE_ICUB_TRACE shows all kernels stretched alike: clocks, not scheduling) -- the pre-roll absorbs that
        steady = {}
        sh = _capi.torch_stream(dev)
        for key, use_ik, adim in (("ik_control", 1, 3), ("joint_control", 0, None)):
            env = iCubReachGymEnv(use_IK=use_ik, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, num_envs=32768, auto_reset=True)
            env.reset()
            eng = env._engine
            adim = eng.act_dim
            st = eng.get_state()
            st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, int(env._max_steps), 32768).astype(np.float32)
            eng.set_state(st)
            o = torch.zeros((32768, eng.obs_dim + 2), device=dev)
            fresh = torch.empty((32768, adim), device=dev)
            pre = int(env._max_steps) + 500
            for k in range(pre):
                fresh.uniform_(-1.0, 1.0)
                eng.step_device(fresh.data_ptr(), o.data_ptr(), sh)
            r2, _ = timed(eng, [torch.rand((32768, adim), device=dev) * 2 - 1 for _ in range(40)], 40)
            steady[key] = {"value": r2["value"], "unit": "env-steps/s", "ms_per_step": r2["ms_per_step"], "preroll_steps": pre, "max_steps": int(env._max_steps),
                           "episode_clocks": "de-synchronised", "actions": "i.i.d. U(-1,1) per (step, env)",
                           "envs_with_robot_object_contact": int(eng.kernel_info()[5]), "outputs_finite": r2["outputs_finite"]}
            env.close()
        steady.update(steady["ik_control"])        # (the keys of rounds 1-2: the IK-control figure)
        # (2) contact-free: iCubReach-v0 kwargs; after reset() 1500 untimed steps with ZERO actions (the hand holds its home pose, the state
        # stays the post-reset one) so that the GPU's clocks are those of THIS half-empty, latency-bound load, then the timed steps with
        # random actions.  (Timing 12 steps right after reset() measures the clock state the reset left behind: 0.34 ms per step after a
        # reset done by the heavy lane-group kernel, 0.65 ms after one done by the pipeline itself.)
        env = iCubReachGymEnv(use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, num_envs=32768)
        env.reset()
        o = torch.zeros((32768, env._engine.obs_dim + 2), device=dev)
        zero = torch.zeros((32768, 3), device=dev)
        for k in range(1500):
            env._engine.step_device(zero.data_ptr(), o.data_ptr(), sh)
        r, _ = timed(env._engine, acts)
        r["workload"] = "iCubReach-v0 (IK position control of the left hand), 32768 envs, hand at its home pose (1500 zero-action steps after reset())"
        r["pipeline"] = ("kw_lane_ik (Cartesian control; targets handed over per env, beside everything else) || kw_dyn (1 thread / env) -> kw_quad (4 lanes / env; envs whose hand touches the object: kw_quad_rc) -> kw_fin"
                         if env._engine.kernel_info()[2] else "lane-group kernel")
        r["_envs_per_wave"] = 16 if env._engine.kernel_info()[2] else 2
        # q, qd of the 20 simulated DoF each way + object 13 f each way + action 3 f + obs 31 f + reward/done + counters
        roof(r, env._engine, 4.0 * (2 * 40 + 2 * 13 + 3 + 31 + 2 + 4), "pmc_icub", "sq_insts_valu_per_wave")
        r["steady_random_actions"] = steady
        env.close()
        # (3) the same measurement at 131072 envs: the pipeline's kernels are latency chains at 32768 envs (half the SIMDs hold
        # one wave), a bigger batch fills them
        env = iCubReachGymEnv(use_IK=1, control_arm='l', control_orientation=0, obj_pose_rnd_std=0, num_envs=131072)
        acts4 = [torch.rand((131072, 3), device=dev) * 2 - 1 for _ in range(4)]
        env.reset()
        o4 = torch.zeros((131072, env._engine.obs_dim + 2), device=dev)
        zero4 = torch.zeros((131072, 3), device=dev)
        for k in range(600):
            env._engine.step_device(zero4.data_ptr(), o4.data_ptr(), sh)
        r4, _ = timed(env._engine, acts4)
        r["same_at_131072_envs"] = {"value": r4["value"], "unit": "env-steps/s", "ms_per_step": r4["ms_per_step"], "outputs_finite": r4["outputs_finite"]}
        out["icub_reach"] = r
        env.close()
    except Exception as e:
        out["icub_reach"] = {"error": repr(e)}
    try:
        from pybullet_robot_envs import _client
        from pybullet_robot_envs.envs.icub_envs.icub_env_with_hands import iCubHandsEnv
        cid = _client.connect(8192)
        robot = iCubHandsEnv(cid, use_IK=1, control_arm='r')
        # BASELINE config 5, "iCub-hand grasp": the reference demo's scripted sequence (helloworld_icub.py:61-107) up to the lift -- above
        # the brick, hand turned, fingers closed on it (4 fingertips in contact), hand raised by 18 cm with the brick in it -- then the
        # timed steps HOLD the brick in the air: the hand pose command jitters by a few mm around the lifted pose, the finger motors keep
        # squeezing (force 10).  Every env is in the coupled solve: 3-4 fingertip contacts, no object-table contact.
        def quat(e):
            cr, sr, cp, sp, cy, sy = m.cos(e[0] / 2), m.sin(e[0] / 2), m.cos(e[1] / 2), m.sin(e[1] / 2), m.cos(e[2] / 2), m.sin(e[2] / 2)
            return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
        e2 = [m.pi / 2, m.pi / 3, -m.pi]
        pos_cl = [0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 0, 0.6, 0.8, 1.0, 1.57, 0.8, 0.5, 0.8]
        robot.pre_grasp(); robot.step_simulation(10)
        robot.apply_action([0.49, 0.0, 0.8] + quat([0, 0, m.pi / 2]), max_vel=5); robot.pre_grasp(); robot.step_simulation(60)
        robot.apply_action([0.485, 0.0, 0.72] + quat(e2), max_vel=5); robot.pre_grasp(); robot.step_simulation(60)
        robot.grasp(pos_cl); robot.step_simulation(60)
        robot.apply_action([0.45, 0, 0.9] + quat(e2), max_vel=5); robot.grasp(pos_cl); robot.step_simulation(60)
        z_obj = float(np.atleast_2d(np.asarray(robot.get_object_pose()))[:, 2].mean())
        # the engine's IK entry point takes Euler angles; (pi/2, pi/3, -pi) lies outside the class's clipping range for roll, which the
        # quaternion path above is not subject to, so the timed commands are the equivalent in-range angles
        st0 = robot._engine.get_state()[0]
        hp = st0[robot._engine.x_off + 6:robot._engine.x_off + 12].astype("float32")
        base = torch.tensor(hp, dtype=torch.float32, device=dev)
        jit = torch.tensor([0.004, 0.004, 0.002, 0.0, 0.0, 0.0], device=dev)
        r, o = timed(robot._engine, [base + (torch.rand((8192, 6), device=dev) - 0.5) * jit for _ in range(4)])
        r["workload"] = ("iCubHandsEnv (60 simulated DoF, one env per wavefront): the reference's scripted grasp, brick held %.0f cm above the "
                         "table by the closed fingers (force 10), hand pose jittering through IK control, 8192 envs" % (100 * (z_obj - 0.65)))
        r["brick_height_above_rest_m"] = z_obj - 0.65
        r["mean_fingertips_in_contact"] = float(o[:, -4].mean())
        r["mean_robot_object_contact_points"] = float(o[:, -3].mean())
        r["_envs_per_wave"] = 1
        roof(r, robot._engine, 1750.0, "pmc_hands", "valu_insts_per_wave")      # ~1.75 KB per env-step
        out["icub_hands_config5"] = r
        _client.disconnect(cid)
    except Exception as e:
        out["icub_hands_config5"] = {"error": repr(e)}
    return out


EV_EVERY = int(os.environ.get("PBRE_BENCH_EV_EVERY", "8"))     # torch event pairs around every EV_EVERY-th step of the timed region


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=TOTAL_ENVS, help="envs in TOTAL (sharded over the GPUs)")
    ap.add_argument("--preroll", type=int, default=1000, help="untimed steps before the timed region of the headline (steady state)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fresh", action="store_true", help="skip the extra measurement right after reset()")
    ap.add_argument("--no-weak", action="store_true", help="skip the extra 131072-envs-per-GPU measurement at N>1")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-inclusive pbre_step measurement (N=1 only)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the informative iCub / iCub-with-hands lines (N=1 only)")
    ap.add_argument("--no-shards", action="store_true", help="skip the per-GPU shard sizes of the strong-scaling split measured on this one GPU (N=1 only)")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (the same command line the driver uses)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stderr.write(p.stdout[-4000:] + "\n" + p.stderr[-8000:] + "\n")
        raise SystemExit(p.returncode or 1)
    print(line)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import numpy as np
    import torch
    import torch.distributed as dist
    from pybullet_robot_envs import _capi
    from pybullet_robot_envs.model.table import panda_table
    from pybullet_robot_envs.sharding import ShardedEngine, GatherPipeline, CtxGatherPipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # test hooks (single-GPU box): PBRE_BENCH_ONE_DEVICE=1 puts every rank on GPU 0, PBRE_BENCH_BACKEND=gloo stages the
    # gather through host memory (RCCL refuses two ranks on one device).  The driver's runs use neither.
    one_dev = os.environ.get("PBRE_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    backend = os.environ.get("PBRE_BENCH_BACKEND", "gloo" if one_dev else "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    tbl, _ = panda_table()
    early_other = other_configs(torch, dev) if (os.environ.get("PBRE_BENCH_OTHER_FIRST") == "1" and world == 1) else None     # (diagnostic)
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

    def clean(d):
        return dict((k, v) for k, v in d.items() if not k.startswith("_"))

    class Job(object):
        """One sharded batch: engine, resident action pool for the timed steps, double-buffered output rows, pipelined gather."""

        def __init__(self, total_envs, pool_steps, desync=True):
            self.sh = ShardedEngine(tbl, total_envs, task=_capi.TASK_PUSH, device_id=local_rank, seed=1234,
                                    obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
            self.total = total_envs
            self.eng = eng = self.sh.engine
            self.n_local = n = self.sh.n_local
            assert self.sh.env_id_base == rank * n
            eng.reset()
            # De-synchronise the episode clocks: after reset() every env has step counter 0, so all envs that do not succeed early
            # would hit max_steps (1000) in the same step and the whole batch would restart together, for ever in phase.  A rollout
            # worker that has been running for a while has episodes of every age: start the counters at U{0..max_steps-1} (keyed by
            # the global env id, so the sharded batch is the same batch).
            st = eng.get_state()
            ages = np.random.default_rng(4321).integers(0, 1000 if (desync and os.environ.get("PBRE_BENCH_NO_DESYNC") != "1") else 1, total_envs)[self.sh.env_id_base:self.sh.env_id_base + n]
            st[:, eng.x_off + 3] = ages.astype(np.float32)
            eng.set_state(st)
            # actions U(-1,1), i.i.d. per (step, env).  Timed steps read slices of a pool generated up front and
            # resident in HBM; the untimed pre-roll draws a fresh slice per step (a short recycled pool would add a constant
            # bias to every env's random walk and drive the joints into their limits within a few hundred steps)
            self.gen = torch.Generator(device=dev)
            self.gen.manual_seed(1234 + rank)
            self.pool = torch.rand((pool_steps, n, eng.act_dim), device=dev, generator=self.gen) * 2 - 1
            self.fresh = torch.empty((n, eng.act_dim), device=dev)
            # the data path: step kernels + one asynchronous, double-buffered gather per step.  With several GPUs the gather is the
            # context's own (include/pbre.h: pbre_comm_init / pbre_step_gather_device -- ncclSend / ncclRecv into rank 0 enqueued from C on
            # the ctx's communication stream; sharding.CtxGatherPipeline); should its communicator not come up on EVERY rank, all ranks
            # fall back to torch.distributed's gather (sharding.GatherPipeline).  PBRE_BENCH_CTX_COMM=0 pins the fallback.
            self.pipe, self.comm_kind, self.comm_note = None, "torch.distributed.gather", None
            # (PBRE_BENCH_CTX_COMM=force: also with a non-nccl bootstrap backend -- the single-GPU control-flow run with tests/fake_rccl as PBRE_RCCL_LIB)
            if world > 1 and (backend == "nccl" or os.environ.get("PBRE_BENCH_CTX_COMM") == "force") and os.environ.get("PBRE_BENCH_CTX_COMM", "1") != "0":
                ok, pipe = 1.0, None
                try:
                    pipe = CtxGatherPipeline(self.sh, dev)
                except Exception as e:
                    ok, self.comm_note = 0.0, repr(e)
                if min(max_over_ranks([-ok])) > -1.0 + 1e-9:      # (max of -ok = -min of ok) some rank failed
                    pipe = None
                if pipe is not None:
                    self.pipe, self.comm_kind = pipe, "context-owned RCCL communicator (pbre_step_gather_device)"
            if self.pipe is None:
                self.pipe = GatherPipeline(self.sh, dev, gather=True, host_staged=(backend != "nccl"))
            self.steps_done = 0

        @property
        def gather(self):
            return self.pipe.gather

        @gather.setter
        def gather(self, v):
            self.pipe.gather = v

        def step(self, ev=None, act=None):
            if act is None:
                act = self.pool[self.pipe.k % self.pool.shape[0]]
            if ev is not None:
                ev[0].record()
            self.pipe.step(act, stream, timing_events=ev)
            self.steps_done += 1

        def preroll(self, upto):
            """untimed steps with fresh i.i.d. actions until the batch has taken `upto` steps since reset()"""
            while self.steps_done < upto:
                self.fresh.uniform_(-1, 1, generator=self.gen)
                self.step(act=self.fresh)
            self.drain()

        def drain(self):
            self.pipe.drain()

        def timed(self, steps, warmup, events=False):
            for _ in range(warmup):
                self.step()
            self.drain()
            barrier()
            # event pairs around every EV_EVERY-th (8th) step only: an event record is a barrier packet on the stream, and a pair per step
            # costs ~10% of the 0.2 ms kernel it brackets
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] if events else None
            # one HIP event pair around the WHOLE region on the stream the step kernels are launched on (`side` is torch's current stream):
            # a step is one launch and launches follow each other without a gap (profiles/r05_step_kernels.txt), so span / steps is the
            # dominant kernel's average launch duration over exactly the steps `ms_per_step` is the wall time of -- the same statistic,
            # and never longer than the step (roofline.kernel_ms must not exceed ms_per_step)
            r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            r0.record(side)
            for k in range(steps):
                self.step(evs[k] if (events and k % EV_EVERY == 0) else None)
            r1.record(side)
            self.drain()
            barrier()
            elapsed = time.perf_counter() - t0
            torch.cuda.synchronize()
            pair = float(np.mean([a.elapsed_time(b) for a, b in evs[::EV_EVERY]])) if events else 0.0
            el, pr, dev_ms = max_over_ranks([elapsed, pair, r0.elapsed_time(r1) / steps])
            return {"value": self.total * steps / el, "unit": "env-steps/s", "ms_per_step": el / steps * 1e3, "_elapsed": el, "_pair_ms": pr, "_dev_ms": dev_ms}

        def complex_frac(self):
            return self.eng.kernel_info()[5] / float(self.n_local)

        def timed_closed_loop(self, steps, warmup):
            """(side key, N > 1, context-owned communicator) the CLOSED loop: a policy on rank 0 computes a(t + 1) from the gathered
            observations of step t, so per step: action scatter (rank 0 -> ranks) -> step kernels -> row gather (ranks -> rank 0) -> policy,
            nothing overlapped.  The policy is a stand-in of negligible cost that really reads the gathered rows (tanh of the observed joint
            angles + resident noise), so the data dependence is there."""
            pipe = self.pipe
            acts = torch.zeros((self.total, self.eng.act_dim), device=dev) if rank == 0 else None
            noise = (torch.rand((8, self.total, self.eng.act_dim), device=dev, generator=self.gen) - 0.5) if rank == 0 else None
            def one(k):
                b = pipe.closed_loop_step(acts, stream)
                if rank == 0:
                    torch.tanh(pipe.all[b][:, 9:9 + self.eng.act_dim] * 3.0, out=acts)
                    acts.mul_(0.6).add_(noise[k % 8])
                self.steps_done += 1
            for k in range(warmup):
                one(k)
            self.drain(); barrier()
            t0 = time.perf_counter()
            for k in range(steps):
                one(k)
            self.drain(); barrier()
            el = max_over_ranks([time.perf_counter() - t0])[0]
            return {"value": self.total * steps / el, "unit": "env-steps/s", "ms_per_step": el / steps * 1e3,
                    "loop": "per step: pbre_scatter_actions_device -> step kernels -> pbre_step_gather_device's exchange -> pbre_gather_wait -> policy on rank 0 (reads the gathered rows)"}

        def timed_with_residual_threshold(self, steps, thr=1e-7, settle=100):
            """(side key) the same stationary batch stepped with Bullet's exit test of the sweep loop on (pbre_physics.solver_residual_threshold
            = PyBullet's documented solverResidualThreshold default; the engine's default is 0 = all 150 sweeps): `settle` untimed steps, the
            timed ones, the per-env sweep counts of the last step; then the threshold is switched off again."""
            self.drain()
            self.eng.set_physics(solver_residual_threshold=thr)
            try:
                for _ in range(settle):
                    self.fresh.uniform_(-1, 1, generator=self.gen)
                    self.step(act=self.fresh)
                self.drain()
                r = clean(self.timed(steps, 5))
                sw = self.eng.get_sweeps()
                r.update({"solver_residual_threshold": thr, "sweeps_last_step": {"median": float(np.median(sw)), "mean": float(sw.mean()), "p90": float(np.percentile(sw, 90)),
                                                                               "frac_at_the_cap": float((sw >= 150).mean())}})
            finally:
                self.drain()
                self.eng.set_physics(solver_residual_threshold=0.0)
            return r

    total = args.envs
    job = Job(total, 2 * (args.steps + args.warmup))
    eng, n_local = job.eng, job.n_local

    # (side key) right after reset(): no complex env yet (only the step counters differ between envs)
    fresh = None
    if not args.no_fresh:
        fresh = job.timed(args.steps, args.warmup)
        fresh.update({"start_state": "fresh reset() of every env (reference reset_simulation: robots at the home pose, no complex env); step counters de-synchronised", "complex_env_frac_rank0": job.complex_frac()})

    # headline: steady state after the pre-roll
    job.preroll(args.preroll - args.warmup)
    csum0 = eng.kernel_info()[7]
    head = job.timed(args.steps, args.warmup, events=True)
    complex_per_step = ((eng.kernel_info()[7] - csum0) % (1 << 31)) / float(args.steps + args.warmup)
    # (side key) the same timed region five more times, back to back, without event pairs: the spread of the headline on this box
    regions = [head] + [job.timed(args.steps, 0) for _ in range(5)]
    rep = [r["ms_per_step"] for r in regions]
    region_dev_ms = float(np.median([r["_dev_ms"] for r in regions]))      # device time per launch: median over the same six regions as `ms_per_step`
    repeats = {"ms_per_step": {"median": float(np.median(rep)), "min": float(np.min(rep)), "max": float(np.max(rep))}, "samples": len(rep),
               "value_at_median": total * 1e3 / float(np.median(rep)),
               "note": "six timed regions of exactly --steps steps each on the same stationary batch, back to back (sample 0 with HIP event pairs around every 8th step); `value` is the median's"}
    # `value` / `ms_per_step`: the MEDIAN of these six timed regions of exactly --steps steps each (the driver's 20 steps are a 4 ms region; a
    # single one carries the box's scheduling noise); the first region alone is kept as `first_timed_region`
    first_region = {"ms_per_step": head["ms_per_step"], "value": head["value"]}
    head["ms_per_step"] = float(np.median(rep)); head["value"] = total * 1e3 / head["ms_per_step"]
    steps_before = job.steps_done - args.steps
    complex_after = job.complex_frac()
    finite = bool(torch.isfinite(job.pipe.out[0]).all() and torch.isfinite(job.pipe.out[1]).all())
    comm_kind, comm_note = job.comm_kind, job.comm_note
    comm_info = job.pipe.info() if hasattr(job.pipe, "info") else None
    done_frac = float(job.pipe.out[(job.pipe.k - 1) & 1][:, -1].mean())
    # the dominant kernel alone: mean of the HIP event pairs the library records around it on its stream -- round 5: the step IS one kernel
    # (k_fused: the complex envs' row waves and the simple envs' waves in one grid), so this is the step's device time without launch gaps
    kern_ms_sampled = float(eng.timing()[3])
    fused_now = eng.kernel_info()[13] > 0
    # one launch per step: the launch duration is the region's device span / steps (same regions, same median as `ms_per_step`); the two-kernel
    # step (PBRE_FUSED=0) keeps the library's sampled pairs around k_fast
    kern_ms = region_dev_ms if fused_now else kern_ms_sampled
    info = eng.kernel_info()
    episodes = float(torch.as_tensor(eng.get_state()[:, eng.x_off + 5]).mean()) if rank == 0 else 0.0
    # (side key, SURVEY section 5 "metrics": contact-count histogram) which contact kinds the envs of the stationary batch are in, from the
    # downloaded state of rank 0's first 16384 envs through the host-side restatement of the step's detection rule (model/contacts.py)
    contact_hist = None
    if rank == 0:
        try:
            from pybullet_robot_envs.model import contacts as _ct
            from pybullet_robot_envs.model.table import panda_table as _pt
            st_s = eng.get_state()[:16384]
            fl = _ct.contact_flags(_pt()[0], st_s, eng.ndof, eng.get_physics())
            nm = {_ct.OBJECT_TABLE: "object_table", _ct.ROBOT_OBJECT: "robot_object", _ct.ROBOT_TABLE: "robot_table"}
            contact_hist = {"envs_sampled": int(st_s.shape[0]),
                            "envs_by_contact_kinds": {("+".join(n for b, n in nm.items() if k & b) or "none"): int(c) for k, c in zip(*np.unique(fl, return_counts=True))},
                            "note": "contact = distance below the contact margin (the step's own detection rule) in the state after the timed region"}
        except Exception as e:      # informative only
            contact_hist = {"error": repr(e)}

    # (side key, N = 1) Bullet's residual exit on: the stationary batch at the headline size
    rt_side = None
    if world == 1 and os.environ.get("PBRE_BENCH_NO_RT") != "1":
        try:
            rt_side = {str(total): job.timed_with_residual_threshold(max(args.steps, 50))}
        except Exception as e:
            rt_side = {"error": repr(e)}

    # extra at N>1: the closed loop (a policy that needs obs(t) to produce a(t + 1): the gather cannot hide behind the next step)
    closed_loop = None
    if world > 1:
        if isinstance(job.pipe, CtxGatherPipeline):
            try:
                closed_loop = job.timed_closed_loop(args.steps, args.warmup)
                comm_info = job.pipe.info()
            except Exception as e:
                closed_loop = {"error": repr(e)}
        else:
            closed_loop = {"error": "needs the context-owned communicator (pbre_scatter_actions_device); this run fell back to torch.distributed's gather"}

    # extra at N>1: the same steps without the gather (every rank's consumer reads its own shard's rows)
    no_gather = None
    if world > 1:
        try:
            job.gather = False
            no_gather = clean(job.timed(args.steps, args.warmup))
        except Exception as e:
            no_gather = {"error": repr(e)}

    # extra at N=1: what env.step() delivers through the host-buffer entry point (action upload + kernels + row download
    # through pinned staging buffers, SURVEY 8(d)'s literal metric)
    host = None
    if world == 1 and not args.no_host_path:
        try:
            # (64 i.i.d. action batches on the host: a pool of four, recycled, pushes thousands of envs into joint limits within a few hundred steps)
            g_h = torch.Generator(device=dev); g_h.manual_seed(4242 + rank)
            acts = [(torch.rand((n_local, job.pool.shape[2]), device=dev, generator=g_h) * 2 - 1).cpu().numpy() for _ in range(64)]
            # An engine of its own, stepped through the host-buffer entry points only (its batch: reset, de-synchronised episode clocks, 300
            # host-path steps with i.i.d. actions).  On the job's engine -- which has taken thousands of pbre_step_device launches on torch's
            # stream -- the same pipelined loop measures 0.93 ms per step instead of 0.39 (wait for the rows 0.78 ms instead of 0.26;
            # profiles/r06_host_path.txt): unexplained so far, PBRE_BENCH_HOST_JOB_ENGINE=1 measures it there.
            host_engine = "own"
            if os.environ.get("PBRE_BENCH_HOST_JOB_ENGINE") == "1":
                host_engine = "job"
            else:
                eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n_local, device_id=local_rank, env_id_base=rank * n_local, seed=1234,
                                   obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET)
                eng.reset()
                st_h = eng.get_state()
                st_h[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n_local).astype(np.float32)
                eng.set_state(st_h)
                for k in range(300):
                    eng.step(acts[(7 * k) % 64], copy=False)
            for k in range(2):
                eng.step(acts[k], copy=False)
            t0 = time.perf_counter()
            for k in range(10):
                eng.step(acts[(5 * k + 1) % 64], copy=False)
            el = time.perf_counter() - t0
            ms = eng.timing()
            sync = {"value": n_local * 10 / el, "unit": "env-steps/s", "ms_per_step": el / 10 * 1e3, "h2d_ms": ms[0], "kernels_ms": ms[1], "d2h_ms": ms[2],
                    "note": "Engine.step(): one step at a time, host-synchronous (zero-copy: the kernels reach over PCIe themselves)"}
            # round 6: the same metric PIPELINED (pbre_step_async / pbre_step_wait: upload, kernels and download of consecutive steps on three
            # streams, two steps in flight) -- an open loop, like `value`; the floor is the row download over PCIe
            ns = 40
            for k in range(4):
                eng.step_pipelined(acts[(3 * k + 2) % 64])
            calls = []
            t0 = time.perf_counter()
            for k in range(ns):
                tc = time.perf_counter()
                eng.step_pipelined(acts[(11 * k + 5) % 64])      # (copy the actions into the page-locked slot, wait for the rows of the step before last, enqueue)
                calls.append(time.perf_counter() - tc)
            el = time.perf_counter() - t0
            eng.step_wait(); eng.step_wait()
            row_mb = n_local * (eng.obs_dim + 2) * 4 / 1e6
            host = {"value": n_local * ns / el, "unit": "env-steps/s", "ms_per_step": el / ns * 1e3,
                    "rows_MB_per_step": row_mb, "d2h_GBps_if_download_bound": row_mb / (el / ns * 1e3),
                    "ms_per_call_last_8": [round(x * 1e3, 3) for x in calls[-8:]],
                    "host_phase_ms_per_call": dict(zip(("copy_actions", "wait_rows", "enqueue"), [round(x / max(1, eng._async["phase_s"][3]) * 1e3, 4) for x in eng._async["phase_s"][:3]])),
                    "synchronous": sync, "engine": host_engine, "complex_envs_at_the_end": int(eng.kernel_info()[5]), "one_launch_steps": int(eng.kernel_info()[13]),
                    "note": "SURVEY 8(d) literal metric: numpy actions in page-locked memory in, [obs|reward|done] rows out, upload + kernels + download, pipelined over "
                            "calls (Engine.step_async / step_wait, open loop, two steps in flight); never `value`, which is device-resident stepping (pbre_step_device)"}
            if host_engine == "own":
                eng.close()
        except Exception as e:
            host = {"error": repr(e)}
    del job

    # extra at N=1: the same pre-roll with SYNCHRONISED episode clocks (every env starts its first episode at step 0, so all envs that
    # do not succeed early restart together every 1000 steps): right behind such a mass restart the batch is close to the fresh state,
    # which is what a plain "reset(), 1000 steps, measure" protocol sees
    sync_clocks = None
    if world == 1 and not args.no_fresh:
        try:
            j3 = Job(total, args.steps + args.warmup, desync=False)
            j3.preroll(args.preroll - args.warmup)
            c0 = j3.eng.kernel_info()[7]
            sync_clocks = clean(j3.timed(args.steps, args.warmup))
            sync_clocks["complex_envs_per_step"] = ((j3.eng.kernel_info()[7] - c0) % (1 << 31)) / float(args.steps + args.warmup)
            del j3
        except Exception as e:
            sync_clocks = {"error": repr(e)}

    # extra at N=1: the per-GPU shards of BASELINE's strong-scaling split (131072 envs over 2 / 4 / 8 GPUs) measured on THIS GPU, fresh and
    # stationary (the headline's protocol: de-synchronised clocks, pre-roll, auto-reset counted; 200 timed steps -- below the
    # machine-filling batch a step's time depends on whether one of its few complex envs has a robot-object contact).  With the
    # driver's own SCALE runs absent (no 8-GPU node) this is what the 1 -> 8 curve can be projected from: N GPUs step the batch in the
    # time one GPU needs for its 131072 / N shard (+ the gather, which is overlapped).
    shards = None
    if world == 1 and not args.no_shards and total == TOTAL_ENVS:
        shards = {}
        try:
            for n_sh in (65536, 32768, 16384):
                js = Job(n_sh, 256)
                fr = clean(js.timed(100, 5))
                js.preroll(max(args.preroll, 1000))
                c0 = js.eng.kernel_info()[7]
                stt = clean(js.timed(200, 5))
                inf = js.eng.kernel_info()
                shards[str(n_sh)] = {"gpus_in_the_split": TOTAL_ENVS // n_sh, "fresh_ms_per_step": fr["ms_per_step"], "stationary_ms_per_step": stt["ms_per_step"],
                                     "fresh_env_steps_per_s": fr["value"], "stationary_env_steps_per_s": stt["value"],
                                     "complex_envs_per_step": ((inf[7] - c0) % (1 << 31)) / 205.0,
                                     "simple_env_kernel": "k_fast_pair (robot wave + object wave per 64 envs)" if inf[10] > 0 else "k_fast",
                                     "steps_that_were_one_launch": inf[13]}
                if n_sh == 16384 and isinstance(rt_side, dict) and "error" not in rt_side:
                    rt_side[str(n_sh)] = js.timed_with_residual_threshold(200)
                del js
            one = head["ms_per_step"]
            shards["projection"] = {"stationary_speedup_vs_one_gpu": {str(TOTAL_ENVS // int(k)): one / v["stationary_ms_per_step"] for k, v in shards.items() if k.isdigit()},
                                    "fresh_speedup_vs_one_gpu": ({str(TOTAL_ENVS // int(k)): fresh["ms_per_step"] / v["fresh_ms_per_step"] for k, v in shards.items() if k.isdigit()} if fresh else None),
                                    "note": "single-GPU measurements of the shard sizes, before the (overlapped) gather; not a multi-GPU run"}
        except Exception as e:
            shards["error"] = repr(e)

    # extra at N>1: weak scaling -- every GPU keeps the 131072-env shard (131072 x N envs in total), same gather
    weak = None
    if world > 1 and not args.no_weak:
        try:
            j2 = Job(total * world, args.steps + args.warmup)
            j2.preroll(args.preroll - args.warmup)
            weak = clean(j2.timed(args.steps, args.warmup))
            weak.update({"envs_total": total * world, "envs_per_gpu": j2.n_local})
            del j2
        except Exception as e:   # informative only
            weak = {"error": repr(e)}

    if rank == 0:
        kern_s = max(kern_ms, 1e-9) * 1e-3
        ach_gbs = ALG_BYTES_PER_ENV_STEP * n_local / kern_s / 1e9
        ach_tf = ALG_FLOP_PER_ENV_STEP * n_local / kern_s / 1e12
        # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, corrected as
        # calibrated in the profile file).  Counters cannot be read from inside this process: the figure is the committed
        # per-env summary of the counter passes over this same command, scaled to this launch's env count.
        traffic, traffic_src = None, None
        fused = info[13] > 0
        dom_kernel = ("k_fused<7, tail>" if (len(info) > 15 and info[15] > 0) else "k_fused<7>") if fused else "k_fast<7>"      # (tail: the instantiation with tail pairs)
        pmc, src = _profile("pmc_hbm", "step_kernel" if fused else "k_fast<7>")
        traffic_note = None
        if pmc:
            traffic, traffic_src = pmc["hbm_bytes_per_env_step"] * n_local, src
            traffic_note = ("counters of the kernel that steps the stationary batch (%s); FETCH_SIZE + WRITE_SIZE of one launch, complex envs' rows included"
                            % pmc.get("variant"))
        sq = None
        d, src = _profile("pmc_sq", "step_kernel" if fused else "k_fast<7>")
        if d:
            sq = {"valu_insts_per_wave": d["valu_insts_per_wave"], "valu_active_over_wave_cycles": d["valu_active_over_wave_cycles"],
                  "wait_any_over_wave_cycles": d["wait_any_over_wave_cycles"], "variant": d.get("variant"), "source": src}
        rccl = None
        if world > 1:
            try:
                rccl = {"backend": backend, "ranks_seen": (comm_info or {}).get("ranks_seen", dist.get_world_size()), "version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                        "gather": comm_kind, "context_communicator": comm_info, "note": comm_note}
            except Exception as e:
                rccl = {"backend": backend, "ranks_seen": dist.get_world_size(), "version": repr(e)}
        res = {
            "metric": "env-steps/sec (whole node), Panda-push 128k envs",
            "value": head["value"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "pandaPushGymEnv joint-control step, %d envs in total (%d per GPU), 150 PGS iters, dt 1/240, "
                                   "obj_pose_rnd_std 0.05, tg_pose_rnd_std 0.2, actions U(-1,1) resident in HBM, done envs re-initialised "
                                   "in the step (auto-reset) and counted" % (total, n_local),
                       "envs_total": total, "envs_per_gpu": n_local, "parallelism": "dp%d" % world,
                       "collective": "rccl gather of [obs|reward|done] rows to rank 0 per step, overlapped with the next step"
                                     if world > 1 else "none",
                       "rccl": rccl,
                       "outputs_finite": finite,
                       "start_state": "steady state: %d untimed steps since reset() before the timed region (pre-roll + warm-up); episode "
                                      "clocks de-synchronised (initial step counters U{0..999}), so episodes of every age coexist and "
                                      "envs finish and restart in every step" % steps_before,
                       "complex_env_frac_rank0": complex_after, "complex_envs_per_step_timed_region_rank0": complex_per_step,
                       "done_frac_last_step_rank0": done_frac,
                       "mean_episodes_completed_per_env_rank0": episodes},
            "repeats": repeats, "first_timed_region": first_region,
            "solver_residual_threshold_1e-7": ({"stationary": rt_side,
                                                "note": "side key: pbre_physics.solver_residual_threshold = 1e-7 (PyBullet's documented solverResidualThreshold default "
                                                        "[EXT-UNVERIFIED]); `value` is measured with the engine's default 0 = all 150 sweeps (DESIGN.md section 2)"} if rt_side else None),
            "k_fast_variant": {"steps_that_were_one_launch_since_reset": info[13], "vgprs_one_launch_kernel": info[14],
                               "steps_with_3_waves_per_simd_variant_since_reset": info[8], "vgprs_2_wave_variant": info[0], "vgprs_3_wave_variant": info[9],
                               "steps_with_the_pair_kernel_since_reset": info[10], "vgprs_pair_kernel": info[11],
                               "steps_with_tail_pairs_since_reset": info[15] if len(info) > 15 else None,
                               "note": "round 5: a step is ONE launch (k_fused: the complex envs' row waves + the simple envs' waves in one grid; PBRE_FUSED=0: the two kernels on two streams of rounds 1-4, "
                                       "where launch_step picks the 168-VGPR k_fast for steps in which the row waves would push waves of the 256-VGPR build into a second round, PBRE_FAST3); "
                                       "the simple envs' waves are the pair mapping (robot wave + object wave per 64 envs) for batches of up to 65536 envs per GPU (PBRE_PAIR); "
                                       "round 6: PBRE_TAIL_PAIR=1 steps the last chunks of a machine-filling batch -- as many as the row waves keep out of the first round -- as such pairs too (tail pairs; measured +1.3 % here, -34 % with few complex envs: off by default)"},
            "contact_histogram_rank0": contact_hist,
            "nan_inf_guard": {"bad_env_steps_since_create": info[12], "note": "env-steps whose state was not finite (pbre_kernel_info[12]); such envs are returned with done = 1 and restarted"},
            "shards": shards,
            "fresh_reset": clean(fresh) if fresh else None,
            "steady_synchronised_clocks": sync_clocks,
            "weak_scaling_128k_per_gpu": weak,
            "sharded_consumers_no_gather": no_gather,
            "closed_loop": closed_loop,      # N > 1: `value` overlaps the gather of step t with the kernels of step t + 1 (an open loop); this is the figure a policy that needs obs(t) gets
            "host_inclusive": host,      # SURVEY 8(d) literal: upload + kernels + download through the host-buffer entry point; `value` is device-resident stepping
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_note": traffic_note,
                         "kernel": dom_kernel, "kernel_ms": kern_ms, "kernel_ms_sampled_pairs_mean": kern_ms_sampled, "step_launch_pair_ms": head["_pair_ms"],
                         "kernel_ms_note": "HIP event pair around each timed region on the launch stream / steps, median of the six regions `ms_per_step` is the median of "
                                           "(one launch per step, no gaps: the launch's average duration); kernel_ms_sampled_pairs_mean = the library's event pairs around every 8th launch",
                         "algorithmic_bytes_per_launch": ALG_BYTES_PER_ENV_STEP * n_local,
                         "note": "path is fp32-VALU issue bound (%.0f FLOP per algorithmic byte against a machine balance of ~20 FLOP/B); the HBM "
                                 "fraction is small by construction, see valu" % (ALG_FLOP_PER_ENV_STEP / ALG_BYTES_PER_ENV_STEP)},
            "valu": {"achieved": ach_tf, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP32_VALU_PEAK_TFLOPS,
                     # round 6 (profiles/r06_ubench_valu.txt, tools/ubench/valu_rate.hip): a SIMD issues a wave64 v_fma_f32 / v_mul / v_add whose
                     # operands are VGPRs on distinct banks every 2.2 cycles from >= 2 waves (135-138 TF measured, the spec's 157.3 TF is 2.0),
                     # v_pk_fma_f32 every 4.2-4.4 (the same FLOP rate); an SGPR / constant operand, v_max, v_fmac or a repeated VGPR operand make it
                     # 4.1-4.4 cycles; ONE wave issues at most one VALU instruction per 4.45-5 cycles (8.25 when it depends on the previous one).
                     # Rounds 1-5 priced against "78.6 TF scalar-FMA peak" from a 64-thread-workgroup wall-clock benchmark of SGPR-operand FMAs.
                     "peak_measured_fma": 138.0, "frac_measured_fma": ach_tf / 138.0,
                     "issue_model": {"simd_cycles_per_wave_instr_best": 2.2, "simd_cycles_per_wave_instr_sgpr_operand_or_packed": 4.3,
                                     "lone_wave_cycles_per_instr_independent": 4.45, "lone_wave_cycles_per_instr_dependent": 8.25,
                                     "source": "profiles/r06_ubench_valu.txt"},
                     "flop_per_env_step": ALG_FLOP_PER_ENV_STEP, "sq_counters": sq, "vgprs_k_fast": info[0], "vgprs_k_fast_rc": info[6], "vgprs_k_step": info[1]},
        }
        if world == 1 and not args.no_other_configs:
            res["other_configs"] = early_other if early_other is not None else other_configs(torch, dev)
        if not args.no_cpu_baseline and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is informative; never lose the GPU line over it
                res["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
