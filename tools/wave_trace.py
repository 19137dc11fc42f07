"""What is the LONGEST row wave of each stationary step made of?  Needs the trace build of the joint-control step unit:
    tools/build_variant.sh wtrace "-DPBRE_WAVE_TRACE" pbre_step_2_false
    python tools/wave_trace.py [--envs 131072] [--steps 300] [--lib pybullet-robot-envs_amd/csrc/libpbre_wtrace.so]
Every row wave of k_fused records the ticks from the start of Core::step to the end of its sweeps and a bit mask of its rows / solver path
(csrc/pbre_panda.hpp: PBRE_WAVE_TRACE).  Per step (host-synchronised, one step at a time): the step's wall time, the number of row waves, and the
longest wave's record; then the steps grouped by what their longest wave was."""
import argparse, collections, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
import numpy as np
import torch
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table


def describe(bits):
    paths = [k for k in range(16) if (bits >> k) & 1]
    ro, ot, rt, lim = (bits >> 16) & 15, (bits >> 20) & 15, (bits >> 24) & 3, (bits >> 30) & 1
    names = {10: "switched-to-clamping", 11: "STARTED-OVER", 12: "clamp-free", 13: "robot-only chain", 14: "two zipped chains", 15: "per-slot loops"}
    return "%s | RO %d OT %d RT %d%s" % ("+".join(names.get(p, str(p)) for p in paths), bin(ro).count("1"), bin(ot).count("1"), bin(rt).count("1"), " LIMIT" if lim else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=131072)
    ap.add_argument("--preroll", type=int, default=1100)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--diag", type=int, default=0, help="after the pre-roll: 1 = the k_fast waves hold their slots asleep, 2 = busy with a dependent v_fma chain, "
                    "instead of stepping their envs (pbre_debug_wave_diag; what sharing a SIMD costs the row waves, DESIGN 5.3) -- use a few steps only")
    ap.add_argument("--lib", default=os.path.join(ROOT, "pybullet-robot-envs_amd", "csrc", "libpbre_wtrace.so"))
    a = ap.parse_args()
    lib = _capi.load(a.lib)
    lib.pbre_debug_wave_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
    lib.pbre_debug_wave_trace.restype = C.c_int
    dev = torch.device("cuda", 0)
    tbl, _ = panda_table()
    n = a.envs
    eng = _capi.Engine(tbl, task=_capi.TASK_PUSH, num_envs=n, seed=1234, obj_pose_rnd_std=0.05, tg_pose_rnd_std=0.2, flags=_capi.F_AUTO_RESET, lib=lib)
    eng.reset()
    st = eng.get_state()
    st[:, eng.x_off + 3] = np.random.default_rng(4321).integers(0, 1000, n).astype(np.float32)
    eng.set_state(st)
    stream = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    out = torch.zeros((n, eng.obs_dim + 2), device=dev)
    act = torch.empty((n, eng.act_dim), device=dev)
    for _ in range(a.preroll):
        act.uniform_(-1, 1, generator=gen)
        eng.step_device(act.data_ptr(), out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (3 * 16384))()
    lib.pbre_debug_wave_trace(buf, 16384, 1)
    if a.diag:
        assert lib.pbre_debug_wave_diag(C.c_int(a.diag)) == 0
        print("diag mode %d: the k_fast waves do not step their envs from here on" % a.diag)
    rec = []
    persist, prev_so = [0, 0], set()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(a.steps):
        act.uniform_(-1, 1, generator=gen)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            e0.record(stream)
            eng.step_device(act.data_ptr(), out.data_ptr(), stream.cuda_stream)
            e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        m = lib.pbre_debug_wave_trace(buf, 16384, 1)
        r = np.frombuffer(buf, dtype=np.uint64, count=3 * max(m, 0)).reshape(-1, 3).copy()
        so_set = set(int(x) for x in r[((r[:, 1] >> np.uint64(11)) & np.uint64(1)) != 0, 2]) if m > 0 else set()
        if k > 0:
            persist[0] += len(so_set & prev_so); persist[1] += len(so_set)
        prev_so = so_set
        if m <= 0:
            rec.append((ms, 0, 0, 0, 0, 0)); continue
        i = int(np.argmax(r[:, 0]))
        started_over = int(((r[:, 1] >> np.uint64(11)) & np.uint64(1)).sum())
        switched = int(((r[:, 1] >> np.uint64(10)) & np.uint64(1)).sum())
        rec.append((ms, m, int(r[i, 0]), int(r[i, 1]), started_over, switched))
    ms = np.array([x[0] for x in rec]); tk = np.array([x[2] for x in rec], float)
    ok = tk > 0
    # ticks -> us: least squares through the steps whose longest wave sets the step time (the upper half)
    hi = ok & (ms >= np.median(ms))
    scale = float((ms[hi] * 1e3).sum() / tk[hi].sum()) if hi.any() else 0.01
    print("envs %d, %d steps: step time (events around each host-synchronised step) median %.1f us, mean %.1f, p75 %.1f, max %.1f; row waves per step median %d; 1 tick ~ %.4f us"
          % (n, a.steps, np.median(ms) * 1e3, ms.mean() * 1e3, np.percentile(ms, 75) * 1e3, ms.max() * 1e3, int(np.median([x[1] for x in rec])), scale))
    groups = collections.defaultdict(list)
    for (m_, cnt, t, b, so, sw_) in rec:
        groups[describe(b)].append((m_ * 1e3, t * scale, t))
    print("steps grouped by their longest row wave (Core::step start .. end of sweeps):")
    for g, v in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        v = np.array(v)
        print("  %4d steps  step time mean %6.1f us (min %6.1f max %6.1f)  longest wave mean %6.1f us = %6.1f k ticks   %s" % (len(v), v[:, 0].mean(), v[:, 0].min(), v[:, 0].max(), v[:, 1].mean(), v[:, 2].mean() / 1e3, g))
    print("waves that started over: %d in all; %d of them (%.0f %%) step an env whose wave also started over in the step before" % (persist[1], persist[0], 100.0 * persist[0] / max(1, persist[1])))
    so = np.array([x[4] for x in rec])
    print("row waves that left the clamp-free stages for the clamping ones (bit 10), per step: mean %.2f" % np.mean([x[5] for x in rec]))
    print("steps with a wave that started over: %d of %d (their step time mean %.1f us; the others %.1f us)" % ((so > 0).sum(), len(so), ms[so > 0].mean() * 1e3 if (so > 0).any() else 0.0, ms[so == 0].mean() * 1e3))


if __name__ == "__main__":
    main()
