"""Thin ctypes binding of libpbre.so (C-ABI in include/pbre.h).

This is the only place the Python package touches native code.  The library is the
HIP engine built in-tree by `__graft_entry__.build()` (csrc/build.sh); there is NO CPU
fallback: if the shared object is missing or no GPU is usable, loading/creating fails
loudly with RuntimeError.
"""
import ctypes as C
import os
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBRE_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libpbre.so")   # PBRE_LIB: build-variant A/B runs

STATE_FLOATS = 48
ROBOT_PANDA, ROBOT_ICUB, ROBOT_ICUB_HANDS, ROBOT_PANDA_ARM = 0, 1, 2, 3
TASK_REACH, TASK_PUSH, TASK_PUSH_GOAL = 0, 1, 2
SHAPE_BOX, SHAPE_SPHERE, SHAPE_CYLINDER, SHAPE_HULL = 0, 1, 2, 3
F_NO_OBJECT, F_AUTO_RESET, F_FORCE_GENERAL, F_COMPLEX_ROWS, F_COMPLEX_LANES, F_SEQ_MOTORS, F_SEQ_OBJECT = 1, 2, 4, 8, 16, 32, 64


class Physics(C.Structure):
    _fields_ = [("dt", C.c_double), ("gravity_z", C.c_double), ("solver_iters", C.c_int32),
                ("erp", C.c_double), ("linear_slop", C.c_double), ("contact_margin", C.c_double),
                ("lin_damping", C.c_double), ("ang_damping", C.c_double), ("max_coord_vel", C.c_double),
                ("max_motor_impulse", C.c_double), ("limit_max_impulse", C.c_double),
                ("table_c", C.c_double * 3), ("table_h", C.c_double * 3), ("table_mu", C.c_double),
                ("ground_z", C.c_double), ("obj_h", C.c_double * 3), ("obj_mass", C.c_double),
                ("obj_inertia", C.c_double * 3), ("obj_mu", C.c_double), ("implicit_joint_damping", C.c_int32), ("obj_shape", C.c_int32),
                ("solver_residual_threshold", C.c_double)]


class Config(C.Structure):
    _fields_ = [("robot", C.c_int32), ("task", C.c_int32), ("num_envs", C.c_int32), ("device_id", C.c_int32),
                ("env_id_base", C.c_uint64), ("seed", C.c_uint64),
                ("use_ik", C.c_int32), ("num_controlled_joints", C.c_int32), ("action_repeat", C.c_int32),
                ("max_steps", C.c_int32), ("flags", C.c_int32),
                ("obj_pose_rnd_std", C.c_double), ("tg_pose_rnd_std", C.c_double),
                ("target_dist_min", C.c_double), ("act_scale", C.c_double),
                ("kp_act", C.c_double), ("kd_act", C.c_double), ("kp_hold", C.c_double), ("kd_hold", C.c_double),
                ("ws_lim", C.c_double * 2 * 3), ("h_table", C.c_double), ("home", C.c_double * 64),
                ("phys", Physics),
                ("ik_damping", C.c_double), ("ik_residual", C.c_double), ("ik_max_iters", C.c_int32),
                ("home_hand_pose", C.c_double * 6), ("robot_ws", C.c_double * 2 * 3),
                ("control_orientation", C.c_int32), ("reward_type", C.c_int32), ("num_joints_ctrl", C.c_int32),
                ("act_dof", C.c_int32 * 64), ("ik_pos_scale", C.c_double), ("ik_rot_scale", C.c_double),
                ("eu_lim", C.c_double * 2 * 3), ("ik_link_offset", C.c_double * 3), ("ik_absolute", C.c_int32), ("robot_level", C.c_int32),
                ("robot_table", C.c_void_p), ("robot_table_len", C.c_size_t)]


_LIB = None


def load(path=None):
    """Load libpbre.so (once).  Raises RuntimeError when the HIP extension has not been built."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("libpbre.so not found at %s -- build the HIP engine first "
                           "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % p)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's).  If libpbre.so
    # pulled in the system copy first and torch loaded its bundled copy afterwards, the second runtime would find no
    # device.  Importing torch first (when it is installed) makes libpbre.so bind to the runtime torch uses, which is also
    # what sharing streams and device pointers with torch tensors requires.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(p)
    lib.pbre_last_error.restype = C.c_char_p
    lib.pbre_last_error.argtypes = [C.c_void_p]
    lib.pbre_destroy.restype = None
    for name in ("pbre_default_config", "pbre_create", "pbre_dims", "pbre_reset", "pbre_step", "pbre_step_device",
                 "pbre_sync", "pbre_get_state", "pbre_set_state", "pbre_observe", "pbre_settle", "pbre_obs_limits",
                 "pbre_timing", "pbre_kernel_info", "pbre_set_physics", "pbre_get_physics", "pbre_state_floats", "pbre_set_motors", "pbre_apply_action", "pbre_get_motor_state", "pbre_set_motor_state",
                 "pbre_get_state_cols", "pbre_set_physics_per_env", "pbre_reset_snapshot", "pbre_get_sweeps", "pbre_set_object_hull",
                 "pbre_comm_probe", "pbre_comm_unique_id", "pbre_comm_init", "pbre_step_gather_device", "pbre_gather_wait", "pbre_comm_info", "pbre_scatter_actions_device"):
        getattr(lib, name).restype = C.c_int
    lib.pbre_comm_last_error.restype = C.c_char_p
    lib.pbre_comm_last_error.argtypes = [C.c_void_p]
    lib.pbre_host_alloc.restype = C.c_void_p
    lib.pbre_host_alloc.argtypes = [C.c_size_t]
    lib.pbre_host_free.restype = None
    lib.pbre_host_free.argtypes = [C.c_void_p]
    if path is None:
        _LIB = lib
    return lib


def _set_rccl_lib(path=None):
    """PBRE_RCCL_LIB for csrc/pbre_comm.hip's dlopen: an explicit path, else (unless already set) the librccl.so torch bundles -- the
    copy torch.distributed's "nccl" backend has loaded or will load -- so that one process never holds two RCCLs."""
    if path:
        os.environ["PBRE_RCCL_LIB"] = path
    elif not os.environ.get("PBRE_RCCL_LIB"):
        try:
            import torch
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(cand):
                os.environ["PBRE_RCCL_LIB"] = cand
        except Exception:
            pass


def _fp(a):
    return a.ctypes.data_as(C.c_void_p)


STREAM_LEGACY = 1      # PBRE_STREAM_LEGACY (include/pbre.h) == hipStreamLegacy


def torch_stream(device=None):
    """hipStream_t of torch's current stream on `device` as the int pbre_step_device takes.  Torch's default stream is HIP's
    legacy null stream, whose handle is 0; 0 would select the engine's own non-blocking stream (which is not ordered against
    the null stream), so it is mapped to PBRE_STREAM_LEGACY."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream or STREAM_LEGACY


class Engine:
    """One pbre_ctx: `num_envs` environments on one GPU."""

    def __init__(self, robot_table, task=TASK_PUSH, num_envs=1, lib=None, robot=ROBOT_PANDA, **overrides):
        self.lib = lib or load()
        self.cfg = Config()
        rc = self.lib.pbre_default_config(C.byref(self.cfg), C.c_int32(robot), C.c_int32(task))
        if rc != 0:
            raise RuntimeError("pbre_default_config failed: %d" % rc)
        self.cfg.num_envs = int(num_envs)
        phys = overrides.pop("phys", None)
        for k, v in overrides.items():
            if not hasattr(self.cfg, k):
                raise TypeError("unknown pbre_config field %r" % k)
            if isinstance(v, (list, tuple, np.ndarray)):          # array fields: home, act_dof, eu_lim, ...
                arr = getattr(self.cfg, k)
                flat = np.asarray(v).reshape(-1)
                if len(arr) and hasattr(arr[0], "__len__"):
                    cols = len(arr[0])
                    for i, x in enumerate(flat):
                        arr[i // cols][i % cols] = x
                else:
                    for i, x in enumerate(flat):
                        arr[i] = x
            else:
                setattr(self.cfg, k, v)
        hull = None
        if phys:
            phys = dict(phys)
            hull = phys.pop("obj_hull", None)           # a convex-hull object (model/objects.py: hull_physics): set right after pbre_create
            if hull is not None:
                phys["obj_shape"] = SHAPE_BOX           # (pbre_create takes a primitive; obj_h = the hull's bounding box until the hull replaces it)
            for k, v in phys.items():
                if not hasattr(self.cfg.phys, k):
                    raise TypeError("unknown pbre_physics field %r" % k)
                if isinstance(v, (list, tuple, np.ndarray)):
                    for i, x in enumerate(v):
                        getattr(self.cfg.phys, k)[i] = x
                else:
                    setattr(self.cfg.phys, k, v)
        if os.environ.get("PBRE_SEQ_OBJECT") == "1":      # A/B knob (tools/, bench side runs): PBRE_F_SEQ_OBJECT on every engine of the process
            self.cfg.flags |= F_SEQ_OBJECT
        self._table = np.ascontiguousarray(robot_table, dtype=np.float64)
        self.cfg.robot_table = self._table.ctypes.data
        self.cfg.robot_table_len = self._table.size
        self._ctx = C.c_void_p()
        rc = self.lib.pbre_create(C.byref(self.cfg), C.byref(self._ctx))
        if rc != 0:
            raise RuntimeError("pbre_create failed (%d): %s" % (rc, self.lib.pbre_last_error(None).decode()))
        od, ad, n = C.c_int32(), C.c_int32(), C.c_int32()
        self._chk(self.lib.pbre_dims(self._ctx, C.byref(od), C.byref(ad), C.byref(n)))
        self.obs_dim, self.act_dim, self.num_envs = od.value, ad.value, n.value
        self.state_floats = int(self.lib.pbre_state_floats(self._ctx))
        # state record layout (include/pbre.h): Q[W] | V[W] | X[16]; object pose at Q[ndof..ndof+7)
        self.ndof = int(self._table[3])
        # where the object's pose / twist sits inside the Q / V records: behind the DoF LANES of the kernel shape the engine picked (9, 20, 32,
        # 60; include/pbre.h) -- equal to ndof for every model but the soft-pinned floating-base iCub (26 DoF on the 32-lane shape)
        self.obj_off = 9 if self.ndof <= 9 else (20 if self.ndof <= 20 else (32 if self.ndof <= 32 else 60))
        self.v_off = (self.state_floats - 16) // 2
        self.x_off = self.state_floats - 16
        if hull is not None:
            self.set_object_hull(hull)
        # page-locked staging buffers of the host path (pbre_step DMAs straight from / into them): actions, and two row
        # buffers used alternately so that the arrays step(copy=False) returned stay valid for one more step
        self._pinned = []
        self._act = self._pinned_array((self.num_envs, self.act_dim))
        self._outs = [self._pinned_array((self.num_envs, self.obs_dim + 2)) for _ in range(2)]
        self._flip = 0

    def _pinned_array(self, shape):
        nbytes = int(np.prod(shape)) * 4
        p = self.lib.pbre_host_alloc(C.c_size_t(nbytes))
        if not p:
            raise RuntimeError("pbre_host_alloc(%d) failed" % nbytes)
        self._pinned.append(p)
        a = np.ctypeslib.as_array((C.c_float * (nbytes // 4)).from_address(p)).reshape(shape)
        a[...] = 0
        return a

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libpbre error %d: %s" % (rc, self.lib.pbre_last_error(self._ctx).decode()))

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.pbre_destroy(self._ctx)
            self._ctx = None
            self._act = self._outs = None
            self._async = None
            for p in self._pinned:
                self.lib.pbre_host_free(C.c_void_p(p))
            self._pinned = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def obs_limits(self):
        lo = np.zeros(self.obs_dim, np.float32)
        hi = np.zeros(self.obs_dim, np.float32)
        self._chk(self.lib.pbre_obs_limits(self._ctx, _fp(lo), _fp(hi)))
        return lo, hi

    def reset(self, mask=None):
        obs = np.zeros((self.num_envs, self.obs_dim), np.float32)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            assert m.shape == (self.num_envs,)
        self._chk(self.lib.pbre_reset(self._ctx, _fp(m) if m is not None else None, _fp(obs)))
        return obs

    def reset_snapshot(self, mask):
        """Fast reset of the masked envs from the settled snapshot of the last full reset (pbre_reset_snapshot): one kernel."""
        obs = np.zeros((self.num_envs, self.obs_dim), np.float32)
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        assert m.shape == (self.num_envs,)
        self._chk(self.lib.pbre_reset_snapshot(self._ctx, _fp(m), _fp(obs)))
        return obs

    def step(self, actions, copy=True):
        """Host-buffer step: (raw obs [N, obs_dim], reward [N], done [N]) float32.  copy=False returns views of a page-locked
        row buffer that the step after next overwrites (two buffers alternate) -- no 18 MB of copies per 131072-env step."""
        a = np.asarray(actions)
        if a.shape != (self.num_envs, self.act_dim):
            raise ValueError("actions must have shape (%d, %d), got %r" % (self.num_envs, self.act_dim, a.shape))
        np.copyto(self._act, a, casting="unsafe")
        o = self._outs[self._flip]
        self._flip ^= 1
        self._chk(self.lib.pbre_step(self._ctx, _fp(self._act), _fp(o)))
        if copy:
            return o[:, :self.obs_dim].copy(), o[:, self.obs_dim].copy(), o[:, self.obs_dim + 1].copy()
        return o[:, :self.obs_dim], o[:, self.obs_dim], o[:, self.obs_dim + 1]

    def step_async(self, actions):
        """Pipelined host-buffer step (pbre_step_async): enqueue upload + step + download and return; `step_wait()` returns the rows of the
        oldest step in flight.  At most two in flight -- the open loop  step_async(a0); for t: step_async(a[t]); rows = step_wait()  overlaps
        the download of step t - 1 with the kernels of step t.  Uses three page-locked (actions, rows) slots of its own."""
        if getattr(self, "_async", None) is None:
            self._async = {"act": [self._pinned_array((self.num_envs, self.act_dim)) for _ in range(3)],
                           "out": [self._pinned_array((self.num_envs, self.obs_dim + 2)) for _ in range(3)], "issued": 0, "waited": 0}
        A = self._async
        k = A["issued"] % 3          # (three slots: the views step_wait() returned for step t - 1 stay valid while steps t and t + 1 are in flight)
        np.copyto(A["act"][k], np.asarray(actions).reshape(self.num_envs, self.act_dim), casting="unsafe")
        self._chk(self.lib.pbre_step_async(self._ctx, _fp(A["act"][k]), _fp(A["out"][k])))
        A["issued"] += 1

    def step_pipelined(self, actions):
        """The open-loop pipeline in the order that keeps the host off the critical path: copy `actions` into the next page-locked slot (while
        the GPU works), THEN wait for the rows of the step before last, THEN enqueue this step.  Returns (obs, reward, done) views of the step
        two calls back, None for the first two calls; drain with step_wait() twice at the end."""
        if getattr(self, "_async", None) is None:
            self.step_async(actions)                      # (first call: creates the slots)
            return None
        A = self._async
        k = A["issued"] % 3
        t0 = time.perf_counter()
        np.copyto(A["act"][k], np.asarray(actions).reshape(self.num_envs, self.act_dim), casting="unsafe")
        t1 = time.perf_counter()
        rows = self.step_wait() if A["issued"] - A["waited"] >= 2 else None
        t2 = time.perf_counter()
        self._chk(self.lib.pbre_step_async(self._ctx, _fp(A["act"][k]), _fp(A["out"][k])))
        A["issued"] += 1
        ph = A.setdefault("phase_s", [0.0, 0.0, 0.0, 0])      # host seconds spent copying the actions / waiting for rows / enqueueing, calls
        ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += time.perf_counter() - t2; ph[3] += 1
        return rows

    def step_wait(self, copy=False):
        """(raw obs, reward, done) of the oldest step_async not yet waited for: views of a page-locked row buffer that is overwritten by
        the third step_async from now (copy=True: copies)"""
        A = self._async
        self._chk(self.lib.pbre_step_wait(self._ctx))
        o = A["out"][A["waited"] % 3]
        A["waited"] += 1
        if copy:
            return o[:, :self.obs_dim].copy(), o[:, self.obs_dim].copy(), o[:, self.obs_dim + 1].copy()
        return o[:, :self.obs_dim], o[:, self.obs_dim], o[:, self.obs_dim + 1]

    def step_device(self, d_actions_ptr, d_out_ptr, stream=None):
        """Asynchronous step on device-resident buffers.  `stream`: a hipStream_t handle as an int -- pass
        `torch_stream(device)` so that the step is ordered with the torch work that produced the actions and consumes the rows;
        None = the engine's own non-blocking stream (NOT ordered against torch streams: the caller orders inputs / outputs)."""
        self._chk(self.lib.pbre_step_device(self._ctx, C.c_void_p(d_actions_ptr), C.c_void_p(d_out_ptr),
                                            C.c_void_p(stream or 0)))

    # ---- the sharded batch's gather, owned by the context (include/pbre.h: pbre_comm_*; csrc/pbre_comm.hip) ----
    @staticmethod
    def comm_probe(lib=None, rccl_lib=None):
        """any rank: raises unless the RCCL library loads and exports every entry point the exchanges use (no ncclGetUniqueId: that starts
        a bootstrap thread and a listening socket, which only rank 0 needs)"""
        _set_rccl_lib(rccl_lib)
        lib = lib or load()
        rc = lib.pbre_comm_probe()
        if rc != 0:
            raise RuntimeError("pbre_comm_probe failed (%d): %s" % (rc, lib.pbre_comm_last_error(None).decode()))

    @staticmethod
    def comm_unique_id(lib=None, rccl_lib=None):
        """rank 0: the 128-byte id every rank's comm_init needs.  rccl_lib: path of the RCCL library to dlopen (default: torch's bundled
        copy when torch is importable, so that a process which also uses torch.distributed holds ONE RCCL)."""
        _set_rccl_lib(rccl_lib)
        lib = lib or load()
        buf = (C.c_ubyte * 128)()
        rc = lib.pbre_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("pbre_comm_unique_id failed (%d): %s" % (rc, lib.pbre_comm_last_error(None).decode()))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world, rccl_lib=None):
        _set_rccl_lib(rccl_lib)
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        rc = self.lib.pbre_comm_init(self._ctx, buf, C.c_int32(rank), C.c_int32(world))
        if rc != 0:
            raise RuntimeError("pbre_comm_init failed (%d): %s" % (rc, self.lib.pbre_comm_last_error(None).decode()))

    def comm_info(self):
        info = (C.c_int32 * 5)()
        self._chk_comm(self.lib.pbre_comm_info(self._ctx, info, C.c_int32(5)))
        return {"ranks_seen": info[0], "rank": info[1], "rccl_version_code": info[2], "exchanges": info[3], "action_scatters": info[4]}

    def scatter_actions_device(self, d_actions_all_ptr, d_actions_local_ptr, stream=None):
        """The way back of a closed loop: rank 0's [G * num_envs, act_dim] actions -> every rank's [num_envs, act_dim] slice, one grouped
        RCCL exchange in `stream` order (pbre_scatter_actions_device)."""
        self._chk_comm(self.lib.pbre_scatter_actions_device(self._ctx, C.c_void_p(d_actions_all_ptr or 0), C.c_void_p(d_actions_local_ptr), C.c_void_p(stream or 0)))

    def step_gather_device(self, d_actions_ptr, d_rows_local_ptr, d_rows_all_ptr, stream=None):
        """pbre_step_device + the one grouped RCCL exchange of the step's rows into rank 0's stacked buffer, all enqueued from C
        (asynchronous; alternate between two buffer pairs, gather_wait before reading)."""
        self._chk_comm(self.lib.pbre_step_gather_device(self._ctx, C.c_void_p(d_actions_ptr), C.c_void_p(d_rows_local_ptr),
                                                        C.c_void_p(d_rows_all_ptr or 0), C.c_void_p(stream or 0)))

    def gather_wait(self, stream=None, host=False):
        self._chk_comm(self.lib.pbre_gather_wait(self._ctx, C.c_void_p(stream or 0), C.c_int32(1 if host else 0)))

    def _chk_comm(self, rc):
        if rc != 0:
            raise RuntimeError("libpbre communicator error %d: %s" % (rc, self.lib.pbre_comm_last_error(self._ctx).decode()))

    def sync(self):
        self._chk(self.lib.pbre_sync(self._ctx))

    def get_state(self):
        s = np.zeros((self.num_envs, self.state_floats), np.float32)
        self._chk(self.lib.pbre_get_state(self._ctx, _fp(s)))
        return s

    def get_state_cols(self, first, count=1):
        """[N, count] float32: `count` consecutive floats of every state record from float `first` (no whole-batch download)."""
        s = np.zeros((self.num_envs, count), np.float32)
        self._chk(self.lib.pbre_get_state_cols(self._ctx, C.c_int32(first), C.c_int32(count), _fp(s)))
        return s

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float32)
        assert s.shape == (self.num_envs, self.state_floats)
        self._chk(self.lib.pbre_set_state(self._ctx, _fp(s)))

    def observe(self):
        obs = np.zeros((self.num_envs, self.obs_dim), np.float32)
        self._chk(self.lib.pbre_observe(self._ctx, _fp(obs)))
        return obs

    def _motor_w(self):
        return (self.state_floats - 16) // 2

    def get_motor_state(self):
        """Robot-level engines: [N, 4, W] target | positionGain | force scale | max velocity of every DoF lane (W = 128 / 32)."""
        m = np.zeros((self.num_envs, 4, self._motor_w()), np.float32)
        self._chk(self.lib.pbre_get_motor_state(self._ctx, _fp(m)))
        return m

    def set_motor_state(self, m):
        m = np.ascontiguousarray(m, dtype=np.float32)
        if m.shape != (self.num_envs, 4, self._motor_w()):
            raise ValueError("motor state must be [%d, 4, %d]" % (self.num_envs, self._motor_w()))
        self._chk(self.lib.pbre_set_motor_state(self._ctx, _fp(m)))

    def apply_action(self, actions, max_vel=-1.0):
        """Robot-level engines: the command half of apply_action (IK / clipped joint targets -> persistent motors), no simulation step."""
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.num_envs, self.act_dim):
            raise ValueError("actions must be [%d, %d]" % (self.num_envs, self.act_dim))
        self._chk(self.lib.pbre_apply_action(self._ctx, _fp(a), C.c_double(max_vel)))

    def set_motors(self, dofs, targets, kp, max_force=0.0, mask=None, max_vel=0.0):
        """Robot-level engines: persistent POSITION_CONTROL command of the given DoF (same targets in every selected env)."""
        d = np.ascontiguousarray(dofs, dtype=np.int32)
        t = np.ascontiguousarray(targets, dtype=np.float32)
        if d.shape != t.shape or d.ndim != 1:
            raise ValueError("dofs and targets must be 1-D and of equal length")
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        if m is not None and m.shape != (self.num_envs,):
            raise ValueError("mask must have num_envs entries")
        self._chk(self.lib.pbre_set_motors(self._ctx, C.c_int32(d.size), _fp(d), _fp(t), C.c_double(kp), C.c_double(max_force),
                                           C.c_double(max_vel), None if m is None else _fp(m)))

    def settle(self, n, flags=0):
        self._chk(self.lib.pbre_settle(self._ctx, C.c_int32(n), C.c_int32(flags)))

    def get_physics(self):
        ph = Physics()
        self._chk(self.lib.pbre_get_physics(self._ctx, C.byref(ph)))
        return ph

    def set_physics(self, **fields):
        """Update batch-uniform physics constants, e.g. set_physics(obj_mass=0.2, obj_mu=0.8, lin_damping=0.1)."""
        ph = self.get_physics()
        fields = dict(fields)
        hull = fields.pop("obj_hull", None)               # (model/objects.py: hull_physics) the primitive fields first, then the hull
        if hull is not None:
            fields["obj_shape"] = SHAPE_BOX
        for k, v in fields.items():
            if not hasattr(ph, k):
                raise TypeError("unknown pbre_physics field %r" % k)
            if isinstance(v, (list, tuple)):
                for i, x in enumerate(v):
                    getattr(ph, k)[i] = x
            else:
                setattr(ph, k, v)
        self._chk(self.lib.pbre_set_physics(self._ctx, C.byref(ph)))
        if hull is not None:
            self.set_object_hull(hull)

    def set_object_hull(self, verts):
        """The object as the convex hull of `verts` ([n, 3], 4 <= n <= 32, object frame, origin = centre of mass): include/pbre.h
        pbre_set_object_hull.  Mass, inertia and friction stay what set_physics / the constructor's `phys` said."""
        v = np.ascontiguousarray(verts, dtype=np.float64)
        if v.ndim != 2 or v.shape[1] != 3:
            raise ValueError("verts must be [n, 3]")
        self._chk(self.lib.pbre_set_object_hull(self._ctx, v.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(len(v))))

    def get_sweeps(self):
        """[N] int32: the sweeps every env's solver ran in the last simulation step -- only with set_physics(solver_residual_threshold=...)
        > 0 (PyBullet's solverResidualThreshold; the reference leaves it at PyBullet's default, panda_push_gym_env.py:122)."""
        sw = np.empty(self.num_envs, np.int32)
        self._chk(self.lib.pbre_get_sweeps(self._ctx, sw.ctypes.data_as(C.POINTER(C.c_int32))))
        return sw

    def set_physics_per_env(self, obj_mass=None, obj_mu=None, obj_lin_damping=None, mask=None, robot_lin_damping=None):
        """Per-env object mass / lateral friction / linear damping and robot link damping ([N] arrays or None = unchanged): domain
        randomisation of the Panda task envs (reference change_physics_params, called per env and episode by the Dyn-Rand training)."""
        def arr(x):
            if x is None:
                return None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, np.float32), (self.num_envs,)))
            return a
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        a, b, c, d = arr(obj_mass), arr(obj_mu), arr(obj_lin_damping), arr(robot_lin_damping)
        self._chk(self.lib.pbre_set_physics_per_env(self._ctx, None if m is None else _fp(m), None if a is None else _fp(a),
                                                    None if b is None else _fp(b), None if c is None else _fp(c), None if d is None else _fp(d)))

    def timing(self):
        ms = (C.c_double * 4)()
        self._chk(self.lib.pbre_timing(self._ctx, ms, C.c_int32(4)))
        return list(ms)

    def kernel_info(self):
        info = (C.c_int32 * 16)()
        self._chk(self.lib.pbre_kernel_info(self._ctx, info, C.c_int32(16)))
        return list(info)


class MultiEngine(object):
    """`num_envs` environments sharded over several GPUs of one node from ONE process (the Gym classes' `devices=[...]` kwarg, SURVEY
    8b): one pbre_ctx per device, env i on device i // (N / G), RNG streams keyed by the global env id (so results are bitwise those
    of a single engine), every call fanned out to the shards from one host thread per device (ctypes releases the GIL during the
    C call).  Same methods as `Engine`; the device-pointer entry point is per shard (`shards[k].step_device`).  For one process per
    GPU under torch.distributed use `sharding.ShardedEngine` instead."""

    def __init__(self, robot_table, devices, num_envs=1, env_id_base=0, **kw):
        from concurrent.futures import ThreadPoolExecutor
        devices = [int(d) for d in devices]
        g = len(devices)
        if g < 1 or num_envs % g != 0:
            raise ValueError("num_envs (%d) must be divisible by the number of devices (%d)" % (num_envs, g))
        self.devices, self.n_shard = devices, num_envs // g
        kw.pop("device_id", None)
        self.shards = [Engine(robot_table, num_envs=self.n_shard, device_id=d, env_id_base=int(env_id_base) + k * self.n_shard, **kw)
                       for k, d in enumerate(devices)]
        e0 = self.shards[0]
        self.obs_dim, self.act_dim, self.state_floats = e0.obs_dim, e0.act_dim, e0.state_floats
        self.ndof, self.v_off, self.x_off, self.cfg, self.lib = e0.ndof, e0.v_off, e0.x_off, e0.cfg, e0.lib
        self.obj_off = e0.obj_off
        self.num_envs = int(num_envs)
        self._pool = ThreadPoolExecutor(max_workers=g)

    def _sl(self, k):
        return slice(k * self.n_shard, (k + 1) * self.n_shard)

    def _map(self, fn):
        return list(self._pool.map(fn, range(len(self.shards))))

    def _cat(self, parts):
        return np.concatenate(parts, axis=0)

    def close(self):
        self._pool.shutdown(wait=True)         # shard calls still in flight finish before their contexts are destroyed
        for e in self.shards:
            e.close()

    def obs_limits(self):
        return self.shards[0].obs_limits()

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        return self._cat(self._map(lambda k: self.shards[k].reset(None if m is None else m[self._sl(k)])))

    def step(self, actions, copy=True):
        a = np.asarray(actions)
        if a.shape != (self.num_envs, self.act_dim):
            raise ValueError("actions must have shape (%d, %d), got %r" % (self.num_envs, self.act_dim, a.shape))
        r = self._map(lambda k: self.shards[k].step(a[self._sl(k)], copy=False))
        return tuple(self._cat([x[i] for x in r]) for i in range(3))

    def reset_snapshot(self, mask):
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        return self._cat(self._map(lambda k: self.shards[k].reset_snapshot(m[self._sl(k)])))

    def step_device(self, *a, **kw):
        raise RuntimeError("step_device is per device: use MultiEngine.shards[k].step_device with buffers on that shard's GPU")

    def sync(self):
        self._map(lambda k: self.shards[k].sync())

    def observe(self):
        return self._cat(self._map(lambda k: self.shards[k].observe()))

    def get_state(self):
        return self._cat(self._map(lambda k: self.shards[k].get_state()))

    def get_state_cols(self, first, count=1):
        return self._cat(self._map(lambda k: self.shards[k].get_state_cols(first, count)))

    def set_state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float32)
        assert s.shape == (self.num_envs, self.state_floats)
        self._map(lambda k: self.shards[k].set_state(s[self._sl(k)]))

    def settle(self, n, flags=0):
        self._map(lambda k: self.shards[k].settle(n, flags))

    def get_physics(self):
        return self.shards[0].get_physics()

    def set_physics(self, **fields):
        for e in self.shards:
            e.set_physics(**fields)

    def set_object_hull(self, verts):
        for e in self.shards:
            e.set_object_hull(verts)

    def get_sweeps(self):
        return self._cat(self._map(lambda k: self.shards[k].get_sweeps()))

    def set_physics_per_env(self, obj_mass=None, obj_mu=None, obj_lin_damping=None, mask=None, robot_lin_damping=None):
        def part(x, k):
            return None if x is None else np.broadcast_to(np.asarray(x, np.float32), (self.num_envs,))[self._sl(k)]
        for k, e in enumerate(self.shards):
            e.set_physics_per_env(part(obj_mass, k), part(obj_mu, k), part(obj_lin_damping, k), None if mask is None else np.asarray(mask)[self._sl(k)],
                                  part(robot_lin_damping, k))

    def get_motor_state(self):
        return self._cat(self._map(lambda k: self.shards[k].get_motor_state()))

    def set_motor_state(self, m):
        m = np.ascontiguousarray(m, dtype=np.float32)
        self._map(lambda k: self.shards[k].set_motor_state(m[self._sl(k)]))

    def apply_action(self, actions, max_vel=-1.0):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        self._map(lambda k: self.shards[k].apply_action(a[self._sl(k)], max_vel))

    def set_motors(self, dofs, targets, kp, max_force=0.0, mask=None, max_vel=0.0):
        for k, e in enumerate(self.shards):
            e.set_motors(dofs, targets, kp, max_force, None if mask is None else np.asarray(mask)[self._sl(k)], max_vel)

    def timing(self):
        t = [e.timing() for e in self.shards]
        return [max(x[i] for x in t) for i in range(4)]

    def kernel_info(self):
        infos = [e.kernel_info() for e in self.shards]
        out = list(infos[0])
        for i in (3, 4, 5, 7):
            out[i] = sum(x[i] for x in infos)
        return out


def make_engine(robot_table, devices=None, **kw):
    """An `Engine` on one GPU, or a `MultiEngine` over `devices` (a list of HIP device ordinals) when more than one is given."""
    if devices is not None and len(devices) > 1:
        return MultiEngine(robot_table, devices, **kw)
    if devices is not None and len(devices) == 1:
        kw["device_id"] = int(devices[0])
    return Engine(robot_table, **kw)
