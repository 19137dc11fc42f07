"""The C-ABI library loads and exports every symbol include/pbre.h declares; without a GPU it fails loudly
(no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pybullet_robot_envs import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pbre.h")).read()
    return sorted(set(re.findall(r"\b(pbre_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported_by_hip_library(hip_lib):
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(hip_lib, n), "libpbre.so does not export %s" % n


def test_emulation_library_exports_same_abi(emu_lib):
    for n in _declared():
        assert hasattr(emu_lib, n)


def test_config_struct_layout_matches(hip_lib):
    cfg = _capi.Config()
    assert hip_lib.pbre_default_config(C.byref(cfg), 0, 1) == 0
    assert cfg.task == 1 and cfg.max_steps == 1000 and cfg.num_controlled_joints == 7
    assert cfg.target_dist_min == 0.1 and cfg.act_scale == 0.05 and cfg.kp_act == 0.5 and cfg.kp_hold == 0.2
    assert cfg.h_table == 0.625 and cfg.phys.solver_iters == 150 and abs(cfg.phys.dt - 1 / 240) < 1e-15
    assert list(cfg.home[:9]) == [0.0, -0.54, 0.0, -2.6, -0.30, 2.0, 1.0, 0.02, 0.02]
    assert hip_lib.pbre_default_config(C.byref(cfg), 0, 0) == 0 and cfg.target_dist_min == 0.03
    assert hip_lib.pbre_default_config(C.byref(cfg), 5, 1) == _capi_err("ARG")


def _capi_err(name):
    return {"ARG": -1, "TABLE": -2, "DEVICE": -3, "UNSUPPORTED": -4}[name]


def test_bad_arguments_are_rejected(panda, emu_lib):
    with pytest.raises(RuntimeError, match="robot_table"):
        _capi.Engine(np.zeros(10), lib=emu_lib)
    with pytest.raises(RuntimeError, match="action_repeat out of range"):
        _capi.Engine(panda["table"], lib=emu_lib, action_repeat=1000)
    eng = _capi.Engine(panda["table"], lib=emu_lib, num_envs=2)
    with pytest.raises(ValueError):
        eng.step(np.zeros((3, 7), np.float32))


def test_no_cpu_fallback(panda, hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|hip"):
        _capi.Engine(panda["table"], lib=hip_lib, num_envs=4)
    with pytest.raises(RuntimeError, match="not found"):
        _capi.load(os.path.join(ROOT, "does_not_exist.so"))
