import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pybullet-robot-envs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The iCub's lane-per-env pipeline (pbre_lane.hip) is the engine's default from 16384 envs on; the tests run small batches, so they
    # switch it on explicitly (tests of the lane-group kernel set PBRE_ICUB_LANE=0 themselves, the default rule has a test of its own)
    os.environ.setdefault("PBRE_ICUB_LANE", "1")


@pytest.fixture(scope="session", autouse=True)
def built():
    """Oracle + host-emulation libraries (CPU only; the HIP library is built by __graft_entry__.build()).  Session-wide and
    first: the `make` subprocesses must be forked before any test initialises the HIP runtime in this process."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "host_emu")])
    return True


@pytest.fixture(scope="session")
def panda(built):
    from pybullet_robot_envs.model.table import panda_table, PANDA_SPHERES
    tbl, model = panda_table()
    return {"table": tbl, "model": model, "spheres": PANDA_SPHERES}


@pytest.fixture(scope="session")
def emu_lib(built):
    from pybullet_robot_envs import _capi
    return _capi.load(os.path.join(ROOT, "tests", "host_emu", "build", "libpbre_emu.so"))


@pytest.fixture(scope="session")
def hip_lib(built):
    from pybullet_robot_envs import _capi
    if not os.path.exists(_capi.LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "pybullet-robot-envs_amd", "csrc", "build.sh")])
    return _capi.load()
