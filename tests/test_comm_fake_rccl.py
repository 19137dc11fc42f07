"""The context-owned exchanges of the sharded batch with world > 1 (include/pbre.h: pbre_comm_init, pbre_step_gather_device,
pbre_gather_wait, pbre_scatter_actions_device; csrc/pbre_comm_impl.hpp) executed on CPU: the lane emulation compiles the same
source on a host runtime, tests/fake_rccl supplies ncclSend / ncclRecv between the processes over shared memory.  Rank 0 compares the
stacked rows of every step with an unsharded engine's: bit for bit (env i on rank i // (N / G), results independent of G).
The same worker runs with two ranks on one GPU in tests/test_gpu_rccl.py."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(kind, world, total, steps, mode="closed", timeout=300):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_rccl")])
    with tempfile.TemporaryDirectory() as d:
        rdv = os.path.join(d, "uid")
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "comm_worker.py"), kind, str(r), str(world), rdv, str(total), str(steps), mode],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, env=env) for r in range(world)]
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=timeout)[0])
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and "COMM_OK rank %d of %d" % (r, world) in o, "rank %d failed:\n%s" % (r, o[-3000:])
    return outs


@pytest.mark.parametrize("world,mode", [(2, "closed"), (4, "closed"), (2, "open"), (8, "closed"), (8, "open")])      # (8: the driver's SCALE run is --gpus 8)
def test_context_owned_gather_and_scatter_world_gt_1(emu_lib, world, mode):
    run_world("emu", world, 8 * world, 8, mode)


def test_nccl_abi_declared_in_comm_impl_matches_rccl_header():
    """csrc/pbre_comm_impl.hpp declares the few RCCL types it binds instead of including <rccl/rccl.h> (no build-time dependency);
    where the header exists the declarations are checked against it."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h on this box")
    src = r'''
#include <rccl/rccl.h>
static_assert(sizeof(ncclUniqueId) == 128, "unique id");
static_assert((int)ncclSuccess == 0 && (int)ncclFloat == 7 && (int)ncclFloat32 == 7 && (int)ncclUint8 == 1, "enum values");
static_assert(sizeof(ncclComm_t) == sizeof(void*) && sizeof(ncclResult_t) == sizeof(int) && sizeof(ncclDataType_t) == sizeof(int), "sizes");
int main() { return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "chk.cpp")
        open(f, "w").write(src)
        subprocess.check_call(["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-fsyntax-only", f])
