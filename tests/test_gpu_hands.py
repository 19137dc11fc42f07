"""iCub with hands on the MI355X (pbre_hands.hip: Shape128 / DevLanes128) vs the fp64 oracle, through the C-ABI."""
import numpy as np
import pytest

import parity
from pybullet_robot_envs import _capi

pytestmark = pytest.mark.gpu


def test_hands_joint_control_and_finger_commands(hip_lib):
    parity.check_hands(_capi.Engine, hip_lib, "r", 0, n=3, steps=4)


def test_hands_ik_control(hip_lib):
    parity.check_hands(_capi.Engine, hip_lib, "l", 1, n=2, steps=3)


def test_hands_fingertip_contacts(hip_lib):
    parity.check_hands_contacts(_capi.Engine, hip_lib, "r")
    parity.check_hands_contacts(_capi.Engine, hip_lib, "l")


def test_hands_device_matches_lane_emulation(hip_lib, emu_lib):
    """Same fp32 algorithm and summation order on both sides: the device and its CPU lane emulation agree to rounding of the
    transcendental functions after a reset (203 steps) and a grasp sequence."""
    from pybullet_robot_envs.model.table import GRASP_POS
    outs = []
    for lib in (hip_lib, emu_lib):
        eng, ora, info = parity.make_hands_pair(_capi.Engine, lib, 1, "r", 0)
        eng.reset()
        eng.set_motors(info["fingers"], GRASP_POS, 0.1, 10.0)
        a = np.asarray(info["home"], np.float32)[info["controlled"]][None, :]
        for _ in range(5):
            ob, rw, dn = eng.step(a)
        outs.append((eng.get_state(), ob))
    assert np.abs(outs[0][0] - outs[1][0]).max() < 1e-4
    assert np.abs(outs[0][1] - outs[1][1]).max() < 1e-3


def test_hands_batch_ragged_and_masked_reset(hip_lib):
    n = 67                                              # not a multiple of the 4 envs of a block
    eng, ora, info = parity.make_hands_pair(_capi.Engine, hip_lib, n, "r", 0, obj_std=0.05)
    obs = eng.reset()
    assert np.isfinite(obs).all()
    s0 = eng.get_state()
    assert np.abs(s0[:, :60] - s0[0, :60]).max() < 1e-5          # the robot settles identically; only the object pose is sampled
    assert np.ptp(s0[:, 60]) > 0.01
    a = np.tile(np.asarray(info["home"], np.float32)[info["controlled"]], (n, 1))
    a[:, 3] += 0.3
    for _ in range(10):
        ob, rw, dn = eng.step(a)
    s1 = eng.get_state()
    mask = np.zeros(n, np.uint8); mask[5] = 1; mask[66] = 1
    eng.reset(mask)
    s2 = eng.get_state()
    keep = mask == 0
    assert np.array_equal(s1[keep], s2[keep])
    assert np.abs(s2[5, :60] - s0[5, :60]).max() < 1e-5 and s2[5, eng.x_off + 5] == 1
    # the reset envs' motors were re-initialised (home targets), the others keep the commanded shoulder angle
    for _ in range(30):
        ob, rw, dn = eng.step(a)
    assert np.isfinite(eng.get_state()).all()


def test_motor_state_roundtrip_completes_the_checkpoint(hip_lib):
    """state record + motor record = the whole simulator state: restoring both reproduces the following steps bit for bit."""
    from pybullet_robot_envs.model.table import GRASP_POS
    eng, ora, info = parity.make_hands_pair(_capi.Engine, hip_lib, 5, "r", 0, obj_std=0.03)
    eng.reset()
    eng.set_motors(info["fingers"], GRASP_POS, 0.1, 10.0, mask=[1, 0, 1, 0, 1])
    mot = eng.get_motor_state()
    assert mot.shape == (5, 4, 128) and (mot[0, 2, info["fingers"]] < 1).all() and (mot[1, 2, info["fingers"]] == 1).all()
    a = np.tile(np.asarray(info["home"], np.float32)[info["controlled"]], (5, 1))
    a[:, 5] += 0.2
    for _ in range(4):
        eng.step(a)                  # joint control drives all 37 controlled joints, fingers included: default force again
    st, mot = eng.get_state(), eng.get_motor_state()
    assert (mot[:, 2, info["fingers"]] == 1).all() and np.allclose(mot[:, 1, info["fingers"]], 0.5)
    ref = [eng.step(a)[0].copy() for _ in range(3)]
    eng.set_state(st); eng.set_motor_state(mot)
    for k in range(3):
        assert np.array_equal(eng.step(a)[0], ref[k])


def test_reference_grasp_demo_sequence(hip_lib):
    """helloworld_icub.py's scripted phases on a batch: IK reaches the commanded hand poses, the closing fingers touch the object."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import demo_icub_hands
    from pybullet_robot_envs import _client
    from pybullet_robot_envs.envs.icub_envs.icub_env_with_hands import iCubHandsEnv
    cid = _client.connect(6, lib=hip_lib)
    robot = iCubHandsEnv(cid, use_IK=1, control_arm='r')
    lines = []
    demo_icub_hands.run(robot, log=lines.append)
    assert len(lines) == 7
    hand = lambda l: np.array(l.split("hand [")[1].split("]")[0].split(), float)
    assert np.abs(hand(lines[1]) - [0.49, 0.0, 0.8]).max() < 5e-3
    assert np.abs(hand(lines[2]) - [0.485, 0.0, 0.72]).max() < 5e-3
    assert np.abs(hand(lines[5]) - [0.3, -0.2, 0.9]).max() < 5e-3
    obj = lambda l: np.array(l.split("object [")[1].split("]")[0].split(), float)
    assert np.abs(obj(lines[3]) - obj(lines[2])).max() > 5e-3          # the closing fingers reached (and moved) the object
    obs, _ = robot.get_observation()
    assert obs.shape == (6, 46) and np.isfinite(obs).all()
    n, f = robot.check_contact_fingertips()
    assert n.shape == (6,) and f.shape == (6, 5)
    _client.disconnect(cid)


def test_hands_sharding_invariance(hip_lib):
    """RNG streams are keyed by the global env id: two shards with env_id_base 0 / 4 reproduce one 8-env engine bit for bit."""
    kw = dict(obj_std=0.04)
    full, _, info = parity.make_hands_pair(_capi.Engine, hip_lib, 8, "r", 0, **kw)
    a, _, _ = parity.make_hands_pair(_capi.Engine, hip_lib, 4, "r", 0, env_id_base=0, **kw)
    b, _, _ = parity.make_hands_pair(_capi.Engine, hip_lib, 4, "r", 0, env_id_base=4, **kw)
    of, oa, ob_ = full.reset(), a.reset(), b.reset()
    assert np.array_equal(of, np.concatenate([oa, ob_])) and np.ptp(of[:, 46]) > 1e-3        # object x differs between envs
    act = np.tile(np.asarray(info["home"], np.float32)[info["controlled"]], (8, 1))
    act += np.random.default_rng(3).uniform(-0.2, 0.2, act.shape).astype(np.float32)
    for _ in range(3):
        rf, ra, rb = full.step(act), a.step(act[:4]), b.step(act[4:])
    assert np.array_equal(rf[0], np.concatenate([ra[0], rb[0]]))


def test_implicit_joint_damping_option(hip_lib, panda):
    parity.check_implicit_damping(_capi.Engine, hip_lib, panda["table"], n=64)


def test_hands_force_limited_reset(hip_lib):
    parity.check_hands_force_limited_reset(_capi.Engine, hip_lib, n=3)


def test_config5_at_size(hip_lib):
    """BASELINE config 5 at its per-GPU size (8192 = 65536 / 8 envs of the 60-DoF iCub with hands): reset and the reference demo's
    scripted phases (pre-grasp -> approach -> grasp -> lift -> move -> open, helloworld_icub.py:61-125) on the whole batch.
    Size-independent properties: everything finite, unit quaternions, the grasp HOLDS -- at least three fingertips on the brick after
    the closing phase, the brick goes up with the hand (>= 5 cm above its rest height at the end of the lift; measured 19 cm), is
    carried to the right and falls back onto the table when the hand opens -- and, the object pose not being randomised here, all
    8192 replicas are bit-identical."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import demo_icub_hands
    from pybullet_robot_envs import _client
    from pybullet_robot_envs.envs.icub_envs.icub_env_with_hands import iCubHandsEnv
    n = 8192
    cid = _client.connect(n, lib=hip_lib)
    robot = iCubHandsEnv(cid, use_IK=1, control_arm='r')
    eng = robot._engine
    seen = {"tips": 0, "points": 0, "obj": []}

    def log(line):
        st = eng.get_state()
        nd = eng.ndof
        assert np.isfinite(st).all(), line
        assert np.abs(np.linalg.norm(st[:, nd + 3:nd + 7], axis=1) - 1).max() < 1e-5, line
        assert np.array_equal(st, np.broadcast_to(st[0], st.shape)), "replicas diverged: " + line
        seen["tips"] = max(seen["tips"], int(st[0, nd + 12])); seen["points"] = max(seen["points"], int(st[0, nd + 13]))
        seen["obj"].append(st[0, nd:nd + 3].copy())
    demo_icub_hands.run(robot, log=log)
    print("config 5 at size:", seen["tips"], "fingertips /", seen["points"], "robot-object contact points seen at the phase ends")
    obj = seen["obj"]            # after reset, 1 above, 2 turned, 3 closed, 4 up, 5 right, 6 open
    assert seen["tips"] >= 3, seen
    assert obj[4][2] - obj[2][2] >= 0.05, ("the grasp did not lift the brick", obj)
    assert np.hypot(obj[5][0] - 0.3, obj[5][1] + 0.2) < 0.08 and obj[5][2] > obj[2][2] + 0.05, ("the brick was not carried with the hand", obj)
    assert 0.625 + 0.02 < obj[6][2] < 0.625 + 0.045, ("the released brick did not come to rest on the table (on one of its faces)", obj)
    obs, _ = robot.get_observation()
    assert obs.shape == (n, 46) and np.isfinite(obs).all()
    _client.disconnect(cid)


@pytest.mark.parametrize("arm", ["r", "l"])
def test_hands_five_fingertips_on_the_object(hip_lib, arm):
    w = parity.check_hands_five_fingertips(_capi.Engine, hip_lib, arm)
    print("five fingertips (%s):" % arm, w)
    assert w["fingertips_in_contact"] == 5


def test_scripted_grasp_against_the_oracle(hip_lib):
    """config 5 closed loop: the reference's grasp demo on the engine and, command by command, on the fp64 oracle -- both close on the
    brick, lift it ~19 cm, carry and release it; the brick's position is compared per phase (parity.check_hands_demo_against_oracle)"""
    rep = parity.check_hands_demo_against_oracle(hip_lib, n=2)
    print("scripted grasp, engine vs oracle:", rep)
    assert rep["lift_oracle_m"] > 0.15 and rep["lift_engine_m"] > 0.15



def test_hands_solver_residual_threshold(hip_lib):
    parity.check_hands_residual_threshold(_capi.Engine, hip_lib, n=2, steps=2)
