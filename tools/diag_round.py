"""Crafted iCub contact states (tests/parity.py: check_icub_contact_states) per object primitive and kernel path, in measurement mode:
prints the worst difference to the oracle per contact kind (hand on the object / on the table / both / joint limit).  GPU box.
    python tools/diag_round.py"""
import os, sys, json
sys.path.insert(0, "tests"); sys.path.insert(0, "pybullet-robot-envs_amd")
os.environ["PBRE_PARITY_MEASURE"] = "1"
import parity
from pybullet_robot_envs import _capi
lib = _capi.load()
for name in ("YcbTennisBall", "YcbMasterChefCan", "duck_vhacd", None):
    for lane in ("1", "0"):
        os.environ["PBRE_ICUB_LANE"] = lane
        rep = parity.check_icub_contact_states(_capi.Engine, lib, n_each=12, obj_name=name)
        print(name, "lane", lane, json.dumps({k: rep[k] for k in ("object", "table", "both", "limit", "skipped_ambiguous", "complex_envs_stepped") if k in rep}))
