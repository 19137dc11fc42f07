"""Bullet's residual exit in the simple class, closed form against explicit rows, on the CPU emulation: two engines get the SAME
state before every step -- one takes the closed path (motor_scan + obj_closed<PERLANE>), the other is forced onto the explicit RT rows
(PBRE_F_SEQ_MOTORS | PBRE_F_SEQ_OBJECT) -- and the sweep counts / joint velocities are compared.  Needs tests/host_emu/build/libpbre_emu.so
(built with -DPBRE_EMU_OC_STATS for the qualification counters).   usage: python tools/rt_closed_check.py [envs=256] [steps=40]"""
import sys, ctypes as C, time
sys.path.insert(0,'tests'); sys.path.insert(0,'pybullet-robot-envs_amd')
import numpy as np
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl,_=panda_table()
lib=_capi.load('tests/host_emu/build/libpbre_emu.so')
n=int(sys.argv[1]) if len(sys.argv)>1 else 256
steps=int(sys.argv[2]) if len(sys.argv)>2 else 40
kw=dict(task=1,num_envs=n,lib=lib,seed=5,obj_pose_rnd_std=0.05,tg_pose_rnd_std=0.2,flags=_capi.F_AUTO_RESET,max_steps=200)
a=_capi.Engine(tbl,**kw)                       # closed
kw2=dict(kw); kw2["flags"]=kw["flags"]|32|64
b=_capi.Engine(tbl,**kw2)                      # explicit rows
for e in (a,b): e.reset(); e.set_physics(solver_residual_threshold=1e-7)
st=a.get_state(); b.set_state(st)
rng=np.random.default_rng(3)
stats=(C.c_long*2)()
lib.pbre_emu_oc_stats(stats,1)
tot=0; flips=0; maxgap=0; worst=0.0; hist=[]
for k in range(steps):
    act=rng.uniform(-1,1,(n,7)).astype(np.float32)
    s0=a.get_state(); b.set_state(s0)           # same state into both every step
    ra=a.step(act); rb=b.step(act)
    sa=a.get_sweeps(); sb=b.get_sweeps()
    d=np.abs(sa-sb); tot+=n; flips+=int((d>0).sum()); maxgap=max(maxgap,int(d.max()))
    same=d==0
    ea=a.get_state(); eb=b.get_state()
    worst=max(worst,float(np.abs(ea[same][:,16:25]-eb[same][:,16:25]).max()))
    hist.append(sa.copy())
lib.pbre_emu_oc_stats(stats,0)
h=np.concatenate(hist)
print("env-steps",tot,"sweep-count flips",flips,"max gap",maxgap,"worst qd diff (same count)",worst)
print("closed path: lanes ok",stats[1],"not ok",stats[0],"  sweeps median",np.median(h),"p90",np.percentile(h,90),"at cap",float((h>=150).mean()))
