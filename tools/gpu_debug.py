import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,os.path.join(ROOT,'pybullet-robot-envs_amd'))
import orc, scenarios, parity
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table, PANDA_SPHERES
np.set_printoptions(precision=5, suppress=True, linewidth=220)
tbl,model=panda_table()
hip=_capi.load(); emu=_capi.load(os.path.join(ROOT,'tests/host_emu/build/libpbre_emu.so'))
panda={"table":tbl,"model":model,"spheres":PANDA_SPHERES}
o=orc.Oracle(tbl); base,_=o.batch_reset(1)
rng=np.random.default_rng(1)
S=parity.contact_states(o,panda,base[0],rng,24,24).astype(np.float32); N=len(S)
a=rng.uniform(-1,1,(N,7)).astype(np.float32)
def run(lib,S,a):
    e=_capi.Engine(tbl,task=1,num_envs=len(S),lib=lib); e.set_state(S); r=e.step(a); return e.get_state(),r
sh,rh=run(hip,S,a); se,re_=run(emu,S,a)
so,out=o.batch_step(S.astype(np.float64),a)
home=scenarios.HOME
for i in range(N):
    s1,_=run(hip,S[i:i+1],a[i:i+1])
    st,info=o.sim_step(S[i].astype(np.float64), np.r_[S[i,:7]+0.05*a[i],home[7:]],[0.5]*7+[0.2]*2,[1.0]*9)
    d=np.abs(sh[i]-se[i]); k=int(d.argmax())
    print(i,"types",[info.type[c] for c in range(info.ncontacts)],"hip-emu %.2e at %d"%(d.max(),k),"hip1-emu %.2e"%np.abs(s1[0]-se[i]).max(),"emu-orc %.2e"%parity.rel(se[i],so[i]).max(), "lam", [round(info.lambda_n[c],3) for c in range(info.ncontacts)])
bad=[i for i in range(N) if np.abs(sh[i]-se[i]).max()>1e-3]
o.params.contact_margin=1e-3+3e-6; sp,_=o.batch_step(S.astype(np.float64),a)
o.params.contact_margin=1e-3-3e-6; sm,_=o.batch_step(S.astype(np.float64),a)
for i in bad:
    print("BAD",i,"hip",sh[i,16:31]); print("   emu",se[i,16:31]); print("   orc",so[i,16:31]); print("  margin+ diff", np.abs(sp[i]-so[i]).max(), "margin- diff", np.abs(sm[i]-so[i]).max())
