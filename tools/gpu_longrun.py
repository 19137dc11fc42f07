import sys, os, time, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'pybullet-robot-envs_amd'))
from pybullet_robot_envs import _capi
from pybullet_robot_envs.model.table import panda_table
tbl,_=panda_table()
n=int(sys.argv[1]) if len(sys.argv)>1 else 32768
eng=_capi.Engine(tbl,task=1,num_envs=n,obj_pose_rnd_std=0.05,tg_pose_rnd_std=0.2)
eng.reset()
dev=torch.device('cuda',0)
s=torch.cuda.Stream(); torch.cuda.set_stream(s)
out=torch.zeros((n,eng.obs_dim+2),device=dev)
g=torch.Generator(device=dev); g.manual_seed(0)
for blk in range(11):
    acts=[torch.rand((n,7),device=dev,generator=g)*2-1 for _ in range(8)]
    torch.cuda.synchronize(); t0=time.perf_counter()
    for k in range(100):
        eng.step_device(acts[k%8].data_ptr(), out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize(); dt=time.perf_counter()-t0
    info=eng.kernel_info()
    done=float(out[:,-1].mean()); fin=bool(torch.isfinite(out).all())
    st=eng.get_state()
    print("steps %4d  %.3f ms/step  %.1fM env-steps/s  rc-path envs %d (%.2f%%) row-kernel envs %d  done %.3f finite %s  obj z min %.3f  moved>1cm %.3f"%((blk+1)*100, dt*10, n*100/dt/1e6, info[5], 100*info[5]/n, info[4], done, fin, st[:,11].min(), (np.hypot(st[:,9]-0.45, st[:,10])>0.06).mean()))
