import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, time, torch
from tsdf_amd import synth
n=512
v=tsdf_amd.TSDFVolume((n,n,n),(3000.,)*3)
bil=tsdf_amd.BilateralFilter(30.0,4.5)
rc=tsdf_amd.GPURaycaster(640,480)
vert=torch.empty((640*480,3),dtype=torch.float32,device='cuda')
s=torch.cuda.current_stream(); v.set_stream(s.cuda_stream)
for i in range(0,8):
    d,cam=synth.depth_frame(i,200,seed=0x5EED0003)
    f=d.copy(); bil.filter(f,640,480)
    v.integrate(f,640,480,cam)
for r in range(3):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(s); rc.raycast_device(v,cam,vert.data_ptr(),None); e1.record(s); torch.cuda.synchronize()
    print("ray ms %.3f"%e0.elapsed_time(e1))
st=rc.stats(v,cam,per_ray_work=True)
pr=st["per_ray"].reshape(480,640,3)
ev,it,adv=pr[...,0],pr[...,1],pr[...,2]
print("fastdiv", v.info().fast_division_verified, "occ",v.occupancy(),"evaluated total",st["evaluated"], "S", st["samples"], "mean/ray", ev.mean(), "max", ev.max(), "p50,p90,p99", np.percentile(ev,[50,90,99]))
print("cell tests per ray: mean", adv.mean(), "max", adv.max(), "p50,p90,p99", np.percentile(adv,[50,90,99]))
t_it=it.reshape(60,8,80,8).max(axis=(1,3)); t_ev=ev.reshape(60,8,80,8); t_adv=adv.reshape(60,8,80,8)
print("per-wave trips: mean", t_it.mean(), "max", t_it.max(), " per-wave max adv iters mean", t_adv.max(axis=(1,3)).mean())
np.set_printoptions(linewidth=250)
print("per-wave trips map (/10), 60x80 waves -> 30x40 blocks max")
print((t_it.reshape(30,2,40,2).max(axis=(1,3))/10).astype(int))
print("cell tests per ray /10 (block mean)")
print((adv.reshape(30,16,40,16).mean(axis=(1,3))/10).round(0).astype(int))
