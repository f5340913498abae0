import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, time, torch
from tsdf_amd import synth
n=512
v=tsdf_amd.TSDFVolume((n,n,n),(3000.,)*3)
bil=tsdf_amd.BilateralFilter(30.0,4.5)
rc=tsdf_amd.GPURaycaster(640,480)
for i in range(0,8):
    d,cam=synth.depth_frame(i,200,seed=0x5EED0003)
    f=d.copy(); bil.filter(f,640,480)
    v.integrate(f,640,480,cam)
st=rc.stats(v,cam,per_ray_work=True)
pr=st["per_ray"].reshape(480,640,3)
ev,hops,cells=pr[...,0],pr[...,1],pr[...,2]
print("per ray: exact", ev.mean(), "hops", hops.mean(), np.percentile(hops,[50,90,99]), "cell tests", cells.mean(), np.percentile(cells,[50,90,99]))
tot=ev+hops+cells
print("lane trips mean", tot.mean(), "per-wave max mean", tot.reshape(60,8,80,8).max(axis=(1,3)).mean())
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/per_ray.npz", ev=ev.astype(np.uint16), hops=hops.astype(np.uint16), cells=cells.astype(np.uint16))
