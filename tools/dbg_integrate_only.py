"""Times integrate alone (kernel-only HIP events) on the bench scene: python tools/dbg_integrate_only.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, time, torch
from tsdf_amd import synth
n=int(sys.argv[1]) if len(sys.argv) > 1 else 512
v=tsdf_amd.TSDFVolume((n,n,n),(3000.,)*3)
bil=tsdf_amd.BilateralFilter(30.0,4.5)
s=torch.cuda.current_stream(); v.set_stream(s.cuda_stream)
fr=[]
N=int(sys.argv[2]) if len(sys.argv) > 2 else 12
for i in range(0,N):
    d,cam=synth.depth_frame(i,200,seed=0x5EED0003)
    f=d.copy(); bil.filter(f,640,480)
    fr.append((torch.from_numpy(f.astype(np.int16)).cuda(),cam))
for i in range(4):
    v.integrate_device(fr[i][0].data_ptr(),640,480,fr[i][1])
torch.cuda.synchronize()
v.set_timing(True)
for i in range(4,N):
    v.integrate_device(fr[i][0].data_ptr(),640,480,fr[i][1])
torch.cuda.synchronize()
print(os.environ.get("TAG",""), "integrate kernel ms", v.kernel_time("integrate"))
