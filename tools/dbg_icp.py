"""Times ICP tracking (init of both pyramids + 19 iterations) on the bench scene: python tools/dbg_icp.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, time, torch
from tsdf_amd import synth
W,H=640,480
d0,cam0=synth.depth_frame(0,200,seed=0x5EED0005,noise=False)
d1,cam1=synth.depth_frame(3,200,seed=0x5EED0005,noise=False)
icp=tsdf_amd.ICPOdometry(W,H,331.0,234.6,591.1,590.1)
s=torch.cuda.current_stream(); icp.set_stream(s.cuda_stream)
a=torch.from_numpy(d0.astype(np.int16)).cuda(); b=torch.from_numpy(d1.astype(np.int16)).cuda()
for r in range(3):
    icp.init_icp_device(a.data_ptr(),model=True); icp.init_icp_device(b.data_ptr()); T=icp.get_incremental_transformation()
torch.cuda.synchronize()
n=20
t=time.time()
for r in range(n):
    icp.init_icp_device(a.data_ptr(),model=True); icp.init_icp_device(b.data_ptr())
torch.cuda.synchronize(); t_init=(time.time()-t)/n*1e3
t=time.time()
for r in range(n):
    T=icp.get_incremental_transformation()
t_iter=(time.time()-t)/n*1e3
print("ICP: init (2 pyramids) %.3f ms, 19 iterations %.3f ms, inliers %.0f" % (t_init,t_iter,icp.last_inliers))
