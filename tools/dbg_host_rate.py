"""Is the host keeping ahead of the GPU?  Host time to ENQUEUE a step (no synchronisation) against the step's GPU time:
    python tools/dbg_host_rate.py"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, tsdf_amd
from tsdf_amd import synth
W, H, n, K, Wu = 640, 480, 512, 100, 10
frames, cams = [], []
for i in range(K + Wu):
    d, cam = synth.depth_frame(i % 200, 200, seed=0x5EED0003); frames.append(d); cams.append(cam)
depth = torch.from_numpy(np.stack(frames).view(np.int16)).cuda()
filt = torch.empty((H * W,), dtype=torch.int16, device="cuda"); tmax = torch.empty((1200,), dtype=torch.int16, device="cuda")
vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda"); norm = torch.empty_like(vert)
vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3); s = torch.cuda.current_stream(); vol.set_stream(s.cuda_stream)
bil = tsdf_amd.BilateralFilter(30.0, 4.5); rc = tsdf_amd.GPURaycaster(W, H)
def step(i):
    bil.filter_device(depth[i].data_ptr(), filt.data_ptr(), W, H, bits=16, stream=s.cuda_stream, tile_max_ptr=tmax.data_ptr())
    vol.integrate_device(filt.data_ptr(), W, H, cams[i], tile_max_ptr=tmax.data_ptr())
    rc.raycast_device(vol, cams[i], vert.data_ptr(), norm.data_ptr())
for i in range(Wu): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(Wu, Wu + K): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us per step; whole step %.1f us; host idle at the end %.1f us per step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6, (t2 - t1) / K * 1e6))
