"""The tracked loop (BASELINE configs[4]) alone, for timelines:  python tools/dbg_tracking.py [frames]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, torch
from tsdf_amd import synth
from tsdf_amd.tracking import FrameToModelTracker
n = 512
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
tr = FrameToModelTracker(vol, 640, 480)
fr = [synth.depth_frame(i, 200, seed=0x5EED0003) for i in range(frames)]
dev = [torch.from_numpy(d.view(np.int16)).cuda() for d, _ in fr]
for i, (d, cam) in enumerate(fr):
    if i == 4:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.process_device(dev[i].data_ptr(), initial_pose=cam.pose().astype(np.float64).reshape(4, 4).T if i == 0 else None)
torch.cuda.synchronize()
print("tracked loop ms per frame %.4f" % ((time.perf_counter() - t0) * 1e3 / (frames - 4)))
