"""The ray cast alone, march kernels against the cell-parallel cast, on the bench scene: python tools/dbg_ray_cells.py [frames] [grid] [inside]
(run once per TSDF_RAY_CELLS setting: the knob is read once per process)"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, time, torch
from tsdf_amd import synth
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
inside = len(sys.argv) > 3 and sys.argv[3] == 'inside'
v = tsdf_amd.TSDFVolume((n, n, n), (3000.,) * 3)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
rc = tsdf_amd.GPURaycaster(640, 480)
vert = torch.empty((640 * 480, 3), dtype=torch.float32, device='cuda')
norm = torch.empty_like(vert)
s = torch.cuda.current_stream(); v.set_stream(s.cuda_stream)
for i in range(frames):
    d, cam = synth.depth_frame(i, 100 if inside else 200, seed=0x5EED0004 if inside else 0x5EED0003, inside=inside)
    f = d.copy(); bil.filter(f, 640, 480)
    v.integrate(f, 640, 480, cam)
for r in range(3):
    rc.raycast_device(v, cam, vert.data_ptr(), norm.data_ptr())
torch.cuda.synchronize()
v.set_timing(True)
t = time.time()
for r in range(20):
    rc.raycast_device(v, cam, vert.data_ptr(), norm.data_ptr())
torch.cuda.synchronize()
wall = (time.time() - t) / 20 * 1e3
bits = int(vert.view(torch.int32).to(torch.int64).sum().item()), int(norm.view(torch.int32).to(torch.int64).sum().item())
print("cells", os.environ.get("TSDF_RAY_CELLS", "default"), "grid", n, "main kernel ms %.4f tail ms %.4f" % (v.kernel_time("raycast")[1], v.kernel_time("raycast_tail")[1]),
      "wall ms per raycast", round(wall, 4), "hits", int((~torch.isnan(vert[:, 0])).sum().item()), "picture bits", bits)
