"""A wall a few voxels behind the face the camera looks through, the camera a few voxels outside: the cell-parallel cast's worst view
(cells of hundreds of pixels).  python tools/dbg_near_wall.py [grid]   (run once per TSDF_RAY_CELLS / TSDF_RAY_CELLS_LOOK setting)"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, torch
from tests.helpers import camera_at
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
v = tsdf_amd.TSDFVolume((n, n, n), (3000.,) * 3)
trunc = v.truncation_distance()
vs = 3000.0 / n
z = (np.arange(n, dtype=np.float32) + 0.5) * vs
plane = np.clip(6.0 * vs - z, -trunc, trunc).astype(np.float32)          # a wall six voxels behind the z = 0 face, seen from outside ...
plane[z > 6.0 * vs + trunc] = trunc                                      # ... and nothing behind its band (as integrate leaves it)
v.set_distance_data(np.repeat(plane, n * n))
rc = tsdf_amd.GPURaycaster(640, 480)
vert = torch.empty((640 * 480, 3), dtype=torch.float32, device='cuda'); norm = torch.empty_like(vert)
for dist in (5, 20, 100):
    cam = camera_at((1500, 1500, -dist * vs), look_at=(1500, 1500, 3000))
    for r in range(3): rc.raycast_device(v, cam, vert.data_ptr(), norm.data_ptr())
    torch.cuda.synchronize(); t = time.time()
    for r in range(10): rc.raycast_device(v, cam, vert.data_ptr(), norm.data_ptr())
    torch.cuda.synchronize()
    print("grid", n, "camera", dist, "voxels outside: ms per cast %.4f" % ((time.time() - t) / 10 * 1e3), "cells" if v.last_raycast_cell_parallel() else "march",
          "hits", int((~torch.isnan(vert[:, 0])).sum().item()))
