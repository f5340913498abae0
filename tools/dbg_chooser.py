"""The cast chosen from measured times (raycast.hip: choose_cast) on four streams of casts: python tools/dbg_chooser.py
  bench   the bench scene from outside (the cells win: 0.100 against 0.149 ms)
  wall    a flat wall in front of the camera (the march wins: 0.094 against 0.122)
  inside  the config-4 view from inside a 1024^3 volume (the cells, sorted front to back: 0.175 against 0.197)
  close   a wall six voxels behind the face the camera looks through, the camera five voxels outside (the march: 0.08 against 2 ms)
Prints per scene the kinds of the casts, the wall time per cast over the stream, and what the two fixed settings cost (TSDF_RAY_CELLS=0 / 2
in processes of their own: the knob is read once)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, torch
from tsdf_amd import synth
scene = sys.argv[1] if len(sys.argv) > 1 else "bench"
n_casts = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = 1024 if scene == "inside" else 512
v = tsdf_amd.TSDFVolume((n, n, n), (3000.,) * 3)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
rc = tsdf_amd.GPURaycaster(640, 480)
vert = torch.empty((640 * 480, 3), dtype=torch.float32, device='cuda'); norm = torch.empty_like(vert)
s = torch.cuda.current_stream(); v.set_stream(s.cuda_stream)
if scene in ("bench", "inside"):
    for i in range(40):
        d, cam = synth.depth_frame(i, 100 if scene == "inside" else 200, seed=0x5EED0004 if scene == "inside" else 0x5EED0003, inside=scene == "inside")
        f = d.copy(); bil.filter(f, 640, 480); v.integrate(f, 640, 480, cam)
else:
    vs = 3000.0 / n
    cam = tsdf_amd.Camera.default_depth_camera()
    cam.move_to(1500.0, 1500.0, -(100.0 if scene == "wall" else 5.0) * vs)
    cam.look_at(1500.0, 1500.0, 1500.0)
    depth = np.full(640 * 480, int(round((100.0 if scene == "wall" else 5.0) * vs + 6.0 * vs)), np.uint16)
    for i in range(4): v.integrate(depth, 640, 480, cam)
kinds, t_casts = [], []
torch.cuda.synchronize()
t0 = time.time()
for j in range(n_casts):
    rc.raycast_device(v, cam, vert.data_ptr(), norm.data_ptr())
    kinds.append(1 if v.last_raycast_cell_parallel() else 0)
torch.cuda.synchronize()
wall = (time.time() - t0) / n_casts * 1e3
runs = "".join("c" if k else "m" for k in kinds)
print("%-6s TSDF_RAY_CELLS=%s: %.4f ms per cast over %d casts; cells in %d of them; last 32: %s; first 80: %s" % (scene, os.environ.get("TSDF_RAY_CELLS", "1 (chooser)"), wall, n_casts, sum(kinds), runs[-32:], runs[:80]))
