"""What the passes of the two ray-march kernels are spent on (a library built with -DTSDF_DIAG_RAY_MIX, TSDF_HIP_LIB):
python tools/dbg_ray_mix.py [frames] [grid] [inside]"""
import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, torch
from tsdf_amd import synth, _capi
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
inside = len(sys.argv) > 3 and sys.argv[3] == 'inside'
v = tsdf_amd.TSDFVolume((n, n, n), (3000.,) * 3)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
rc = tsdf_amd.GPURaycaster(640, 480)
vert = torch.empty((640 * 480, 3), dtype=torch.float32, device='cuda')
for i in range(frames):
    d, cam = synth.depth_frame(i, 100 if inside else 200, seed=0x5EED0004 if inside else 0x5EED0003, inside=inside)
    f = d.copy(); bil.filter(f, 640, 480)
    v.integrate(f, 640, 480, cam)
for r in range(3):
    rc.raycast_device(v, cam, vert.data_ptr(), None)
fn = _capi.lib.tsdf_debug_ray_mix
fn.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
out = (ctypes.c_uint64 * 64)()
fn(out)
rc.raycast_device(v, cam, vert.data_ptr(), None)
fn(out)
c = np.array(out[:], dtype=np.int64)
names = ["block jump", "cell-brick jump", "other slab", "positive cell", "evaluated: hit", "evaluated: ahead to exit", "evaluated: ahead short", "full path"]
for base, what in ((0, "bulk kernel, lane-passes"), (8, "tail kernel, lane-rounds"), (16, "tail kernel, lane 0 of the groups")):
    tot = max(1, int(c[base:base + 8].sum()))
    print(what, "total", tot)
    for i, nm in enumerate(names):
        print("   %-26s %9d  %5.1f %%" % (nm, c[base + i], 100.0 * c[base + i] / tot))
print("bulk wave-passes by active lanes (1-4, 5-16, 17-32, 33-64):", c[24:28].tolist(), " passes 12+:", c[28:32].tolist())
print("cell cast: bricks %d, mixed cells %d, (cell, pixel) pairs %d, pairs whose ray crosses the cell %d, ... not behind a known hit %d, candidate samples %d, evaluated from the cell %d, by the full path %d" % tuple(c[32:40].tolist()))
print("cell cast hits lowered: by the walk %d, by shell tasks %d" % tuple(c[40:42].tolist()))
print("cell cast shell tasks: bricks %d, pixels asked %d, candidate samples %d; face samples from the staged voxels %d" % tuple(c[42:46].tolist()))
