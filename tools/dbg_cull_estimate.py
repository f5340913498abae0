"""CPU estimate (oracle as the counter; no GPU) of what a finer cull of integrate's bricks could skip on a workload:
for frames of the stream, the exact set of voxels the frame updates (weights of a fresh oracle volume), then for several
work-item shapes (x, y, z voxels) the number of items holding any updated voxel and the share of their voxels that is updated.
    python tools/dbg_cull_estimate.py --grid 1024 --inside --frames 8 9 25 50"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from tsdf_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--inside", action="store_true")
ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0004)
ap.add_argument("--stream-frames", type=int, default=100)
ap.add_argument("--frames", type=int, nargs="+", default=[8, 9])
a = ap.parse_args()
n = a.grid
ov = O.Volume((n, n, n), (3000.0,) * 3)
shapes = [(64, 4, 32), (64, 4, 16), (64, 4, 8), (64, 4, 4), (64, 1, 4), (32, 4, 8), (16, 4, 8), (64, 2, 8), (64, 8, 8), (64, 1, 32), (32, 4, 32), (64, 4, 1)]
for fi in a.frames:
    d, cam = synth.depth_frame(fi, a.stream_frames, seed=a.seed, inside=a.inside)
    f = O.bilateral_u16(d, 640, 480, 30.0, 4.5, nthreads=O.max_threads()).reshape(-1)
    ov.clear()
    t0 = time.time()
    ov.integrate(f, 640, 480, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=O.max_threads())
    m = (ov.weight.reshape(n, n, n) > 0)   # [z][y][x]
    U = int(m.sum())
    print("frame %d: U = %d (%.1f s)" % (fi, U, time.time() - t0), flush=True)
    for (sx, sy, sz) in shapes:
        r = m.reshape(n // sz, sz, n // sy, sy, n // sx, sx).any(axis=(1, 3, 5))
        k = int(r.sum())
        print("   item %2dx%dx%-2d: %7d items hold an update, %5.1f M voxels walked, %4.1f %% updated"
              % (sx, sy, sz, k, k * sx * sy * sz / 1e6, 100.0 * U / (k * sx * sy * sz)), flush=True)
