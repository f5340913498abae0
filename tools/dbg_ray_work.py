"""Where the ray march's passes go on the bench scene: per-ray counts of evaluated samples, jumps and cell tests
(tsdf_raycast_evaluated_samples, the single-range instrumented kernel) after `frames` frames of the bench stream.
    python tools/dbg_ray_work.py [frames=40] [grid=512]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, tsdf_amd
from tsdf_amd import synth
W, H = 640, 480
F = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
rc = tsdf_amd.GPURaycaster(W, H)
for i in range(F):
    d, cam = synth.depth_frame(i, 200, seed=0x5EED0003)
    f = d.copy(); bil.filter(f, W, H)
    vol.integrate(f, W, H, cam)
st = rc.stats(vol, cam, per_ray_work=True)
pr = st["per_ray"].reshape(H, W, 3)
ev, hops, cells = pr[..., 0], pr[..., 1], pr[..., 2]
V, N = vol.raycast(W, H, cam)
hit = ~np.isnan(V[:, 0]).reshape(H, W)
print("reference samples %d, evaluated %d; per ray: evaluated %.1f, jumps %.1f, cell tests %.1f; hits %.1f %%" %
      (st["samples"], st["evaluated"], ev.mean(), hops.mean(), cells.mean(), 100 * hit.mean()))
for name, a in (("evaluated", ev), ("jumps", hops), ("cell tests", cells), ("passes", ev + hops + cells)):
    q = np.percentile(a, [10, 50, 90, 99, 99.9, 100])
    print("%-10s p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f   share of the total in the top 10 %% of rays: %.2f" %
          (name, *q, np.sort(a.ravel())[-a.size // 10:].sum() / max(a.sum(), 1)))
# rays next to a depth discontinuity (silhouettes) against the rest
zz = V[:, 2].reshape(H, W)
gy, gx = np.gradient(np.nan_to_num(zz, nan=1e6))
edge = (np.abs(gx) + np.abs(gy)) > 30.0
import scipy.ndimage as ndi
edge = ndi.binary_dilation(edge, iterations=3)
for name, m in (("near silhouettes (%.1f %% of rays)" % (100 * edge.mean()), edge), ("elsewhere", ~edge), ("misses", ~hit)):
    if m.any():
        print("%-40s evaluated %.1f  jumps %.1f  cell tests %.1f" % (name, ev[m].mean(), hops[m].mean(), cells[m].mean()))
occ = vol.occupancy_data(force_rebuild=False)
fine, cell, reach = occ
print("bricks: fine flagged %.2f %%, cell flagged %.2f %%" % (100 * (fine != 0).mean(), 100 * (cell != 0).mean()))
