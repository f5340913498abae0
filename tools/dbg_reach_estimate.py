"""CPU estimate (oracle volume, numpy; no GPU) of the free-space hops of the ray march under two summaries of the brick flags:
  (a) reach[b] = largest clear ALIGNED block of 1/2/4/8/16 bricks that holds b (what raycast.hip used up to round 3), and
  (b) centred cubes from Chebyshev distance maps: d0 = distance in bricks to the nearest flagged brick (cap C0), d2 = distance
      in 4-brick units to the nearest unit holding a flagged brick (cap C2); a sample in b may jump to the exit of the cube of
      (2 d - 1) units centred on its own unit.
Rays of every 4th pixel of the bench scene after `frames` frames; flagged bricks are stepped through brick by brick and not
counted (both schemes do the same work there).   python tools/dbg_reach_estimate.py [frames=40] [grid=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.ndimage as ndi
import oracle as O
from tsdf_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
inside = len(sys.argv) > 3 and sys.argv[3] == "inside"
W, H = 640, 480
ov = O.Volume((n, n, n), (3000.0,) * 3)
T = O.max_threads()
cache = "/tmp/work/reach_est_%d_%d_%d.npz" % (F, n, inside)
have = os.path.exists(cache)
for i in range(F):
    if have:
        d, cam = synth.depth_frame(F - 1, 100 if inside else 200, seed=0x5EED0004 if inside else 0x5EED0003, inside=inside)
        break
    d, cam = synth.depth_frame(i, 100 if inside else 200, seed=0x5EED0004 if inside else 0x5EED0003, inside=inside)
    f = O.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=T).reshape(-1)
    ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=T)
vs = 3000.0 / n
trunc = float(ov.truncation_distance())
tau = 0.01 * trunc
nb = n // 4
if have:
    z = np.load(cache); fine = z["fine"]; th_cached = z["th"]
D = ov.dist.reshape(n, n, n)
if not have:
  if True:
    # fine[b]: some voxel of the brick grown by 2 is <= tau (interior rule; boundary bricks: not flat -> flagged)
    low = ~(D > tau)
    g = ndi.maximum_filter(low.astype(np.uint8), size=5, mode="constant", cval=0)      # voxel grown by 2 either way
    fine = g.reshape(nb, 4, nb, 4, nb, 4).max(axis=(1, 3, 5)).astype(bool)
    notflat = ~((D >= 0.9375 * trunc) & (D <= (1 + 1 / 1024) * trunc))
    gf = ndi.maximum_filter(notflat.astype(np.uint8), size=5, mode="constant", cval=0).reshape(nb, 4, nb, 4, nb, 4).max(axis=(1, 3, 5)).astype(bool)
    bnd = np.zeros((nb, nb, nb), bool); bnd[0] = bnd[-1] = True; bnd[:, 0] = bnd[:, -1] = True; bnd[:, :, 0] = bnd[:, :, -1] = True
    fine = np.where(bnd, gf, fine)
print("bricks flagged: %.2f %%" % (100 * fine.mean()))

# (a) aligned reach classes
def aligned_levels(fine):
    lv = np.where(fine, 0, 1).astype(np.int32)
    cur = fine
    for l in range(1, 5):
        m = cur.shape[0] // 2
        cur = cur.reshape(m, 2, m, 2, m, 2).any(axis=(1, 3, 5))
        up = np.repeat(np.repeat(np.repeat(cur, 2 ** l, 0), 2 ** l, 1), 2 ** l, 2)
        lv = np.where((lv == l) & ~up, l + 1, lv)
    return lv
reach = aligned_levels(fine)
# (b) Chebyshev distances
def cheb(mask, cap):
    d = ndi.distance_transform_cdt(~mask, metric="chessboard").astype(np.int32)
    return np.minimum(d, cap)
C0, C2 = 4, 15
d0 = cheb(fine, C0)
unit = fine.reshape(nb // 4, 4, nb // 4, 4, nb // 4, 4).any(axis=(1, 3, 5))
d2 = cheb(unit, C2)

# rays
pose = cam.pose().astype(np.float64).reshape(4, 4).T
kinv = cam.kinv().astype(np.float64).reshape(3, 3).T
ys, xs = np.mgrid[0:H:4, 0:W:4]
pix = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)], -1).astype(np.float64)
dirs = (pix @ kinv.T) @ pose[:3, :3].T
o = pose[:3, 3]
with np.errstate(divide="ignore", invalid="ignore"):
    t0 = (0.0 - o) / dirs; t1 = (3000.0 - o) / dirs
tn = np.nanmax(np.minimum(t0, t1), axis=1); tf = np.nanmin(np.maximum(t0, t1), axis=1)
tn = np.maximum(tn, 0.0)
ok = tf > tn
step = trunc * 0.05
tf = np.minimum(tf, tn + 4402 * step)
if have:
    th = th_cached
else:
    Vo, _ = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=T)
    Vs = Vo.reshape(H, W, 3)[::4, ::4].reshape(-1, 3).astype(np.float64)
    th = ((Vs - o) @ pose[:3, :3])[:, 2]           # camera z of the hit = its ray parameter
    np.savez(cache, fine=fine, th=th)
tend = np.where(np.isnan(th), tf, np.minimum(th, tf))

def march(scheme):
    t = tn.copy() + 1e-3
    hops = np.zeros(t.size, np.int64); band = np.zeros(t.size, np.int64)
    alive = ok & (t < tend)
    for it in range(4000):
        if not alive.any(): break
        idx = np.flatnonzero(alive)
        p = (o + t[idx, None] * dirs[idx]) / vs                     # voxel coordinates
        v = np.clip(np.floor(p).astype(np.int64), 0, n - 1)
        b = v >> 2
        fl = fine[b[:, 2], b[:, 1], b[:, 0]]
        if scheme == "aligned":
            r = reach[b[:, 2], b[:, 1], b[:, 0]]
            sh = 2 + np.maximum(r, 1) - 1
            lo = (v >> sh[:, None]) << sh[:, None]; size = (1 << sh)[:, None]
        else:
            a0 = d0[b[:, 2], b[:, 1], b[:, 0]]; u = b >> 2
            a2 = d2[u[:, 2], u[:, 1], u[:, 0]]
            use2 = a2 >= 2
            lo0 = (b - (np.maximum(a0, 1) - 1)[:, None]) * 4; s0 = (2 * np.maximum(a0, 1) - 1) * 4
            lo2 = (u - (np.maximum(a2, 1) - 1)[:, None]) * 16; s2 = (2 * np.maximum(a2, 1) - 1) * 16
            lo = np.where(use2[:, None], lo2, lo0); size = np.where(use2, s2, s0)[:, None]
        dd = dirs[idx] / vs
        with np.errstate(divide="ignore", invalid="ignore"):
            ex = np.where(dd > 0, (lo + size - p) / dd, np.where(dd < 0, (lo - p) / dd, np.inf))
        dt = np.maximum(ex.min(axis=1), 0.0)
        t[idx] = t[idx] + (np.ceil(dt / step) + 0.0) * step + 1e-6
        hops[idx] += ~fl; band[idx] += fl
        alive[idx] = t[idx] < tend[idx]
    return hops, band

for scheme in ("aligned", "centred"):
    t0_ = time.time()
    hops, band = march(scheme)
    m = ok
    print("%-8s hops per ray %.2f (p50 %.0f, p90 %.0f, p99 %.0f), flagged-brick passes %.2f   (%.0f s)" %
          (scheme, hops[m].mean(), *np.percentile(hops[m], [50, 90, 99]), band[m].mean(), time.time() - t0_))
