"""What a finer flag under the 4^3 cell bricks would be worth to the ray march, estimated outside the kernels: the bench scene's
volume after `frames` frames is read back, every hit ray is walked backwards from its hit sample by sample (torch gathers), and the
cells it crosses are classified as process_sample classifies them:
   clear cell brick (jump) / flagged cell brick + all-positive cell (a "cell test": 8 gathers for one cell) / mixed cell (evaluation)
against the same walk with one flag per 2^3 cells (tight: voxels [2c, 2c+2]^3; loose: the octant summary, voxels [2c, 2c+3]^3).
    python tools/dbg_subbrick_estimate.py [frames=40] [grid=512] [back=400]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, tsdf_amd
import torch.nn.functional as F
from tsdf_amd import synth
W, H = 640, 480
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
back = int(sys.argv[3]) if len(sys.argv) > 3 else 400
vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
for i in range(frames):
    d, cam = synth.depth_frame(i, 200, seed=0x5EED0003)
    f = d.copy(); bil.filter(f, W, H)
    vol.integrate(f, W, H, cam)
V, N = vol.raycast(W, H, cam)
trunc = vol.truncation_distance(); vs = 3000.0 / n
tau = 0.01 * trunc; step = float(np.float32(np.float64(np.float32(trunc)) * 0.05))
D = torch.from_numpy(vol.get_distance_data().reshape(n, n, n)).cuda()          # [z, y, x]
low = (~(D > tau)).float()[None, None]
pos = (-F.max_pool3d(-D[None, None], 2, 1))[0, 0] > 1e-30                          # cells (n-1)^3: all 8 corners positive
def pool(k, s, pad_hi):
    x = F.pad(low, (0, pad_hi, 0, pad_hi, 0, pad_hi), value=0.0)
    return F.max_pool3d(x, k, s)[0, 0] > 0
fine = F.max_pool3d(F.pad(low, (2, 2 + 4, 2, 2 + 4, 2, 2 + 4)), 8, 4)[0, 0] > 0    # brick grown by 2: voxels [4b-2, 4b+5]
cb = pool(5, 4, 4)          # cell brick: voxels [4b, 4b+4]
sb_t = pool(3, 2, 2)        # sub brick, tight: voxels [2c, 2c+2]
sb_l = pool(4, 2, 4)        # sub brick from the octant bits: voxels [2c, 2c+3]
print("grid", n, "frames", frames, "trunc %.3f voxels, step %.3f voxels" % (trunc / vs, step / vs))
print("bricks flagged: fine %.2f %%  cell brick %.2f %%; sub bricks: tight %.2f %%  loose %.2f %%" %
      (100 * fine.float().mean(), 100 * cb.float().mean(), 100 * sb_t.float().mean(), 100 * sb_l.float().mean()))
# rays
pose = cam.pose().reshape(4, 4).T.astype(np.float64); kinv = cam.kinv().reshape(3, 3).T.astype(np.float64)
ys, xs = np.mgrid[0:H, 0:W]
pix = np.stack([xs.ravel(), ys.ravel(), np.ones(W * H)], 0)
dirs = (pose[:3, :3] @ (kinv @ pix)).T
org = pose[:3, 3]
hit = ~np.isnan(V[:, 0])
Vh = torch.from_numpy(V[hit].astype(np.float64)).cuda(); dh = torch.from_numpy(dirs[hit]).cuda(); o = torch.from_numpy(org).cuda()
t_hit = ((Vh - o) * dh).sum(1) / (dh * dh).sum(1)
nr = Vh.shape[0]
def flag_at(a, ix, iy, iz):
    m = a.shape[0]
    ok = (ix >= 0) & (iy >= 0) & (iz >= 0) & (ix < m) & (iy < m) & (iz < m)
    r = torch.zeros_like(ix, dtype=torch.bool)
    r[ok] = a[iz[ok], iy[ok], ix[ok]]
    return r, ok
counts = {k: torch.zeros(nr, device="cuda") for k in ("cell_now", "cell_tight", "cell_loose", "jump_cb", "jump_sb_tight", "jump_sb_loose", "mixed", "band_samples")}
prev = {k: torch.full((nr, 3), -9, device="cuda", dtype=torch.long) for k in ("cell", "cb", "sbt", "sbl")}
for j in range(1, back + 1):
    p = o + (t_hit - j * step)[:, None] * dh                 # world; the volume's offset is 0
    fv = p / vs
    l = torch.floor(fv - 0.5).long(); b = torch.floor(fv).long() >> 2
    lx, ly, lz = l[:, 0], l[:, 1], l[:, 2]
    is_fine, ok = flag_at(fine, b[:, 0], b[:, 1], b[:, 2])
    ok = ok & (lx >= 0) & (ly >= 0) & (lz >= 0) & (lx < n - 1) & (ly < n - 1) & (lz < n - 1)
    band = is_fine & ok
    is_cb, _ = flag_at(cb, lx >> 2, ly >> 2, lz >> 2)
    is_t, _ = flag_at(sb_t, lx >> 1, ly >> 1, lz >> 1)
    is_l, _ = flag_at(sb_l, lx >> 1, ly >> 1, lz >> 1)
    is_pos, _ = flag_at(pos, lx, ly, lz)
    new_cell = (l != prev["cell"]).any(1); new_cb = ((l >> 2) != prev["cb"]).any(1); new_sb = ((l >> 1) != prev["sbt"]).any(1)
    prev["cell"] = l; prev["cb"] = l >> 2; prev["sbt"] = l >> 1
    counts["band_samples"] += band
    counts["jump_cb"] += band & ~is_cb & new_cb
    counts["cell_now"] += band & is_cb & is_pos & new_cell
    counts["mixed"] += band & is_cb & ~is_pos
    counts["jump_sb_tight"] += band & is_cb & ~is_t & new_sb
    counts["cell_tight"] += band & is_cb & is_t & is_pos & new_cell
    counts["jump_sb_loose"] += band & is_cb & ~is_l & new_sb
    counts["cell_loose"] += band & is_cb & is_l & is_pos & new_cell
print("hit rays %d; per hit ray, over the last %d samples before the hit:" % (nr, back))
for k, v in counts.items():
    q = np.percentile(v.cpu().numpy(), [50, 90, 99])
    print("  %-14s mean %.2f   p50 %.0f p90 %.0f p99 %.0f" % (k, v.mean().item(), *q))
m = {k: v.mean().item() for k, v in counts.items()}
print("passes in the flagged band (cell-brick jumps + cell tests + mixed samples, before look-ahead):")
print("  now          %.1f" % (m["jump_cb"] + m["cell_now"] + m["mixed"]))
print("  2^3 tight    %.1f" % (m["jump_cb"] + m["jump_sb_tight"] + m["cell_tight"] + m["mixed"]))
print("  2^3 loose    %.1f" % (m["jump_cb"] + m["jump_sb_loose"] + m["cell_loose"] + m["mixed"]))
