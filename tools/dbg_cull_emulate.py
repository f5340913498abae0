"""CPU emulation (numpy, float64; no GPU) of brick_cull_kernel's three tests on a frame of a workload, for the brick itself and
for the brick cut into children (a brick is kept when a child is): how many bricks / children survive, against the exact count
from the oracle's update mask.  Shows what conservativeness costs integrate at 1024^3.
    python tools/dbg_cull_emulate.py --grid 1024 --inside --frames 9"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as O
from tsdf_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--inside", action="store_true")
ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0004)
ap.add_argument("--stream-frames", type=int, default=100)
ap.add_argument("--frames", type=int, nargs="+", default=[9])
ap.add_argument("--tile", type=int, default=16)
ap.add_argument("--max-tiles", type=int, default=256)
ap.add_argument("--cone", action="store_true")
ap.add_argument("--exact", action="store_true", help="also the exact count from the oracle (slow at 1024^3)")
a = ap.parse_args()
n = a.grid
W, H = 640, 480
vs = 3000.0 / n
trunc = 1.1 * np.sqrt(3.0) * vs


def survivors(shape, f, cam, max_tiles, tile):
    """-> bool array [nz][ny][nx] over items of `shape` voxels: kept by tests (a), (b), (c)."""
    sx, sy, sz = shape
    ip = cam.inverse_pose().astype(np.float64).reshape(4, 4).T
    k = cam.k().astype(np.float64).reshape(3, 3).T
    nx, ny, nz = n // sx, n // sy, n // sz
    tiles_x, tiles_y = (W + tile - 1) // tile, (H + tile - 1) // tile
    img = f.reshape(H, W)
    pad = np.zeros((tiles_y * tile, tiles_x * tile), img.dtype); pad[:H, :W] = img
    tmax = pad.reshape(tiles_y, tile, tiles_x, tile).max(axis=(1, 3)).astype(np.float64)
    # summed-area style max is not available: use a sparse-table (2D range max) over tiles
    import math
    LX, LY = int(math.log2(tiles_x)) + 1, int(math.log2(tiles_y)) + 1
    st = {(0, 0): tmax}
    for j in range(LY):
        for i in range(LX):
            if (i, j) == (0, 0): continue
            if i > 0:
                p = st[(i - 1, j)]; h = 1 << (i - 1)
                q = p.copy(); q[:, :-h] = np.maximum(p[:, :-h], p[:, h:]); st[(i, j)] = q
            else:
                p = st[(i, j - 1)]; h = 1 << (j - 1)
                q = p.copy(); q[:-h, :] = np.maximum(p[:-h, :], p[h:, :]); st[(i, j)] = q
    def range_max(tx0, tx1, ty0, ty1):
        lx = np.floor(np.log2(tx1 - tx0 + 1)).astype(int); ly = np.floor(np.log2(ty1 - ty0 + 1)).astype(int)
        out = np.zeros(tx0.shape)
        for i in range(LX):
            for j in range(LY):
                m = (lx == i) & (ly == j)
                if not m.any(): continue
                t = st[(i, j)]
                x0, x1, y0, y1 = tx0[m], tx1[m] - (1 << i) + 1, ty0[m], ty1[m] - (1 << j) + 1
                out[m] = np.maximum(np.maximum(t[y0, x0], t[y0, x1]), np.maximum(t[y1, x0], t[y1, x1]))
        return out
    bz, by, bx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    Lmax = [np.full(bx.shape, -np.inf) for _ in range(4)]; Lmin = [np.full(bx.shape, np.inf) for _ in range(4)]
    qx_lo = np.full(bx.shape, np.inf); qx_hi = -qx_lo.copy(); qy_lo = qx_lo.copy(); qy_hi = qx_hi.copy(); cz_lo = qx_lo.copy()
    allpos = np.ones(bx.shape, bool); allneg = np.ones(bx.shape, bool)
    for c in range(8):
        vx = bx * sx + (sx - 1 if c & 1 else 0); vy = by * sy + (sy - 1 if c & 2 else 0); vz = bz * sz + (sz - 1 if c & 4 else 0)
        p = np.stack([(vx + 0.5) * vs, (vy + 0.5) * vs, (vz + 0.5) * vs, np.ones(vx.shape)], -1)
        camp = p @ ip.T
        im = camp[..., :3] @ k.T
        allpos &= im[..., 2] > 1e-3; allneg &= im[..., 2] < -1e-3
        for i, L in enumerate((im[..., 0] + im[..., 2], W * im[..., 2] - im[..., 0], im[..., 1] + im[..., 2], H * im[..., 2] - im[..., 1])):
            Lmax[i] = np.maximum(Lmax[i], L); Lmin[i] = np.minimum(Lmin[i], L)
        with np.errstate(divide="ignore", invalid="ignore"):
            qx, qy = im[..., 0] / im[..., 2], im[..., 1] / im[..., 2]
        qx_lo = np.minimum(qx_lo, qx); qx_hi = np.maximum(qx_hi, qx); qy_lo = np.minimum(qy_lo, qy); qy_hi = np.maximum(qy_hi, qy)
        cz_lo = np.minimum(cz_lo, camp[..., 2])
    allpos = allpos | allneg   # the divisor keeps one sign over the item
    keep = np.ones(bx.shape, bool)
    if a.cone:   # straddlers: no voxel inside the double cone of the image's four side planes
        front_out = (Lmax[0] < -1e-2) | (Lmax[1] < -1e-2) | (Lmax[2] < -1e-2) | (Lmax[3] < -1e-2)
        back_out = (Lmin[0] > 1e-2) | (Lmin[1] > 1e-2) | (Lmin[2] > 1e-2) | (Lmin[3] > 1e-2)
        keep[~allpos & front_out & back_out] = False
    off = (qx_hi < -1) | (qx_lo > W) | (qy_hi < -1) | (qy_lo > H)
    keep[allpos & off] = False
    fx0 = np.maximum(qx_lo - 1, 0); fx1 = np.minimum(qx_hi + 1, W - 1); fy0 = np.maximum(qy_lo - 1, 0); fy1 = np.minimum(qy_hi + 1, H - 1)
    boxed = allpos & keep & (fx0 <= fx1) & (fy0 <= fy1)
    tx0 = np.where(boxed, fx0, 0).astype(int) // tile; tx1 = np.where(boxed, fx1, 0).astype(int) // tile
    ty0 = np.where(boxed, fy0, 0).astype(int) // tile; ty1 = np.where(boxed, fy1, 0).astype(int) // tile
    ntile = (tx1 - tx0 + 1) * (ty1 - ty0 + 1)
    tested = boxed & (ntile <= max_tiles)
    dmax = range_max(tx0, tx1, ty0, ty1)
    keep[tested & (dmax == 0)] = False
    keep[tested & (cz_lo - dmax > trunc * 1.0001 + 2e-3)] = False
    big = boxed & keep & ((np.where(boxed, fx1 - fx0 + 2, 0).astype(int)) * (np.where(boxed, fy1 - fy0 + 1, 0).astype(int)) > 8192)
    area = (np.where(boxed, fx1 - fx0 + 2, 0).astype(int)) * (np.where(boxed, fy1 - fy0 + 1, 0).astype(int))
    if shape == (64, 4, 32):
        aa = area[boxed & keep]
        print("    box pixels of kept one-sign bricks: percentiles 50/90/95/99/max", [int(np.percentile(aa, q)) for q in (50, 90, 95, 99, 100)],
              " > 8192: %d, > 12288: %d, > 16384: %d, > 24576: %d" % tuple(int((aa > t).sum()) for t in (8192, 12288, 16384, 24576)))
    return keep, big, (allpos == False) & keep


for fi in a.frames:
    d, cam = synth.depth_frame(fi, a.stream_frames, seed=a.seed, inside=a.inside)
    f = O.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=O.max_threads()).reshape(-1)
    print("frame", fi)
    for mt in [a.max_tiles]:
        for tile in [a.tile]:
            for shape in [(64, 4, 32)]:
                keep, big, strad = survivors(shape, f, cam, mt, tile)
                sx, sy, sz = shape
                # bricks kept because one of their children is
                kb = keep.reshape(n // 32, 32 // sz, n // 4, 4 // sy, n // 64, 64 // sx).any(axis=(1, 3, 5))
                print("  tiles<=%-7d tile %2d  child %2dx%dx%-2d: %7d children kept (%6.1f M voxels), %6d bricks kept; %5d children with a box > 8192 px, %5d straddle the camera plane"
                      % (mt, tile, sx, sy, sz, int(keep.sum()), keep.sum() * sx * sy * sz / 1e6, int(kb.sum()), int(big.sum()), int(strad.sum())), flush=True)
