"""Randomised differential test: seeded random volumes, cameras, images and depth maps through integrate + ray cast on
the GPU and in the oracle, compared bit for bit.  The hand-written cases elsewhere aim at known corners; this sweep is
for the corners nobody thought of (the culling bounds, the rounding guard band, the skipping slack, the tail queue ...).
Sizes are small enough for the oracle to finish each case in well under a second on the GPU box's host cores."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import Cam, assert_same_floats

pytestmark = pytest.mark.gpu
_SEEN = {"scenes": 0, "updated": 0, "hit": 0}


def random_case(rng):
    dims = tuple(int(v) for v in rng.integers(5, 72, size=3))
    if rng.random() < 0.5:      # cubic voxels half of the time (anisotropic ones switch parts of the skipping off)
        vs = float(rng.uniform(4.0, 60.0))
        phys = tuple(d * vs for d in dims)
    else:
        phys = tuple(float(d * rng.uniform(4.0, 60.0)) for d in dims)
    width, height = int(rng.integers(1, 200)), int(rng.integers(1, 160))
    offset = tuple(float(v) for v in rng.uniform(-500, 500, size=3)) if rng.random() < 0.3 else None
    return dims, phys, width, height, offset


def random_camera(rng, dims, phys, offset, width, height):
    centre = np.array(phys) / 2.0 + (np.array(offset) if offset else 0.0)
    extent = float(max(phys))
    # position: inside the volume, just outside it, or far away
    mode = rng.integers(0, 3)
    radius = (0.2, 0.9, 2.5)[mode] * extent
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    pos = centre + d * radius * rng.uniform(0.5, 1.0)
    target = centre + rng.normal(size=3) * 0.15 * extent
    cam = tsdf_amd.Camera(float(rng.uniform(0.6, 1.4) * width + 20), float(rng.uniform(0.6, 1.4) * width + 20),
                          width / 2.0 + float(rng.uniform(-3, 3)), height / 2.0 + float(rng.uniform(-3, 3)))
    cam.move_to(*pos)
    cam.look_at(*target)
    return cam, float(np.linalg.norm(pos - centre))


def random_depth(rng, width, height, scale):
    kind = rng.integers(0, 4)
    n = width * height
    if kind == 0:
        d = np.full(n, scale, np.float64)
    elif kind == 1:
        d = scale * rng.uniform(0.3, 1.7, size=n)
    elif kind == 2:
        yy, xx = np.mgrid[0:height, 0:width]
        d = (scale * (0.6 + 0.5 * np.sin(xx / 7.0) * np.cos(yy / 5.0))).reshape(-1)
    else:
        d = scale * (1.0 + 0.02 * rng.standard_normal(n))
    d = np.clip(np.rint(d), 0, 65535)
    d[rng.random(n) < rng.choice([0.0, 0.02, 0.5])] = 0
    return d.astype(np.uint16)


@pytest.mark.parametrize("seed", range(40))
def test_random_scene(oracle, seed):
    rng = np.random.default_rng(0xF022 + seed)
    dims, phys, width, height, offset = random_case(rng)
    gv = tsdf_amd.TSDFVolume(dims, phys)
    ov = oracle.Volume(dims, phys)
    if offset is not None:
        gv.offset(*offset); gv.clear()
        ov.offset(*offset); ov.clear()
    threads = oracle.max_threads()
    cams = []
    for f in range(int(rng.integers(1, 4))):
        cam, dist_to_centre = random_camera(rng, dims, phys, offset, width, height)
        depth = random_depth(rng, width, height, max(dist_to_centre, 50.0))
        gv.integrate(depth, width, height, cam)
        ov.integrate(depth, width, height, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
        cams.append(cam)
        if rng.random() < 0.5:       # a ray cast between integrations exercises the flag refresh schedule
            gv.raycast(width, height, cam)
    what = "seed %d dims %s image %dx%d" % (seed, dims, width, height)
    assert_same_floats(gv.get_weight_data(), ov.weight, what + " weights")
    assert_same_floats(gv.get_distance_data(), ov.dist, what + " distances")
    for cam in cams[-2:]:
        V, N = gv.raycast(width, height, cam)
        Vo, No = ov.raycast(width, height, cam.pose(), cam.kinv(), nthreads=threads)
        assert_same_floats(V, Vo, what + " vertices")
        assert_same_floats(N, No, what + " normals")
    _SEEN["scenes"] += 1
    _SEEN["updated"] += int((ov.weight > 0).any())
    _SEEN["hit"] += int((~np.isnan(Vo[:, 0])).any())


@pytest.mark.parametrize("seed", range(10))
def test_axis_aligned_cameras_on_integer_principal_points(oracle, seed):
    """Cameras inside, on a face of and beside the volume that look exactly along an axis, principal point on a pixel: a row and a column
    of rays have a direction component of exactly 0.  For a ray that starts inside the volume the reference's ray_box then leaves an exit
    out of its minimum (NaN compares false, GPURaycaster.cu:197-251) and the ray is sampled far off the grid, where the clamped
    interpolation finds "surfaces"; beside the volume the zero component is a ray that runs along a face.  (Run once more with
    TSDF_RAY_CELLS=2 by the round's evidence runs: the cell-parallel cast walks such rays where it forms the ray records.)"""
    rng = np.random.default_rng(0xA815 + seed)
    dims = tuple(int(v) for v in rng.integers(12, 60, size=3))
    phys = tuple(float(v) for v in rng.uniform(400, 3000, size=3))
    width, height = int(rng.integers(8, 160)), int(rng.integers(8, 120))
    n = dims[0] * dims[1] * dims[2]
    gv = tsdf_amd.TSDFVolume(dims, phys)
    ov = oracle.Volume(dims, phys)
    trunc = gv.truncation_distance()
    # a smooth field with a few surfaces, some of them through the grid's faces
    zz, yy, xx = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
    c = rng.uniform(0.2, 0.8, size=3) * np.array(dims)
    r = np.sqrt((xx - c[0]) ** 2 + (yy - c[1]) ** 2 + (zz - c[2]) ** 2)
    D = np.clip((r - rng.uniform(0.15, 0.6) * min(dims)) * (phys[0] / dims[0]), -trunc, trunc).astype(np.float32)
    if rng.random() < 0.5:
        D = np.minimum(D, np.clip((zz - rng.uniform(0.1, 0.9) * dims[2]) * (phys[2] / dims[2]), -trunc, trunc).astype(np.float32))
    gv.set_distance_data(D.reshape(-1))
    ov.set_distance_data(D.reshape(-1))
    for _ in range(3):
        axis, sign = int(rng.integers(0, 3)), float(rng.choice([-1.0, 1.0]))
        where = rng.integers(0, 3)     # inside, on a face, outside
        pos = rng.uniform(0.1, 0.9, size=3) * np.array(phys)
        if where == 1:
            pos[int(rng.integers(0, 3))] = rng.choice([0.0, 1.0]) * phys[0]
        elif where == 2:
            pos[axis] = -sign * rng.uniform(0.05, 1.0) * phys[axis] + (phys[axis] if sign < 0 else 0.0)
        fwd = np.zeros(3); fwd[axis] = sign
        up = np.zeros(3); up[(axis + 1) % 3] = 1.0
        right = np.cross(up, fwd)
        M = np.eye(4)
        M[:3, 0], M[:3, 1], M[:3, 2], M[:3, 3] = right, up, fwd, pos
        cam = tsdf_amd.Camera(float(rng.uniform(0.4, 1.5) * width), float(rng.uniform(0.4, 1.5) * width), float(rng.integers(0, width)), float(rng.integers(0, height)))
        cam.set_pose_rows(M)
        V, N = gv.raycast(width, height, cam)
        Vo, No = ov.raycast(width, height, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
        what = "seed %d dims %s image %dx%d camera at %s along %s%d" % (seed, dims, width, height, pos, "+" if sign > 0 else "-", axis)
        assert_same_floats(V, Vo, what + " vertices")
        assert_same_floats(N, No, what + " normals")


def test_the_random_scenes_were_not_vacuous():
    if _SEEN["scenes"] < 20:
        pytest.skip("needs the whole sweep")
    assert _SEEN["updated"] >= _SEEN["scenes"] // 2 and _SEEN["hit"] >= _SEEN["scenes"] // 3, _SEEN


@pytest.mark.parametrize("seed", range(16))
def test_random_distance_fields(oracle, seed):
    """Arbitrary distance arrays (not produced by integrate): sign changes anywhere, NaNs, huge values."""
    rng = np.random.default_rng(0xD157 + seed)
    dims, phys, width, height, offset = random_case(rng)
    n = dims[0] * dims[1] * dims[2]
    gv = tsdf_amd.TSDFVolume(dims, phys)
    ov = oracle.Volume(dims, phys)
    if offset is not None:
        gv.offset(*offset)
        ov.offset(*offset)
    trunc = gv.truncation_distance()
    D = np.full(n, trunc, np.float32)
    k = int(rng.integers(1, max(2, n // 20)))
    idx = rng.choice(n, size=k, replace=False)
    D[idx] = (rng.uniform(-1.0, 1.0, size=k) * trunc * rng.choice([1e-6, 0.01, 1.0], size=k)).astype(np.float32)
    if rng.random() < 0.5:
        D[rng.choice(n, size=3, replace=False)] = [np.nan, 1e30, -1e30]
    gv.set_distance_data(D)
    ov.set_distance_data(D)
    for _ in range(2):
        cam, _ = random_camera(rng, dims, phys, offset, width, height)
        V, N = gv.raycast(width, height, cam)
        Vo, No = ov.raycast(width, height, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(V, Vo, "seed %d dims %s vertices" % (seed, dims))
        assert_same_floats(N, No, "seed %d normals" % seed)


@pytest.mark.parametrize("seed", range(12))
def test_smooth_distance_fields(oracle, seed):
    """Smooth fields of every magnitude with shallow zero crossings: where the ray march's look-ahead after an evaluated
    sample (a bound on how fast the interpolant can fall inside a cell) reaches furthest, and where its safety margin is
    all that separates a skipped sample from a hit."""
    rng = np.random.default_rng(0x5A00 + seed)
    dims, phys, width, height, offset = random_case(rng)
    gv = tsdf_amd.TSDFVolume(dims, phys)
    ov = oracle.Volume(dims, phys)
    if offset is not None:
        gv.offset(*offset)
        ov.offset(*offset)
    trunc = gv.truncation_distance()
    z, y, x = np.meshgrid(*(np.arange(d, dtype=np.float64) for d in dims[::-1]), indexing="ij")
    s = np.zeros(x.shape)
    for _ in range(3):
        f = rng.uniform(0.02, 0.6, size=3)          # radians per voxel: from half a grid to ten voxels per period
        s += np.sin(f[0] * x + f[1] * y + f[2] * z + rng.uniform(0, 6.28)) / 3.0
    amp = trunc * float(rng.choice([1e-5, 1e-3, 0.05, 1.0]))
    D = np.minimum(amp * (s + rng.uniform(0.1, 0.9)), trunc).astype(np.float32).ravel()
    gv.set_distance_data(D)
    ov.set_distance_data(D)
    for _ in range(2):
        cam, _ = random_camera(rng, dims, phys, offset, width, height)
        V, N = gv.raycast(width, height, cam)
        Vo, No = ov.raycast(width, height, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(V, Vo, "seed %d dims %s amplitude %g vertices" % (seed, dims, amp))
        assert_same_floats(N, No, "seed %d normals" % seed)


@pytest.mark.parametrize("seed", range(10))
def test_random_slab_splits_equal_the_whole_volume(seed):
    """Integrate + ray cast through 2..6 Z-slabs (owner-of-sample rule, min-k merge) == the whole volume, bit for bit."""
    import torch
    from tsdf_amd import multi
    rng = np.random.default_rng(0x51AB + seed)
    dims, phys, width, height, offset = random_case(rng)
    P = int(rng.integers(2, min(6, dims[2]) + 1))
    whole = tsdf_amd.TSDFVolume(dims, phys)
    slabs = [tsdf_amd.TSDFVolume(dims, phys, slab=multi.slab_range(dims[2], P, r)) for r in range(P)]
    cams = []
    for f in range(int(rng.integers(1, 3))):
        cam, dist_to_centre = random_camera(rng, dims, phys, None, width, height)
        depth = random_depth(rng, width, height, max(dist_to_centre, 50.0))
        for v in [whole] + slabs:
            v.integrate(depth, width, height, cam)
        cams.append(cam)
    Dw = whole.get_distance_data().reshape(dims[2], -1)
    rc = tsdf_amd.GPURaycaster(width, height)
    for cam in cams:
        V, N = whole.raycast(width, height, cam)
        hits = torch.empty((P, width * height, 2), dtype=torch.float32, device="cuda")
        for r, s in enumerate(slabs):
            lo, hi = s.resident_planes()
            assert_same_floats(s.get_distance_data().reshape(hi - lo, -1), Dw[lo:hi], "seed %d slab %d distances" % (seed, r))
            rc.raycast_slab_device(s, cam, hits[r].data_ptr())
            s.synchronize()
        Vm = torch.empty((width * height, 3), dtype=torch.float32, device="cuda")
        tsdf_amd.merge_hits_device(slabs[0], hits.data_ptr(), P, width, height, cam, Vm.data_ptr())
        torch.cuda.synchronize()
        assert_same_floats(Vm.cpu().numpy(), V, "seed %d: %d slabs of %s, image %dx%d" % (seed, P, dims, width, height))


@pytest.mark.parametrize("seed", range(12))
def test_random_bilateral_filters(oracle, seed):
    """Random image sizes (smaller than, equal to and larger than the tiles and the kernel), sigmas and contents, 8 and 16 bit."""
    rng = np.random.default_rng(0xB11A + seed)
    w, h = int(rng.integers(1, 90)), int(rng.integers(1, 70))
    sc, ss = float(rng.uniform(0.5, 40.0)), float(rng.uniform(0.4, 6.0))
    bil = tsdf_amd.BilateralFilter(sc, ss)
    for bits in (8, 16):
        hi = 256 if bits == 8 else int(rng.choice([300, 5000, 65536]))
        img = rng.integers(0, hi, size=w * h).astype(np.uint8 if bits == 8 else np.uint16)
        if rng.random() < 0.5:
            img[rng.random(w * h) < 0.1] = 0
        got = img.copy()
        bil.filter(got, w, h)
        exp = oracle.bilateral_u8(img, w, h, sc, ss) if bits == 8 else oracle.bilateral_u16(img, w, h, sc, ss, nthreads=4)
        assert np.array_equal(got, np.asarray(exp).reshape(-1)), "seed %d %dx%d sigma (%.2f, %.2f) %d bit" % (seed, w, h, sc, ss, bits)


@pytest.mark.parametrize("seed", range(6))
def test_random_icp_steps(oracle, seed):
    """Random plane / sphere scenes and poses through one ICP step: same inliers, sums within 1e-4 of the oracle's."""
    rng = np.random.default_rng(0x1C9 + seed)
    W, H = 640, 480
    yy, xx = np.mgrid[0:H, 0:W]
    def scene(shift):
        z = 1500.0 + 0.3 * (xx - 320 + shift[0]) + 0.2 * (yy - 240 + shift[1]) + 80.0 * np.sin((xx + shift[0]) / 40.0) * np.cos((yy + shift[1]) / 55.0)
        d = np.clip(np.rint(z + shift[2]), 0, 65535).astype(np.uint16)
        d[rng.random((H, W)) < 0.02] = 0
        return d.reshape(-1)
    d0, d1 = scene((0, 0, 0)), scene(tuple(rng.uniform(-2, 2, size=3)))
    icp = tsdf_amd.ICPOdometry(W, H, 331.0, 234.6, 591.1, 590.1)
    icp.init_icp_model(d0)
    icp.init_icp(d1)
    T = oracle.se3_exp(rng.normal(size=6) * np.array([0.004, 0.004, 0.004, 0.002, 0.002, 0.002]))
    R, t = T[:3, :3].astype(np.float32), T[:3, 3].astype(np.float32)
    level = int(rng.integers(0, 3))
    rows, cols, div = H >> level, W >> level, 1 << level
    A, b, res, inl = icp.estimate_step(level, R, t)
    maps = [icp.get_map(k, level) for k in ("vmap_curr", "nmap_curr", "vmap_prev", "nmap_prev")]
    import math
    ang = float(np.float32(math.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))
    Ao, bo, reso, inlo, _ = oracle.icp_step(R.T.reshape(-1), t, *maps, rows, cols, np.float32(591.1) / div, np.float32(590.1) / div,
                                            np.float32(331.0) / div, np.float32(234.6) / div, 0.10, ang)
    assert inl == inlo and inl > 100
    assert np.max(np.abs(A - Ao)) <= 1e-4 * np.max(np.abs(Ao))
    assert np.max(np.abs(b - bo)) <= 1e-4 * max(np.max(np.abs(bo)), 1e-6 * np.max(np.abs(Ao)))
    assert abs(res - reso) <= 1e-4 * max(reso, 1e-12)
