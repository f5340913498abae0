"""GPU parity: GPURaycaster.raycast (process_ray + compute_normals in HIP, through the C ABI) against the CPU
oracle.  north_star's tolerance is 1e-4 relative with an identical NaN mask; the assertion here is bit-exact
vertices and normals (the kernels keep the reference's operation order, fp contraction off)."""
import os

import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, Cam, assert_same_floats, camera_at, sphere_tsdf
from tsdf_amd import synth

pytestmark = pytest.mark.gpu


def volumes_with(oracle, size, physical, dist, offset=None):
    gv = tsdf_amd.TSDFVolume(size, physical)
    ov = oracle.Volume(size, physical)
    if offset is not None:
        gv.offset(*offset)
        ov.offset(*offset)
    gv.set_distance_data(dist)
    ov.set_distance_data(dist)
    return gv, ov


def compare(oracle, gv, ov, cam, width=W, height=H, what=""):
    V, N = gv.raycast(width, height, cam)
    Vo, No = ov.raycast(width, height, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    assert np.array_equal(np.isnan(V), np.isnan(Vo)), what + ": NaN masks differ"
    assert_same_floats(V, Vo, what + " vertices")
    assert_same_floats(N, No, what + " normals")
    return V, N


@pytest.mark.parametrize("cam_pos", [(450, 150, 150), (-150, 150, 450)])
def test_analytic_sphere_from_the_reference_test_poses(oracle, cam_pos):
    # Test_TSDF_RayCast.cpp:413-443 / :565-594: sphere TSDF in a 256 mm cube seen from two poses (the
    # reference renders 300^3 voxels to a PNG without assertions; 96^3 keeps the oracle fast)
    n, phys = 96, 256.0
    dist = sphere_tsdf(oracle, n, phys, 80.0)
    gv, ov = volumes_with(oracle, (n, n, n), (phys,) * 3, dist)
    cam = camera_at(cam_pos, look_at=(128, 128, 128))
    V, _ = compare(oracle, gv, ov, cam, what="sphere from %s" % (cam_pos,))
    hits = ~np.isnan(V[:, 0])
    assert hits.sum() > 1000
    r = np.linalg.norm(V[hits].astype(np.float64) - 128.0, axis=1)
    assert np.all(np.abs(r - 80.0) < 2.0 * gv.truncation_distance())


def test_integrated_scene_rendered_from_the_first_pose(oracle):
    n = 96
    frames = [synth.depth_frame(i, 4, seed=0x5EED0002) for i in range(4)]
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    for d, cam in frames:
        gv.integrate(d, W, H, cam)
        ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
    V, N = compare(oracle, gv, ov, frames[0][1], what="integrated scene")
    hits = ~np.isnan(V[:, 0])
    assert hits.mean() > 0.5
    finite_normals = np.isfinite(N).all(axis=1) & (np.abs(N).sum(axis=1) > 0)
    assert np.allclose(np.linalg.norm(N[finite_normals], axis=1), 1.0, atol=1e-5)


def test_camera_inside_the_volume_and_rays_that_miss(oracle):
    n = 64
    dist = sphere_tsdf(oracle, n, 3000.0, 600.0)
    gv, ov = volumes_with(oracle, (n, n, n), (3000.0,) * 3, dist)
    inside = camera_at((1500, 1500, 1500), yaw_pitch_roll=(0.4, 0.1, -0.3))   # inside the sphere: every sample <= 0
    compare(oracle, gv, ov, inside, what="camera inside")
    away = camera_at((1500, 1500, -2000), look_at=(1500, 9000, -2000))        # looks past the volume
    V, _ = compare(oracle, gv, ov, away, what="camera looking away")
    assert np.isnan(V).all()
    grazing = camera_at((-500, 1500, 1500), look_at=(3000, 1600, 1500))
    compare(oracle, gv, ov, grazing, what="grazing")


def test_offset_volume_and_anisotropic_voxels(oracle):
    n = (40, 56, 72)
    phys = (2000.0, 2400.0, 2800.0)
    ov0 = oracle.Volume(n, phys)
    rng = np.random.RandomState(4)
    trunc = np.float32(ov0.truncation_distance())
    zz = (np.arange(n[2], dtype=np.float32) + 0.5)[:, None, None] * ov0.voxel_size()[2]
    plane = np.clip(np.float32(1400.0) - zz + rng.uniform(-5, 5, (n[2], n[1], n[0])).astype(np.float32), -trunc, trunc)
    gv, ov = volumes_with(oracle, n, phys, plane.reshape(-1), offset=(-300.0, 250.0, 100.0))
    cam = camera_at((700, 1400, -900), look_at=(700, 1450, 1500))
    V, _ = compare(oracle, gv, ov, cam, what="offset volume")
    assert (~np.isnan(V[:, 0])).sum() > 10000


def test_small_odd_image_sizes(oracle):
    n = 32
    dist = sphere_tsdf(oracle, n, 512.0, 150.0)
    gv, ov = volumes_with(oracle, (n, n, n), (512.0,) * 3, dist)
    base = camera_at((256, 256, -400))
    for (w, h) in ((1, 1), (17, 9), (100, 75)):
        k = base.k().copy()
        k[0] *= w / 640.0; k[4] *= h / 480.0; k[6] = w / 2.0; k[7] = h / 2.0
        cam = Cam(base.pose(), base.inverse_pose(), k, oracle.mat3_inverse(k))
        compare(oracle, gv, ov, cam, width=w, height=h, what="%dx%d" % (w, h))


def test_alternating_image_sizes_on_one_volume(oracle):
    """The per-pixel first-hit words live in the volume handle, double buffered across ray casts: casting a large, a
    small and again the large image from different poses must not see each other's leftovers."""
    n = 48
    dist = sphere_tsdf(oracle, n, 512.0, 150.0)
    gv, ov = volumes_with(oracle, (n, n, n), (512.0,) * 3, dist)
    poses = [camera_at((256, 256, -400)), camera_at((-300, 200, 100), look_at=(256, 256, 256)), camera_at((256, 700, 900), look_at=(256, 256, 256))]
    for rnd, (w, h) in enumerate(((160, 120), (33, 20), (160, 120), (160, 120), (8, 8), (33, 20), (160, 120))):
        base = poses[rnd % len(poses)]
        k = base.k().copy()
        k[0] *= w / 640.0; k[4] *= h / 480.0; k[6] = w / 2.0; k[7] = h / 2.0
        cam = Cam(base.pose(), base.inverse_pose(), k, oracle.mat3_inverse(k))
        compare(oracle, gv, ov, cam, width=w, height=h, what="cast %d at %dx%d" % (rnd, w, h))
        if rnd == 3:       # vertices only in between (the other resolve kernel)
            Vv = tsdf_amd.GPURaycaster(w, h).get_vertices(gv, cam)
            Vo, _ = ov.raycast(w, h, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
            assert_same_floats(Vv, Vo, "vertices only, cast %d" % rnd)


def test_ray_timeout_cap_of_4402_samples(oracle):
    # Q8: with an empty (all +trunc) grid every ray marches until the far face or the 4402-sample cap; at
    # 448^3 / 3000 mm the step is 0.638 mm, so the cap (2808 mm of camera-z) comes first
    n = 448
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    cam = camera_at((1500, 1500, -200))
    k = cam.k().copy(); k[0] /= 10; k[4] /= 10; k[6] = 32; k[7] = 24
    small = Cam(cam.pose(), cam.inverse_pose(), k, oracle.mat3_inverse(k))
    V, _ = compare(oracle, gv, ov, small, width=64, height=48, what="empty volume")
    assert np.isnan(V).all()
    st = tsdf_amd.GPURaycaster(64, 48).stats(gv, small)
    _, _, so = ov.raycast(64, 48, small.pose(), small.kinv(), nthreads=oracle.max_threads(), stats=True)
    assert st["samples"] == so["samples"] and st["hits"] == 0
    assert so["sample_count"].max() == 4402


def test_roofline_counters_match_the_oracle(oracle):
    n = 64
    d, cam = synth.depth_frame(0, 4, seed=8)
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    gv.integrate(d, W, H, cam)
    ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv())
    st = tsdf_amd.GPURaycaster(W, H).stats(gv, cam)
    _, _, so = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads(), stats=True)
    assert st["samples"] == so["samples"]
    assert st["touched"] == so["touched"]
    assert st["hits"] == so["hits"]


def test_get_vertices_only_and_normals_kernel_alone(oracle):
    n = 48
    dist = sphere_tsdf(oracle, n, 1000.0, 300.0)
    gv, ov = volumes_with(oracle, (n, n, n), (1000.0,) * 3, dist)
    cam = camera_at((500, 500, -800))
    V = tsdf_amd.GPURaycaster(W, H).get_vertices(gv, cam)
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    assert_same_floats(V, Vo, "get_vertices")
    # last row / last column are zero, NaN elsewhere when a neighbour missed (Q11)
    N = No.reshape(H, W, 3)
    assert np.all(N[-1] == 0) and np.all(N[:, -1] == 0)


def test_slab_hit_records_merge_to_the_single_volume_result(oracle):
    """Two Z-slabs on one GPU: per-slab hit records + merge == whole-volume raycast (and == oracle slabs)."""
    import ctypes as C
    from tsdf_amd._capi import lib, check
    n = 64
    frames = [synth.depth_frame(i, 3, seed=77) for i in range(3)]
    whole = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    slabs = [tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000), slab=s) for s in ((0, 20), (20, 41), (41, 64))]
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    for d, cam in frames:
        whole.integrate(d, W, H, cam)
        ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
        for s in slabs:
            s.integrate(d, W, H, cam)
    cam = frames[0][1]
    Vw, Nw = whole.raycast(W, H, cam)
    import torch
    hits = torch.empty((len(slabs), W * H, 2), dtype=torch.float32, device="cuda")     # {k, t} records
    rc = tsdf_amd.GPURaycaster(W, H)
    for i, s in enumerate(slabs):
        rc.raycast_slab_device(s, cam, hits[i].data_ptr())
        s.synchronize()
        lo, hi = s.owned_planes()
        slo, shi = s.resident_planes()
        os_ = oracle.Volume((n, n, n), (3000, 3000, 3000), z_store=(slo, shi))
        os_.set_distance_data(ov.dist.reshape(n, -1)[slo:shi])
        ho = os_.raycast_slab(W, H, cam.pose(), cam.kinv(), (lo, hi), nthreads=oracle.max_threads())
        assert np.array_equal(hits[i].cpu().numpy().view(np.uint32), ho), "slab %d records {k, t} differ from the oracle's" % i
    V = torch.empty((W * H, 3), dtype=torch.float32, device="cuda")
    Nn = torch.empty_like(V)
    tsdf_amd.merge_hits_device(slabs[0], hits.data_ptr(), len(slabs), W, H, cam, V.data_ptr())
    tsdf_amd.compute_normals_device(W, H, V.data_ptr(), Nn.data_ptr())
    torch.cuda.synchronize()
    assert_same_floats(V.cpu().numpy(), Vw, "merged vertices")
    assert_same_floats(Nn.cpu().numpy(), Nw, "merged normals")
    V2, N2 = torch.zeros_like(V), torch.zeros_like(V)          # the same in one launch
    tsdf_amd.merge_hits_normals_device(slabs[1], hits.data_ptr(), len(slabs), W, H, cam, V2.data_ptr(), N2.data_ptr())
    torch.cuda.synchronize()
    assert_same_floats(V2.cpu().numpy(), Vw, "merged vertices (one launch)")
    assert_same_floats(N2.cpu().numpy(), Nw, "merged normals (one launch)")
    # the oracle's own merge of the records: the protocol, restated on the CPU
    Vo = ov.merge_hits(hits.cpu().numpy(), W, H, cam.pose(), cam.kinv())
    assert_same_floats(Vo, Vw, "oracle merge of the records")


# ---------------------------------------------------------------------------------------------------------
# Exact empty-space skipping (brick occupancy): adversarial volumes.  The ray caster may only skip samples
# that the reference's march would have passed without a hit, so every case must stay bit-identical.

def _speck_volume(oracle, n, rng, n_specks, values):
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    trunc = np.float32(ov.truncation_distance())
    d = np.full((n, n, n), trunc, np.float32)
    zs, ys, xs = (rng.randint(0, n, n_specks) for _ in range(3))
    d[zs, ys, xs] = rng.choice(np.asarray(values, np.float32) * trunc, n_specks)
    return d.reshape(-1)


@pytest.mark.parametrize("values", [(-1.0, -0.5), (0.0,), (0.005, 0.0099, 1e-6), (0.0101, 0.02, 0.5)])
def test_skipping_isolated_voxels_at_brick_corners_and_faces(oracle, values):
    # single voxels that are negative / zero / barely positive / just above the occupancy threshold, placed at
    # random (hence also on brick faces, edges and corners): neighbours within two voxels must not be skipped
    n = 64
    rng = np.random.RandomState(hash(values) & 0xFFFF)
    dist = _speck_volume(oracle, n, rng, 400, values)
    gv, ov = volumes_with(oracle, (n, n, n), (3000.0,) * 3, dist)
    for cam in (camera_at((1500, 1500, -800)), camera_at((-700, 900, 500), look_at=(1500, 1500, 1500)),
                camera_at((1400, 1600, 1450), yaw_pitch_roll=(1.1, 0.3, -0.2))):
        compare(oracle, gv, ov, cam, what="specks %s" % (values,))


def test_skipping_thin_oblique_sheet_and_noise(oracle):
    n = 96
    ov0 = oracle.Volume((n, n, n), (3000, 3000, 3000))
    vs, trunc = ov0.voxel_size(), np.float32(ov0.truncation_distance())
    c = (np.arange(n, dtype=np.float32) + 0.5) * vs[0]
    zz, yy, xx = np.meshgrid(c, c, c, indexing="ij")
    nrm = np.array([0.31, -0.22, 0.92], np.float32)
    nrm /= np.linalg.norm(nrm)
    sd = (xx * nrm[0] + yy * nrm[1] + zz * nrm[2]) - np.float32(1650.0)
    rng = np.random.RandomState(12)
    sd = -sd + rng.uniform(-0.2, 0.2, sd.shape).astype(np.float32) * trunc * (np.abs(sd) < 3 * trunc)
    dist = np.clip(sd, -trunc, trunc).astype(np.float32)
    dist[np.abs(sd) > 1.5 * trunc] = trunc          # a thin shell: free space on both sides
    gv, ov = volumes_with(oracle, (n, n, n), (3000.0,) * 3, dist.reshape(-1))
    for cam in (camera_at((1500, 1500, -900)), camera_at((2900, 200, 300), look_at=(1200, 1700, 1900))):
        V, _ = compare(oracle, gv, ov, cam, what="oblique sheet")
        assert (~np.isnan(V[:, 0])).sum() > 50000


def test_skipping_fully_random_volume(oracle):
    n = 48
    rng = np.random.RandomState(99)
    ov0 = oracle.Volume((n, n, n), (3000, 3000, 3000))
    dist = rng.uniform(-1, 1, n ** 3).astype(np.float32) * np.float32(ov0.truncation_distance())
    dist[rng.rand(n ** 3) < 0.9] = np.float32(ov0.truncation_distance())
    gv, ov = volumes_with(oracle, (n, n, n), (3000.0,) * 3, dist)
    compare(oracle, gv, ov, camera_at((300, 2500, -400), look_at=(1500, 1500, 1500)), what="random volume")


def test_skipping_flags_follow_integrate_clear_and_upload(oracle):
    n = 80
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    d0, cam0 = synth.depth_frame(0, 6, seed=41)
    gv.integrate(d0, W, H, cam0)
    ov.integrate(d0, W, H, cam0.inverse_pose(), cam0.k(), cam0.kinv(), nthreads=oracle.max_threads())
    compare(oracle, gv, ov, cam0, what="after first integrate")
    # clear, integrate a different frame from another pose: stale flags must not survive
    gv.clear()
    ov.clear()
    d1, cam1 = synth.depth_frame(3, 6, seed=41)
    gv.integrate(d1, W, H, cam1)
    ov.integrate(d1, W, H, cam1.inverse_pose(), cam1.k(), cam1.kinv(), nthreads=oracle.max_threads())
    compare(oracle, gv, ov, cam1, what="after clear + integrate")
    # whole-array upload replaces everything
    dist = sphere_tsdf(oracle, n, 3000.0, 500.0)
    gv.set_distance_data(dist)
    ov.set_distance_data(dist)
    compare(oracle, gv, ov, cam0, what="after upload")
    # and integrating on top of an uploaded volume keeps marking bricks
    gv.integrate(d0, W, H, cam0)
    ov.integrate(d0, W, H, cam0.inverse_pose(), cam0.k(), cam0.kinv(), nthreads=oracle.max_threads())
    compare(oracle, gv, ov, cam1, what="upload + integrate")


def test_skipping_anisotropic_voxels_disable_or_keep_exactness(oracle):
    # step = 0.055*|voxel_size| can exceed a quarter of the smallest voxel edge: skipping must switch itself off
    n = (96, 96, 24)
    phys = (600.0, 600.0, 2400.0)
    ov0 = oracle.Volume(n, phys)
    trunc = np.float32(ov0.truncation_distance())
    zz = (np.arange(n[2], dtype=np.float32) + 0.5)[:, None, None] * ov0.voxel_size()[2]
    d = np.clip(np.float32(1300.0) - zz, -trunc, trunc) * np.ones((n[2], n[1], n[0]), np.float32)
    gv, ov = volumes_with(oracle, n, phys, d.reshape(-1))
    compare(oracle, gv, ov, camera_at((300, 300, -500)), what="anisotropic z-coarse")
    compare(oracle, gv, ov, camera_at((-400, 300, 1200), look_at=(300, 300, 1250)), what="anisotropic side view")


def test_sphere_golden_vectors(oracle):
    """tests/golden/oracle_sphere_raycast.npz (generated by tests/golden/make_golden.py): stored inputs -> stored outputs."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_sphere_raycast.npz"))
    for tag in ("a", "b"):
        gv = tsdf_amd.TSDFVolume((64, 64, 64), (256.0, 256.0, 256.0))
        gv.set_distance_data(f[tag + "_dist"])
        pose = f[tag + "_pose"]
        k, _ = oracle.camera_k(591.1 / 4, 590.1 / 4, 331.0 / 4, 234.6 / 4)
        cam = Cam(pose, oracle.mat4_inverse(pose), k, f[tag + "_kinv"])
        V, N = gv.raycast(160, 120, cam)
        assert_same_floats(V, f[tag + "_vertices"], "golden sphere vertices " + tag)
        assert_same_floats(N, f[tag + "_normals"], "golden sphere normals " + tag)


_PROBE = r"""
import sys, numpy as np
import tsdf_amd
from tsdf_amd import synth
n = 96
gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
for i in range(3):
    d, cam = synth.depth_frame(i, 4, seed=0x5EED0002)
    f = d.copy(); bil.filter(f, synth.WIDTH, synth.HEIGHT)
    gv.integrate(f, synth.WIDTH, synth.HEIGHT, cam)
    if i == 1:
        gv.raycast(synth.WIDTH, synth.HEIGHT, cam)      # a ray cast between integrations (flag refresh schedule)
d, cam = synth.depth_frame(0, 4, seed=0x5EED0002)
V, N = gv.raycast(synth.WIDTH, synth.HEIGHT, cam)
np.savez(sys.argv[1], V=V, N=N, D=gv.get_distance_data())
"""


@pytest.mark.parametrize("env", [
    {"TSDF_RAY_SEGMENTS": "1", "TSDF_RAY_TRIP_BUDGET": "1", "TSDF_RAY_TAIL_LANES": "64"},     # everything in the tail kernel
    {"TSDF_RAY_SEGMENTS": "1", "TSDF_RAY_TRIP_BUDGET": "1", "TSDF_RAY_TAIL_PIECE": "1"},      # ... in the shortest pieces
    {"TSDF_RAY_SEGMENTS": "2", "TSDF_RAY_TRIP_BUDGET": "3", "TSDF_RAY_TAIL_PIECE": "100000"}, # ... never cut
    {"TSDF_RAY_SEGMENTS": "2", "TSDF_RAY_TRIP_BUDGET": "2", "TSDF_RAY_TAIL_LANES": "1", "TSDF_RAY_TAIL_PIECE": "7"},
    {"TSDF_RAY_SEGMENTS": "8", "TSDF_RAY_TRIP_BUDGET": "24", "TSDF_RAY_TAIL_LANES": "16", "TSDF_RAY_TAIL_PIECE": "256"},  # round 1g
    {"TSDF_RAY_SEGMENTS": "3", "TSDF_RAY_TRIP_BUDGET": "2", "TSDF_RAY_TAIL_LANES": "4", "TSDF_RAY_TAIL_GRID": "7"},
    {"TSDF_RAY_SEGMENTS": "16", "TSDF_RAY_TRIP_BUDGET": "100000"},                             # nothing in the tail kernel
    {"TSDF_RAY_SEGMENTS": "64", "TSDF_RAY_TRIP_BUDGET": "5", "TSDF_RAY_TAIL_LANES": "8"},
    {"TSDF_RAY_RANGE_ORDER": "0"},                                                             # sample ranges dispatched near to far (round 1)
    {"TSDF_RAY_RANGE_ORDER": "2", "TSDF_RAY_SEGMENTS": "5"},                                   # ... last, first, then far to near
    {"TSDF_RAY_LEARNED_ORDER": "0"},                                                           # tiles in launch order (no order learnt from the previous cast)
    {"TSDF_RAY_TILE_MAP": "0", "TSDF_RAY_TRIP_BUDGET": "4"},                                   # every eighth tile per XCD; many long waves to order
    {"TSDF_RAY_TILE_MAP": "1"},                                                                # one contiguous eighth of the image per XCD (round 2)
    {"TSDF_DEBUG_SORT": "1"},                                                                  # integrate: brick list in index order
    {"TSDF_DEBUG_SORT": "2"},                                                                  # ... scattered
    {"TSDF_INT_GRID_PER_CU": "3"},                                                             # integrate: resident grid walking the brick list
    {"TSDF_OCC_REBUILD_PERIOD": "0"},                                                          # sticky flags only
    {"TSDF_OCC_REBUILD_PERIOD": "1"},                                                          # flags rebuilt every frame
    {"TSDF_RAY_ENTRY_BOUND": "0"},                                                             # no per-tile entry bound (round 4)
    {"TSDF_RAY_ENTRY_BOUND": "0", "TSDF_RAY_LEARNED_ORDER": "0"},
    {"TSDF_RAY_CELLS": "0"},                                                                   # the march kernels, whatever the view (round 5)
    {"TSDF_RAY_CELLS": "2"},                                                                   # the cell-parallel cast wherever the view allows it
    {"TSDF_RAY_CELLS": "2", "TSDF_RAY_CELLS_GRID": "3"},                                       # ... every wave through many bricks
    {"TSDF_RAY_CELLS": "2", "TSDF_RAY_CELLS_PAIRS": "64"},                                     # ... each brick in many parts
    {"TSDF_RAY_CELLS": "2", "TSDF_RAY_CELLS_LOOK": "0"},                                       # ... the bricks not projected when listed (none dropped, never in parts)
])
def test_schedule_knobs_do_not_change_a_bit(oracle, tmp_path, env):
    """How the march is cut into sample ranges, passes and lane groups, and when the occupancy flags are refreshed, is
    scheduling only: every setting must give the oracle's image."""
    import os
    import subprocess
    import sys
    out = str(tmp_path / "probe.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _PROBE, out], check=True, env=e, cwd=root, timeout=600)
    got = np.load(out)
    n = 96
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    if "TSDF_INT_GRID_PER_CU" in env:                   # the integrate knob: the distances themselves against the oracle
        oi = oracle.Volume((n, n, n), (3000, 3000, 3000))
        ob = oracle.BilateralFilter(30.0, 4.5) if hasattr(oracle, "BilateralFilter") else None
        bil = tsdf_amd.BilateralFilter(30.0, 4.5)
        for i in range(3):
            d, c = synth.depth_frame(i, 4, seed=0x5EED0002)
            f = d.copy(); bil.filter(f, synth.WIDTH, synth.HEIGHT)      # (the filter's parity is covered elsewhere)
            oi.integrate(f, synth.WIDTH, synth.HEIGHT, c.inverse_pose(), c.k(), c.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(got["D"], oi.dist, "distances with %s" % env)
    ov.set_distance_data(got["D"])                      # (integrate parity is covered elsewhere)
    _, cam = synth.depth_frame(0, 4, seed=0x5EED0002)
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    assert np.array_equal(np.isnan(got["V"]), np.isnan(Vo))
    assert_same_floats(got["V"], Vo, "vertices with %s" % env)
    assert_same_floats(got["N"], No, "normals with %s" % env)


@pytest.mark.parametrize("case", ["positive_taps_that_extrapolate_below_zero", "flat_rim_with_a_surface_inside", "almost_flat", "rim_written_by_integrate"])
def test_bricks_at_the_grid_boundary_are_skipped_only_when_their_voxels_are_flat(oracle, case):
    """In the outer half-voxel shell the reference extrapolates (Q10): with taps 0.1 and 0.9 x trunc a sample half a voxel outside the
    first voxel centre is 1.5 * 0.1 - 0.5 * 0.9 < 0 -- a hit among positive voxels.  Bricks at the boundary are therefore clear for the
    ray caster only when every voxel in reach is FLAT (common.hpp: OccGrid); whatever the flags say, every ray must end where the
    oracle's does."""
    n, phys = 64, 640.0
    rng = np.random.default_rng(11)
    probe = tsdf_amd.TSDFVolume((n, n, n), (phys,) * 3)
    trunc = np.float32(probe.truncation_distance())
    probe.close()
    D = np.full((n, n, n), trunc, np.float32)              # [z, y, x]
    if case == "positive_taps_that_extrapolate_below_zero":
        D[:, :, 0] = 0.1 * trunc; D[:, :, 1] = 0.9 * trunc         # x = 0 face
        D[0, :, :] = 0.05 * trunc; D[1, :, :] = 0.6 * trunc        # z = 0 face
        D[:, n - 1, :] = 0.2 * trunc; D[:, n - 2, :] = 0.99 * trunc  # far y face (upper tap clamped: no extrapolation there)
    elif case == "flat_rim_with_a_surface_inside":
        D[20:44, 20:44, 30] = -0.3 * trunc; D[20:44, 20:44, 29] = 0.4 * trunc
    elif case == "almost_flat":
        # inside the band everywhere but for a few voxels just outside it, some in the rim, some not
        D[:] = (np.float32(0.94) + np.float32(0.06) * rng.random(D.shape, dtype=np.float32)) * trunc
        for (z, y, x), f in zip(rng.integers(0, n, size=(40, 3)), rng.choice([0.93, 0.5, 1.01, 0.0], size=40)):
            D[z, y, x] = np.float32(f) * trunc
        D[3, 3, 0] = 0.93 * trunc; D[3, 4, 1] = 1.002 * trunc; D[n - 1, n - 2, n - 1] = 0.2 * trunc
    gv, ov = volumes_with(oracle, (n, n, n), (phys,) * 3, D.reshape(-1))
    if case == "rim_written_by_integrate":
        # a surface that leaves the volume through its faces, fused from depth images: the flags are the marks integrate left, then
        # (second picture) the rebuilt ones
        gv.close()
        gv = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
        ov = oracle.Volume((n, n, n), (3000.0,) * 3)
        for i in range(3):
            d, cam = synth.depth_frame(i * 4, 200, seed=0x5EED0003)
            gv.integrate(d, W, H, cam)
            ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
            compare(oracle, gv, ov, cam, what="%s, frame %d" % (case, i))
        gv.occupancy_data(force_rebuild=True)
        compare(oracle, gv, ov, cam, what="%s, rebuilt flags" % case)
        return
    c = phys / 2
    for pos, look in (((-300.0, c + 7, c - 5), (c, c, c)), ((c + 3, c - 8, -250.0), (c, c, c)), ((c, phys + 280.0, c + 11), (c, c, c)),
                      ((-200.0, -180.0, -150.0), (c, c, c)), ((c, c, c), (0.0, c + 40, c - 30))):
        compare(oracle, gv, ov, camera_at(pos, look_at=look), what="%s from %s" % (case, pos))



# ---- round 4: the per-tile entry bound (EntryParams, tsdf_amd/csrc/common.hpp) --------------------------------------------------

def _cast_both(oracle, gv, ov, cam, width=W, height=H, what=""):
    V, N = gv.raycast(width, height, cam)
    Vo, No = ov.raycast(width, height, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    assert_same_floats(V, Vo, what + " vertices")
    assert_same_floats(N, No, what + " normals")
    return Vo


@pytest.mark.parametrize("n", [64, 72, 100])      # whole super blocks of bricks (wave summary), and grids that are not (workgroup summary)
def test_entry_bound_views_from_outside_inside_and_along_the_faces(oracle, n):
    """Rays start at the nearest flagged unit of 4^3 bricks their 16 x 16 tile can see.  Views for which the bound is made (camera in
    front of everything flagged), views that switch it off (a flagged unit straddles the camera plane), and views that graze the
    grid's faces; the same volume is cast from one pose after the other, so each cast's launch also resets the words of the next."""
    frames = [synth.depth_frame(i, 12, seed=0x5EED0002) for i in range(4)]
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    for d, cam in frames:
        gv.integrate(d, W, H, cam)
        ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
    hits = 0
    for name, cam in (("outside", frames[0][1]),
                      ("outside, rolled", camera_at((1400, 1300, -900), yaw_pitch_roll=(0.1, -0.05, 0.8))),
                      ("far away", camera_at((1500, 1500, -6000))),
                      ("inside, in front of the sphere", camera_at((1500, 1400, 900))),
                      ("inside the sphere's shell", camera_at((1500, 1400, 1460))),
                      ("from behind the wall", camera_at((1500, 1500, 3600), look_at=(1500, 1400, 1800))),
                      ("along the x = 0 face", camera_at((5, 1500, -300))),
                      ("from a corner", camera_at((-800, -700, -900), look_at=(1500, 1400, 1900))),
                      ("outside again", frames[2][1])):
        Vo = _cast_both(oracle, gv, ov, cam, what="%d^3, %s:" % (n, name))
        hits += int((~np.isnan(Vo[:, 0])).sum())
    assert hits > 200000


def test_entry_bound_small_images_and_images_that_are_not_whole_tiles(oracle):
    n = 64
    d, cam0 = synth.depth_frame(0, 5, seed=3)
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    gv.integrate(d, W, H, cam0)
    ov.integrate(d, W, H, cam0.inverse_pose(), cam0.k(), cam0.kinv(), nthreads=oracle.max_threads())
    for (w, h) in ((640, 480), (33, 17), (1, 1), (250, 131), (640, 480), (16, 16)):
        cam = tsdf_amd.Camera(591.1 * w / 640.0, 590.1 * w / 640.0, w / 2.0 + 1.0, h / 2.0 - 0.4)
        cam.move_to(1500, 1300, -600)
        cam.look_at(1500, 1400, 1900)
        _cast_both(oracle, gv, ov, cam, w, h, what="%dx%d:" % (w, h))


def test_entry_bound_cameras_that_do_not_qualify_cast_without_it(oracle):
    """K^-1 whose last row is not (0, 0, 1), a sheared / scaled pose block: the camera depth of a sample is not near + t (or the
    host cannot invert the pose): those views march as before."""
    n = 64
    d, cam0 = synth.depth_frame(0, 5, seed=3)
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    gv.integrate(d, W, H, cam0)
    ov.integrate(d, W, H, cam0.inverse_pose(), cam0.k(), cam0.kinv(), nthreads=oracle.max_threads())
    kinv = cam0.kinv().copy(); kinv[2] = 1.0e-5; kinv[8] = 1.01                  # last row (m31, m32, m33), column-major
    _cast_both(oracle, gv, ov, Cam(cam0.pose(), cam0.inverse_pose(), cam0.k(), kinv), what="projective K^-1:")
    pose = cam0.pose().copy(); pose[0:3] *= 1.3; pose[4] += 0.2                   # scaled first column, a shear: still invertible
    _cast_both(oracle, gv, ov, Cam(pose, cam0.inverse_pose(), cam0.k(), cam0.kinv()), what="sheared pose:")
    pose = cam0.pose().copy(); pose[8:11] = pose[0:3]                             # singular 3 x 3 block
    _cast_both(oracle, gv, ov, Cam(pose, cam0.inverse_pose(), cam0.k(), cam0.kinv()), what="singular pose:")
    _cast_both(oracle, gv, ov, cam0, what="and the plain camera after them:")


def test_entry_bound_a_surface_on_the_entry_face_and_speckles_in_free_space(oracle):
    """Set distances by hand: negative voxels in the first planes the rays meet (the bound of those tiles is the entry itself),
    single low voxels scattered through free space (every one flags its unit and pulls the bound of the tiles that see it forward)
    and an untouched half of the image (no unit at all: the rays of those tiles are done at once)."""
    n, phys = 96, 960.0
    rng = np.random.default_rng(5)
    probe = tsdf_amd.TSDFVolume((n, n, n), (phys,) * 3)
    trunc = np.float32(probe.truncation_distance())
    probe.close()
    D = np.full((n, n, n), trunc, np.float32)
    D[0:2, 10:40, 10:50] = -0.4 * trunc                          # on the z = 0 face
    D[60, 20:70, 5:45] = -0.2 * trunc; D[59, 20:70, 5:45] = 0.3 * trunc
    for _ in range(40):
        z, y, x = rng.integers(3, n - 3, size=3)
        if x < 48:
            D[z, y, x] = rng.choice([-0.5, 0.0, 0.004]) * trunc
    gv = tsdf_amd.TSDFVolume((n, n, n), (phys,) * 3)
    ov = oracle.Volume((n, n, n), (phys,) * 3)
    gv.set_distance_data(D.reshape(-1)); ov.set_distance_data(D.reshape(-1))
    for pos, look in (((480, 480, -700), (480, 480, 480)), ((200, 300, -400), (600, 500, 900)), ((480, 480, 300), (480, 480, 900))):
        cam = camera_at(pos, look_at=look)
        _cast_both(oracle, gv, ov, cam, what="hand-made field from %s:" % (pos,))


_CELLS_PROBE = r"""
import sys, json, numpy as np
import tsdf_amd
from tsdf_amd import synth
from tests.helpers import camera_at
n, views = int(sys.argv[2]), json.loads(sys.argv[3])
gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
for i in range(4):
    d, cam = synth.depth_frame(i, 12, seed=0x5EED0002)
    gv.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
out = {"D": gv.get_distance_data()}
for j, v in enumerate(views):
    cam = camera_at(tuple(v["at"]), look_at=tuple(v["look"]) if v.get("look") else None, yaw_pitch_roll=tuple(v["ypr"]) if v.get("ypr") else None)
    V, N = gv.raycast(synth.WIDTH, synth.HEIGHT, cam)
    out["V%d" % j], out["N%d" % j], out["cells%d" % j] = V, N, np.array(gv.last_raycast_cell_parallel())
np.savez(sys.argv[1], **out)
"""


@pytest.mark.parametrize("n", [64, 100, 160])
def test_cell_parallel_cast_from_outside_beside_and_inside(oracle, tmp_path, n):
    """The cell-parallel cast (raycast_cells.hpp, forced on with TSDF_RAY_CELLS=2 in a process of its own): no ray is marched, every
    flagged brick's mixed cells are offered to the pixels they project to.  Views from outside (taken), from a camera whose plane
    cuts through flagged bricks beside it (taken: the cells that straddle the plane in front of which all samples lie are bounded
    from their part in front), along a face and from a corner; from inside the volume, where the samples start at the camera: a
    box that holds the camera asks every pixel, one that reaches across the camera plane beside it is bounded from its side, and a
    brick seen from close by is worked on in parts.  One volume cast from one pose after the other; every picture must be the oracle's, bit for bit."""
    import json
    import os
    import subprocess
    import sys
    views = [
        {"at": (1500, 1300, -600), "look": (1500, 1400, 1900), "cells": True},
        {"at": (1400, 1300, -900), "ypr": (0.1, -0.05, 0.8), "cells": True},
        {"at": (1500, 1500, -6000), "cells": True},
        {"at": (1500, 1400, -250), "look": (3000, 1400, 600), "cells": True},        # the camera plane cuts the volume beside the camera
        {"at": (-300, 1500, 1500), "look": (1500, 1400, 1900), "cells": True},
        {"at": (-800, -700, -900), "look": (1500, 1400, 1900), "cells": True},       # from a corner
        {"at": (1500, 1500, 3600), "look": (1500, 1400, 1800), "cells": True},       # from behind the wall
        {"at": (1500, 1400, 900), "cells": True},                                    # inside
        {"at": (1500, 1400, 1460), "cells": True},                                   # inside the sphere's shell
        {"at": (5, 1500, -300), "cells": True},                                      # along the x = 0 face, outside by 300 mm in z
        {"at": (1500, 1500, -20), "cells": True},                                    # within four voxels of the entry face
        {"at": (900, 1400, 1200), "look": (2900, 1500, 2800), "cells": True},        # inside, towards a far corner: bricks beside and behind the camera
        {"at": (1500, 1400, 2380), "look": (300, 1400, 2390), "cells": True},        # inside, a hand's width from the wall and along it
        {"at": (1500, 40, 1500), "look": (1500, 3000, 1600), "cells": True},         # inside, a voxel or two from the y = 0 face
        {"at": (1500, 1300, -600), "look": (1500, 1400, 1900), "cells": True},
    ]
    out = str(tmp_path / "cells.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TSDF_RAY_CELLS="2")
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _CELLS_PROBE, out, str(n), json.dumps(views)], check=True, env=e, cwd=root, timeout=900)
    got = np.load(out)
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    ov.set_distance_data(got["D"])
    hits = 0
    for j, v in enumerate(views):
        cam = camera_at(tuple(v["at"]), look_at=v.get("look"), yaw_pitch_roll=v.get("ypr"))
        Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(got["V%d" % j], Vo, "%d^3, view %d %s: vertices" % (n, j, v["at"]))
        assert_same_floats(got["N%d" % j], No, "%d^3, view %d %s: normals" % (n, j, v["at"]))
        assert bool(got["cells%d" % j]) == v["cells"], "%d^3, view %d %s: which kernels ran" % (n, j, v["at"])
        hits += int((~np.isnan(Vo[:, 0])).sum())
    assert hits > 300000


_CELLS_PROBE_K = r"""
import sys, json, numpy as np
import tsdf_amd
from tsdf_amd import synth
n, views = int(sys.argv[2]), json.loads(sys.argv[3])
gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
for i in range(4):
    d, cam = synth.depth_frame(i, 12, seed=0x5EED0002)
    gv.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
out = {"D": gv.get_distance_data()}
for j, v in enumerate(views):
    cam = tsdf_amd.Camera(*v["K"])
    cam.move_to(*v["at"]); cam.look_at(*v["look"])
    V, N = gv.raycast(v["size"][0], v["size"][1], cam)
    out["V%d" % j], out["N%d" % j], out["cells%d" % j] = V, N, np.array(gv.last_raycast_cell_parallel())
np.savez(sys.argv[1], **out)
"""


@pytest.mark.parametrize("cells", ["2", "0"])
def test_cell_parallel_cast_with_unusual_intrinsics(oracle, tmp_path, cells):
    """(cells = 0: the same views through the march kernels.)  The cell-parallel cast bounds a cell's pixels from the view's projection (pixel = K x camera / depth): principal points far off
    the image's centre or off the image, mirrored and anisotropic focal lengths, a fisheye-wide and a tele lens, odd image sizes -- from
    outside and from inside the volume.  Every picture the oracle's, bit for bit; the cast is the cell-parallel one every time."""
    import json
    import os
    import subprocess
    import sys
    n = 100
    views = [
        {"K": (525.0, 525.0, -150.0, 240.0), "size": (640, 480), "at": (1500, 1300, -700), "look": (2300, 1400, 1900)},   # principal point left of the image
        {"K": (525.0, 525.0, 900.0, 600.0), "size": (640, 480), "at": (1500, 1300, -700), "look": (600, 700, 1900)},      # ... beyond its lower right corner
        {"K": (-525.0, 525.0, 320.0, 240.0), "size": (640, 480), "at": (1400, 1300, -900), "look": (1500, 1400, 1900)},    # mirrored
        {"K": (525.0, -400.0, 300.0, 260.0), "size": (640, 480), "at": (1400, 1300, -900), "look": (1500, 1400, 1900)},
        {"K": (300.0, 800.0, 320.0, 100.0), "size": (640, 480), "at": (-500, 1500, 900), "look": (1500, 1400, 1900)},      # anisotropic
        {"K": (120.0, 120.0, 320.0, 240.0), "size": (640, 480), "at": (1500, 1400, -300), "look": (1500, 1400, 1900)},     # very wide: the camera plane's neighbourhood in view
        {"K": (2500.0, 2500.0, 320.0, 240.0), "size": (640, 480), "at": (1500, 1300, -2500), "look": (1500, 1400, 1900)},  # tele: a voxel covers many pixels
        {"K": (97.0, 61.0, 48.0, 30.0), "size": (97, 61), "at": (1500, 1300, -800), "look": (1500, 1400, 1900)},           # a small odd image
        {"K": (525.0, 525.0, -150.0, 240.0), "size": (640, 480), "at": (1500, 1400, 900), "look": (2900, 1500, 2800)},     # off-centre, from inside
        {"K": (120.0, 140.0, 300.0, 250.0), "size": (640, 480), "at": (900, 1400, 1200), "look": (300, 1400, 2390)},       # very wide, from inside
        {"K": (-525.0, -525.0, 320.0, 240.0), "size": (333, 477), "at": (1500, 1400, 2380), "look": (300, 1400, 2390)},    # mirrored twice, inside, along the wall
        # from inside along an axis, the principal point on a pixel: a row / a column of rays has a direction component of exactly 0, for
        # which the reference's ray_box leaves an exit out of its minimum (NaN compares false): those rays are sampled far beyond the
        # grid -- to sample 4402 when it is the z component -- and the clamped interpolation out there finds "surfaces"
        {"K": (525.0, 525.0, 320.0, 240.0), "size": (640, 480), "at": (1500, 1400, 900), "look": (1500, 1400, 2900)},
        {"K": (200.0, 200.0, 320.0, 240.0), "size": (640, 480), "at": (700, 1400, 1500), "look": (2900, 1400, 1500)},
        {"K": (200.0, 200.0, 320.0, 240.0), "size": (640, 480), "at": (1500, 2600, 1500), "look": (1500, 100, 1500.001)},
    ]
    out = str(tmp_path / "cells_k.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TSDF_RAY_CELLS=cells)
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _CELLS_PROBE_K, out, str(n), json.dumps(views)], check=True, env=e, cwd=root, timeout=900)
    got = np.load(out)
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    ov.set_distance_data(got["D"])
    hits = 0
    for j, v in enumerate(views):
        cam = tsdf_amd.Camera(*v["K"])
        cam.move_to(*v["at"])
        cam.look_at(*v["look"])
        Vo, No = ov.raycast(v["size"][0], v["size"][1], cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(got["V%d" % j], Vo, "view %d %s: vertices" % (j, v))
        assert_same_floats(got["N%d" % j], No, "view %d %s: normals" % (j, v))
        assert bool(got["cells%d" % j]) == (cells == "2"), "view %d %s: which kernels ran" % (j, v)
        hits += int((~np.isnan(Vo[:, 0])).sum())
    assert hits > 200000


_CELLS_PROBE_POSE = r"""
import sys, json, numpy as np
import tsdf_amd
from tsdf_amd import synth
n, views = int(sys.argv[2]), json.loads(sys.argv[3])
class M:
    def __init__(self, pose, kinv): self.p, self.ki = np.array(pose, np.float32), np.array(kinv, np.float32)
    def pose(self): return self.p
    def inverse_pose(self): return self.p      # (not used by the cast)
    def k(self): return self.ki
    def kinv(self): return self.ki
gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
for i in range(4):
    d, cam = synth.depth_frame(i, 12, seed=0x5EED0002)
    gv.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
out = {"D": gv.get_distance_data()}
for j, v in enumerate(views):
    V, N = gv.raycast(640, 480, M(v["pose"], v["kinv"]))
    out["V%d" % j], out["N%d" % j], out["cells%d" % j], out["listed%d" % j] = V, N, np.array(gv.last_raycast_cell_parallel()), np.array(gv.last_cell_list())
np.savez(sys.argv[1], **out)
"""


@pytest.mark.parametrize("cells", ["2", "0"])
def test_poses_that_are_not_rigid(oracle, tmp_path, cells):
    """The reference takes whatever 4 x 4 it is given (GPURaycaster.cu:449-455 copies the block; nothing asks for a rotation): blocks that
    shrink (0.5 R), stretch (2 R, 3 along one axis) and shear.  The cell-parallel cast's list drops the bricks no pixel sees by a sphere
    whose radius has to be taken in the CAMERA's frame -- up to the 2-norm of the inverse block larger than in the world (EntryParams::
    r_scale; the advisor's finding of round 5: with 0.5 R visible bricks at the image's edge were dropped).  Both casts, the oracle's bits."""
    import json
    import os
    import subprocess
    import sys
    n = 100
    _, cam = synth.depth_frame(1, 12, seed=0x5EED0002)      # a camera the volume was integrated from: surface out to the image's borders
    P = np.array(cam.pose(), np.float64).reshape(4, 4).T      # rows
    kinv = [float(x) for x in cam.kinv()]
    blocks = [np.eye(3), 0.25 * np.eye(3), 2.0 * np.eye(3), np.diag([1.0, 3.0, 1.0]), np.array([[1.0, 0.4, 0.0], [0.0, 1.0, 0.0], [0.3, 0.0, 1.0]]),
              np.array([[0.4, 0.0, 0.0], [0.2, 0.6, 0.0], [0.0, 0.0, 1.0]])]
    views = []
    for B in blocks:
        Q = P.copy()
        Q[:3, :3] = P[:3, :3] @ B       # the camera's axes scaled / sheared: directions R B d
        views.append({"pose": [float(x) for x in Q.T.reshape(-1)], "kinv": kinv})
    out = str(tmp_path / "cells_pose.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TSDF_RAY_CELLS=cells)
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _CELLS_PROBE_POSE, out, str(n), json.dumps(views)], check=True, env=e, cwd=root, timeout=900)
    got = np.load(out)
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    ov.set_distance_data(got["D"])
    hits = 0
    for j, v in enumerate(views):
        Vo, No = ov.raycast(640, 480, np.array(v["pose"], np.float32), np.array(v["kinv"], np.float32), nthreads=oracle.max_threads())
        assert_same_floats(got["V%d" % j], Vo, "block %d: vertices" % j)
        assert_same_floats(got["N%d" % j], No, "block %d: normals" % j)
        assert bool(got["cells%d" % j]) == (cells == "2")
        hits += int((~np.isnan(Vo[:, 0])).sum())
    assert hits > 100000
    if cells == "2":
        # a pose scaled as a whole shows the same picture of the grid: the bricks in view -- the list -- are the same ones.  (With the
        # radius taken as a world distance the block 0.25 R listed 7 % fewer: visible bricks across the image's border dropped.)
        assert int(got["listed1"]) == int(got["listed0"]) and int(got["listed2"]) == int(got["listed0"]), [int(got["listed%d" % j]) for j in range(3)]


_RECOUNT_PROBE = r"""
import sys, numpy as np
import tsdf_amd
from tsdf_amd import synth
n = 96
gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
for i in range(3):
    d, cam = synth.depth_frame(i, 4, seed=0x5EED0002)
    gv.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
kinds, listed = [], []
for j in range(40):
    V, N = gv.raycast(synth.WIDTH, synth.HEIGHT, cam)
    kinds.append(gv.last_raycast_cell_parallel()); listed.append(gv.last_cell_list())
np.savez(sys.argv[1], V=V, N=N, D=gv.get_distance_data(), kinds=np.array(kinds), listed=np.array(listed))
"""


def test_a_list_over_the_limit_is_counted_again_now_and_then(oracle, tmp_path):
    """The choice of cast goes by the list the last cell-parallel cast built; over the limit the march runs, which builds none.  So that a
    volume does not keep the march for good, every 16th such cast counts the flagged bricks again (choose_cell_cast; the advisor's
    finding of round 5).  With a limit of 10 bricks: the first cast is the cell-parallel one, the rest march, the count in the mirror is
    replaced by the recount at the 17th cast -- and every picture is the oracle's."""
    import os
    import subprocess
    import sys
    out = str(tmp_path / "recount.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TSDF_RAY_CELLS="1", TSDF_RAY_CELLS_LIMIT="10")
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _RECOUNT_PROBE, out], check=True, env=e, cwd=root, timeout=600)
    got = np.load(out)
    kinds, listed = got["kinds"].astype(bool), got["listed"]
    assert kinds[0] and not kinds[1:].any(), kinds
    assert listed[0] > 10 and np.all(listed[:10] == listed[0])
    # (the cast lists tasks -- bricks in view, the large ones in parts --, the recount counts flagged bricks: 3 485 against 660 here)
    assert listed[15] == listed[0] and listed[16] != listed[0] and listed[16] > 10 and np.all(listed[16:] == listed[16]), listed
    ov = oracle.Volume((96, 96, 96), (3000, 3000, 3000))
    ov.set_distance_data(got["D"])
    _, cam = synth.depth_frame(2, 4, seed=0x5EED0002)
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    assert_same_floats(got["V"], Vo, "vertices")
    assert_same_floats(got["N"], No, "normals")


_CHOOSER_PROBE = r"""
import sys, numpy as np
import tsdf_amd
from tsdf_amd import synth
n = 128
gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
cams = []
for i in range(6):
    d, cam = synth.depth_frame(i * 9, 200, seed=0x5EED0003)
    gv.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
    cams.append(cam)
kinds, out = [], {"D": gv.get_distance_data()}
for j in range(60):
    V, N = gv.raycast(synth.WIDTH, synth.HEIGHT, cams[j % 6])
    kinds.append(gv.last_raycast_cell_parallel())
    if j >= 54:
        out["V%d" % (j % 6)], out["N%d" % (j % 6)] = V, N
    if j == 30:     # in the middle of the stream: more surface, flags set by integrate between casts of different kinds
        d, cam = synth.depth_frame(70, 200, seed=0x5EED0003)
        gv.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
        out["D"] = gv.get_distance_data()
out["kinds"] = np.array(kinds)
np.savez(sys.argv[1], **out)
"""


@pytest.mark.parametrize("chooser", ["2", "1"])
def test_the_cast_chooser_changes_no_bit(oracle, tmp_path, chooser):
    """TSDF_RAY_CELLS=1 (the default) picks the cast from measured times and tries the other one now and then (raycast.hip: choose_cast).
    A stream of 60 casts over six views with an integration in the middle; TSDF_RAY_CHOOSER=2 tries the other cast every few casts
    whatever the times say, so that both kinds alternate on one volume's buffers: the last six pictures are the oracle's, bit for bit, and
    both kinds did run.  (1: the default schedule -- whichever casts it takes, the same bits.)"""
    import os
    import subprocess
    import sys
    out = str(tmp_path / "chooser.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TSDF_RAY_CELLS="1", TSDF_RAY_CHOOSER=chooser)
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _CHOOSER_PROBE, out], check=True, env=e, cwd=root, timeout=600)
    got = np.load(out)
    kinds = got["kinds"].astype(bool)
    if chooser == "2":
        assert 5 <= int(kinds.sum()) <= 55, kinds        # both casts, many times each
    ov = oracle.Volume((128, 128, 128), (3000, 3000, 3000))
    ov.set_distance_data(got["D"])
    for q in range(6):
        _, cam = synth.depth_frame(q * 9, 200, seed=0x5EED0003)
        Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(got["V%d" % q], Vo, "view %d: vertices" % q)
        assert_same_floats(got["N%d" % q], No, "view %d: normals" % q)


def test_tiny_images_with_the_list_sorted(oracle, tmp_path):
    """The sorted list's depth-bin counters (516 words with the queue's) are zeroed by the resolve kernel of the cast before: by EVERY word of
    them, also when an image of 7 x 8 pixels is one workgroup of 256 threads.  (It was not: the bins' upper half and the places taken stayed,
    and the next sorted list lost entries -- 4 of 1 250 seeds of the extended fuzz with TSDF_RAY_CELLS_SORT=2, round 6.)  Small images of the
    same volume, cast again and again, every list sorted, from outside and from inside."""
    import json
    import os
    import subprocess
    import sys
    n = 100
    views = []
    # (each image size three times in a row: a cast of another size resets everything with a memset of its own)
    for v in ({"K": (9.0, 9.0, 3.5, 4.0), "size": (7, 8), "at": (1500, 1300, -800), "look": (1500, 1400, 1900)},
              {"K": (14.0, 11.0, 6.0, 4.0), "size": (13, 9), "at": (1500, 1400, 900), "look": (2900, 1500, 2800)},
              {"K": (30.0, 30.0, 10.0, 8.0), "size": (21, 17), "at": (2600, 1300, -300), "look": (1500, 1400, 1900)}):
        views += [v, v, v]
    out = str(tmp_path / "tiny.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TSDF_RAY_CELLS="2", TSDF_RAY_CELLS_SORT="2")
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    subprocess.run([sys.executable, "-c", _CELLS_PROBE_K, out, str(n), json.dumps(views)], check=True, env=e, cwd=root, timeout=900)
    got = np.load(out)
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    ov.set_distance_data(got["D"])
    hits = 0
    for j, v in enumerate(views):
        cam = tsdf_amd.Camera(*v["K"])
        cam.move_to(*v["at"])
        cam.look_at(*v["look"])
        Vo, No = ov.raycast(v["size"][0], v["size"][1], cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(got["V%d" % j], Vo, "cast %d %s: vertices" % (j, v))
        assert_same_floats(got["N%d" % j], No, "cast %d %s: normals" % (j, v))
        assert bool(got["cells%d" % j])
        hits += int((~np.isnan(Vo[:, 0])).sum())
    assert hits > 300


def test_an_uploaded_field_that_flags_every_brick_keeps_the_march(oracle):
    """The cell-parallel cast's work is the number of flagged bricks times their cells' pixels; the choice goes by the list the previous
    cast built.  After a bulk change of the distances that count says nothing: the flags are rebuilt and counted before the first cast
    (count_after_bulk_change).  A field with sign changes everywhere lists every brick -- 262 144 at 256^3, over the limit of 131 072:
    the march runs (and gives the oracle's picture); after clear() and a few integrated frames the cell-parallel cast is back."""
    n = 256
    rng = np.random.default_rng(11)
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    trunc = gv.truncation_distance()
    D = (rng.uniform(-1.0, 1.0, size=n * n * n) * trunc).astype(np.float32)
    gv.set_distance_data(D)
    cam = camera_at((1500, 1400, -900), look_at=(1500, 1400, 1900))
    V, N = gv.raycast(W, H, cam)
    if os.environ.get("TSDF_RAY_CELLS", "1") == "1":
        assert not gv.last_raycast_cell_parallel()
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    ov.set_distance_data(D)
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    assert_same_floats(V, Vo, "every brick flagged: vertices")
    assert_same_floats(N, No, "every brick flagged: normals")
    gv.clear()
    for i in range(3):
        d, c = synth.depth_frame(i, 12, seed=0x5EED0002)
        gv.integrate(d, synth.WIDTH, synth.HEIGHT, c)
    gv.raycast(W, H, cam)
    gv.raycast(W, H, cam)
    if os.environ.get("TSDF_RAY_CELLS", "1") == "1":
        assert gv.last_raycast_cell_parallel()
    gv.close()


def test_the_ray_cast_and_fuzz_suites_again_with_the_cell_parallel_cast_forced():
    """The default choice of cast (choose_cell_cast) keeps the march for views from inside the volume, coarse grids and close-ups: run
    that way, most tests of this file and of test_fuzz_parity.py only ever exercise the march.  Here the same tests run once more in a
    process with TSDF_RAY_CELLS=2 -- the cell-parallel cast wherever the view has a projection (the knob is read once per process) --
    so that the recorded GPU run covers both casts on every scene.  (Left out: the tests that start processes with a setting of their
    own, the bilateral / ICP fuzz that casts nothing, and this test.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, TSDF_RAY_CELLS="2")
    e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
    skip = "not again_with_the_cell and not cast_chooser and not tiny_images and not schedule_knobs and not unusual_intrinsics and not not_rigid and not counted_again and not outside_beside_and_inside and not bilateral and not icp"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_parity_raycast.py", "tests/test_fuzz_parity.py", "-m", "gpu", "-x", "-q", "-k", skip, "-p", "no:cacheprovider"],
                       env=e, cwd=root, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail + "\n" + r.stderr[-2000:]
    import re
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 100, tail
