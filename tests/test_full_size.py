"""Parity at BASELINE's full single-GPU size (512^3, 640x480), where a complete oracle run is still affordable on
the GPU box's host cores for integrate and for the ray cast of every pixel, plus size-independent
properties of the ray caster: splitting the volume into Z-slabs, or the march into sample ranges, or switching the
empty-space skipping off, must not change a single bit."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, assert_same_floats
from tsdf_amd import synth

pytestmark = pytest.mark.gpu
N = 512


@pytest.fixture(scope="module")
def scene(oracle):
    frames = []
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    for i in (0, 7, 15):
        d, cam = synth.depth_frame(i, 200, seed=0x5EED0003)
        f = d.copy()
        bil.filter(f, W, H)
        frames.append((f, cam))
    gv = tsdf_amd.TSDFVolume((N, N, N), (3000.0,) * 3)
    ov = oracle.Volume((N, N, N), (3000.0,) * 3)
    gv.set_counting(True)
    counts = []
    for f, cam in frames:
        gv.integrate(f, W, H, cam)
        u = ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
        counts.append((gv.last_updated_voxels(), u))
    return gv, ov, frames, counts


def test_integrate_512_is_bit_identical_to_the_oracle(scene):
    gv, ov, frames, counts = scene
    for g, o in counts:
        assert g == o and o > 10_000_000
    assert_same_floats(gv.get_weight_data(), ov.weight, "512^3 weights")
    assert_same_floats(gv.get_distance_data(), ov.dist, "512^3 distances")


def test_raycast_512_every_ray_is_bit_identical_to_the_oracle(scene, oracle):
    """All 480 rows (307 200 rays, 1.26e9 reference samples: seconds on the box's host threads), vertices and normals."""
    gv, ov, frames, _ = scene
    cam = frames[-1][1]
    V, Nn = gv.raycast(W, H, cam)
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    assert_same_floats(V, Vo, "512^3 ray cast, every ray: vertices")
    assert_same_floats(Nn, No, "512^3 ray cast, every ray: normals")
    assert (~np.isnan(V[:, 0])).mean() > 0.5


def test_raycast_512_properties(scene):
    import torch
    gv, ov, frames, _ = scene
    cam = frames[0][1]
    V, Nn = gv.raycast(W, H, cam)
    rc = tsdf_amd.GPURaycaster(W, H)
    # (1) the instrumented kernel marches every sample of the reference (no skipping): same number of hits
    st = rc.stats(gv, cam)
    assert st["hits"] == int((~np.isnan(V[:, 0])).sum())
    assert st["evaluated"] < st["samples"] // 20          # and skipping really skips
    # (2) three Z-slabs + min-k merge == the whole volume
    dist = gv.get_distance_data().reshape(N, -1)
    bounds = ((0, 170), (170, 341), (341, N))
    hits = torch.empty((len(bounds), W * H, 2), dtype=torch.float32, device="cuda")
    for i, (zb, ze) in enumerate(bounds):
        s = tsdf_amd.TSDFVolume((N, N, N), (3000.0,) * 3, slab=(zb, ze))
        lo, hi = s.resident_planes()
        s.set_distance_data(dist[lo:hi])
        rc.raycast_slab_device(s, cam, hits[i].data_ptr())
        s.synchronize()
        if i + 1 < len(bounds):
            s.close()
    Vm = torch.empty((W * H, 3), dtype=torch.float32, device="cuda")
    tsdf_amd.merge_hits_device(s, hits.data_ptr(), len(bounds), W, H, cam, Vm.data_ptr())
    torch.cuda.synchronize()
    assert_same_floats(Vm.cpu().numpy(), V, "slab merge at 512^3")
    # (3) normals: unit length wherever defined
    ok = np.isfinite(Nn).all(axis=1) & (np.abs(Nn).sum(axis=1) > 0)
    assert np.allclose(np.linalg.norm(Nn[ok], axis=1), 1.0, atol=1e-5)


def test_config2_256_cubed_all_50_frames(oracle):
    """BASELINE configs[1] (256^3, TUM-style stream of 50 frames, ground-truth poses, integrate + raycast) on the synthetic
    surrogate, every frame: every voxel and every pixel against the oracle."""
    n = 256
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    cams = []
    for i in range(50):
        d, cam = synth.depth_frame(i, 50, seed=0x5EED0002)
        gv.integrate(d, W, H, cam)
        ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
        cams.append(cam)
    assert_same_floats(gv.get_weight_data(), ov.weight, "config 2 weights")
    assert_same_floats(gv.get_distance_data(), ov.dist, "config 2 distances")
    assert ov.weight.max() == 50.0                      # weights are not capped (Q4: max_weight = 15 is never applied)
    V, Nn = gv.raycast(W, H, cams[0])
    Vo, No = ov.raycast(W, H, cams[0].pose(), cams[0].kinv(), nthreads=oracle.max_threads())
    assert_same_floats(V, Vo, "config 2 vertices")
    assert_same_floats(Nn, No, "config 2 normals")
    assert (~np.isnan(V[:, 0])).mean() > 0.8


def _assert_same_floats_chunked(a, b, what, chunk=1 << 26):
    """assert_same_floats for arrays of gigabytes: no whole-array temporaries."""
    a = a.reshape(-1)
    b = b.reshape(-1)
    assert a.shape == b.shape, what
    for o in range(0, a.size, chunk):
        assert_same_floats(a[o:o + chunk], b[o:o + chunk], "%s [%d:]" % (what, o))


def test_config4_1024_cubed_whole_and_in_eight_slabs_against_the_oracle(oracle):
    """BASELINE configs[3] at its full size against the ORACLE (round 3; up to round 2 the eight slabs were compared with the
    product's own whole volume): a 1024^3 volume, two frames of the camera inside it.  (1) whole volume on one GPU: every
    distance, every weight, every ray and normal equal to oracle.Volume on all host threads, bit for bit; (2) 8 Z-slabs of
    128 planes (+ one halo plane each), slab by slab on one GPU: every resident plane equal to the oracle's planes, the
    min-k merge of the slabs' hit records and its normals equal to the ORACLE's picture.  Reference arithmetic:
    src/TSDF/TSDFVolume.cu:308-392, src/RayCaster/GPURaycaster.cu:265-377, 393-427.  (At 1024^3 a ray covers 4402 * 0.279 mm =
    1228 mm, Q8: the camera sits inside the volume, 1 m in front of the wall.)  Host memory: 8 GiB for the oracle's two
    arrays + 4 GiB for one downloaded array at a time."""
    import torch
    from tests.helpers import camera_at
    from tsdf_amd import multi
    n, P = 1024, 8
    cams = [camera_at((1500.0 + 40.0 * i, 1350.0 - 25.0 * i, 1400.0 - 30.0 * i), look_at=synth.LOOK_AT) for i in range(2)]
    frames = []
    for cam in cams:
        z = synth.trace_depth(cam, W, H)
        frames.append(np.clip(np.where(np.isfinite(z), np.rint(z), 0.0), 0, 65535).astype(np.uint16).reshape(-1))
    threads = oracle.max_threads()
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    whole = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    whole.set_counting(True)
    updated = 0
    for f, cam in zip(frames, cams):
        whole.integrate(f, W, H, cam)
        u = whole.last_updated_voxels()
        uo = ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
        assert u == uo, "voxels updated: GPU %d, oracle %d" % (u, uo)
        updated += u
    assert updated > 50_000_000
    d = whole.get_weight_data()
    _assert_same_floats_chunked(d, ov.weight, "1024^3 weights vs oracle")
    d = whole.get_distance_data()
    _assert_same_floats_chunked(d, ov.dist, "1024^3 distances vs oracle")
    del d
    V, Nn = whole.raycast(W, H, cams[0])
    whole.close()
    Vo, No = ov.raycast(W, H, cams[0].pose(), cams[0].kinv(), nthreads=threads)
    assert (~np.isnan(Vo[:, 0])).mean() > 0.5
    assert_same_floats(V, Vo, "1024^3 ray cast, every ray, vs oracle: vertices")
    assert_same_floats(Nn, No, "1024^3 ray cast, every ray, vs oracle: normals")

    Do, Wo = ov.dist.reshape(n, -1), ov.weight.reshape(n, -1)
    rc = tsdf_amd.GPURaycaster(W, H)
    hits = multi.new_hit_records(P, W, H)
    for r in range(P):
        zb, ze = multi.slab_range(n, P, r)
        assert ze - zb == n // P
        s = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3, slab=(zb, ze))
        for f, cam in zip(frames, cams):
            s.integrate(f, W, H, cam)
        lo, hi = s.resident_planes()
        assert (lo, hi) == (zb, min(ze + 1, n))
        assert_same_floats(s.get_distance_data().reshape(hi - lo, -1), Do[lo:hi], "slab %d distances vs oracle" % r)
        assert_same_floats(s.get_weight_data().reshape(hi - lo, -1), Wo[lo:hi], "slab %d weights vs oracle" % r)
        rc.raycast_slab_device(s, cams[0], hits[r].data_ptr())
        s.synchronize()
        if r + 1 < P:
            s.close()
    Vm = torch.empty((W * H, 3), dtype=torch.float32, device="cuda")
    Nm = torch.empty((W * H, 3), dtype=torch.float32, device="cuda")
    tsdf_amd.merge_hits_device(s, hits.data_ptr(), P, W, H, cams[0], Vm.data_ptr())
    tsdf_amd.compute_normals_device(W, H, Vm.data_ptr(), Nm.data_ptr())
    torch.cuda.synchronize()
    assert_same_floats(Vm.cpu().numpy(), Vo, "8-slab merge at 1024^3 vs oracle: vertices")
    assert_same_floats(Nm.cpu().numpy(), No, "8-slab merge at 1024^3 vs oracle: normals")
