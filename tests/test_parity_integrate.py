"""GPU parity: TSDFVolume.integrate (HIP, through the C ABI) against the CPU oracle, bit for bit.

Run on the GPU box with `pytest -m gpu`.  The tolerance north_star allows is 1e-4 relative; because the HIP
kernels keep the reference's operation order with fp contraction off, the assertion here is stronger: every
distance and weight must be bit-identical to the oracle's.
"""
import os

import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, assert_same_floats, camera_at, sphere_depth_map
from tsdf_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_both(oracle, size, physical, frames, width=W, height=H, offset=None, offset_after_clear=None):
    """Integrate `frames` = [(depth, camera)] on the GPU and in the oracle; return both volumes' arrays."""
    gv = tsdf_amd.TSDFVolume(size, physical)
    ov = oracle.Volume(size, physical)
    if offset is not None:                    # offset known before clear(): baked into the grid (Q1)
        gv.offset(*offset); gv.clear()
        ov.offset(*offset); ov.clear()
    if offset_after_clear is not None:        # offset changed after clear(): added on top (Q1)
        gv.offset(*offset_after_clear)
        ov.offset(*offset_after_clear)
    gv.set_counting(True)
    updates = []
    for depth, cam in frames:
        gv.integrate(depth, width, height, cam)
        u = ov.integrate(depth, width, height, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
        updates.append((gv.last_updated_voxels(), u))
    return gv, ov, updates


def check(gv, ov, updates, what):
    assert_same_floats(gv.get_weight_data(), ov.weight, what + " weights")
    assert_same_floats(gv.get_distance_data(), ov.dist, what + " distances")
    for g, o in updates:
        assert g == o, "%s: GPU counted %d updated voxels, oracle %d" % (what, g, o)


def test_wall_frame_of_the_survey_probe(oracle):
    cam = camera_at((1500, 1500, -1000))
    gv, ov, up = run_both(oracle, (128, 128, 128), (3000, 3000, 3000), [(synth.wall_depth(2500), cam)])
    check(gv, ov, up, "wall 128^3")
    assert up[0][0] == 354056          # BASELINE.md section 2 (reference source run by the survey)


def test_config1_frame_with_dropouts(oracle):
    # BASELINE config 1: 128^3, single synthetic frame, ground-truth pose
    cam = camera_at((1500, 1500, -1000))
    gv, ov, up = run_both(oracle, (128, 128, 128), (3000, 3000, 3000), [(synth.config1_depth(), cam)])
    check(gv, ov, up, "config1")


def test_sphere_depth_map_like_the_reference_fixture(oracle):
    depth = sphere_depth_map(W, H, 200, 1200, 2200)     # TestHelpers.cpp:144-183
    cam = camera_at((1500, 1500, -500))
    gv, ov, up = run_both(oracle, (64, 64, 64), (3000, 3000, 3000), [(depth, cam)])
    check(gv, ov, up, "sphere depth")


def test_rotated_poses_accumulate_over_several_frames(oracle):
    frames = []
    for i in range(5):
        d, cam = synth.depth_frame(i, 5, seed=0x5EED0002)
        frames.append((d, cam))
    gv, ov, up = run_both(oracle, (96, 96, 96), (3000, 3000, 3000), frames)
    check(gv, ov, up, "5 frames 96^3")
    assert ov.weight.max() >= 4


def test_general_rotation_and_camera_inside_the_volume(oracle):
    # camera inside the grid: voxels behind the camera project too (Q2) and bricks straddle the camera plane
    d, _ = synth.depth_frame(0, 7, seed=11)
    cam = camera_at((1400, 1350, 600), yaw_pitch_roll=(0.35, -0.2, 0.15))
    gv, ov, up = run_both(oracle, (80, 72, 64), (3000, 2700, 2400), [(d, cam)])
    check(gv, ov, up, "camera inside")
    assert up[0][1] > 0


def test_odd_sizes_that_do_not_fill_the_tiles(oracle):
    d, cam = synth.depth_frame(2, 9, seed=3)
    gv, ov, up = run_both(oracle, (50, 37, 29), (2500, 1850, 1450), [(d, cam)])
    check(gv, ov, up, "50x37x29")


def test_small_image_and_tiny_volume(oracle):
    cam = camera_at((150, 150, -300))
    depth = np.full(32 * 24, 450, np.uint16)
    k = cam.k().copy(); k[0] /= 20; k[4] /= 20; k[6] = 16; k[7] = 12
    from tests.helpers import Cam
    small = Cam(cam.pose(), cam.inverse_pose(), k, oracle.mat3_inverse(k))
    gv, ov, up = run_both(oracle, (8, 8, 8), (300, 300, 300), [(depth, small)], width=32, height=24)
    check(gv, ov, up, "8^3 / 32x24")
    assert up[0][1] > 0


def test_offset_baked_at_clear_and_offset_changed_afterwards(oracle):
    d, cam = synth.depth_frame(1, 6, seed=5)
    gv, ov, up = run_both(oracle, (64, 64, 64), (3000, 3000, 3000), [(d, cam)], offset=(100.0, -50.0, 25.0))
    check(gv, ov, up, "offset at clear")
    # Q1: setting the offset after construction adds it ON TOP of the grid built at clear() time
    gv, ov, up = run_both(oracle, (64, 64, 64), (3000, 3000, 3000), [(d, cam)], offset_after_clear=(100.0, -50.0, 25.0))
    check(gv, ov, up, "offset after clear")


def test_all_invalid_depth_changes_nothing(oracle):
    cam = camera_at((1500, 1500, -1000))
    gv, ov, up = run_both(oracle, (32, 32, 32), (3000, 3000, 3000), [(np.zeros(W * H, np.uint16), cam)])
    check(gv, ov, up, "all-zero depth")
    assert up[0] == (0, 0)
    assert np.all(gv.get_weight_data() == 0)
    assert np.all(gv.get_distance_data() == np.float32(gv.truncation_distance()))


def test_camera_looking_away_updates_nothing_in_front_but_matches_oracle(oracle):
    cam = camera_at((1500, 1500, 4000), look_at=(1500, 1500, 9000))     # volume entirely behind the camera
    gv, ov, up = run_both(oracle, (48, 48, 48), (3000, 3000, 3000), [(synth.wall_depth(1200), cam)])
    check(gv, ov, up, "volume behind camera")


def test_materialised_deformation_grid_gives_the_same_result_as_the_implicit_one(oracle):
    d, cam = synth.depth_frame(0, 4, seed=9)
    gv, ov, up = run_both(oracle, (40, 40, 40), (3000, 3000, 3000), [(d, cam)])
    gv2 = tsdf_amd.TSDFVolume((40, 40, 40), (3000, 3000, 3000))
    assert gv2.deformation() != 0                      # materialises the 24-byte nodes; DEFORM kernel path
    assert gv2.info().deformation_materialised == 1
    gv2.integrate(d, W, H, cam)
    assert_same_floats(gv2.get_distance_data(), ov.dist, "materialised nodes")
    assert_same_floats(gv2.get_weight_data(), ov.weight, "materialised nodes")


def test_custom_deformation_nodes_move_the_voxel_centres(oracle):
    n = 24
    d, cam = synth.depth_frame(0, 4, seed=21)
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    vs = ov.voxel_size()
    rng = np.random.RandomState(3)
    zz, yy, xx = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    tr = np.stack([(xx + 0.5) * vs[0], (yy + 0.5) * vs[1], (zz + 0.5) * vs[2]], -1).astype(np.float32)
    tr += rng.uniform(-40, 40, tr.shape).astype(np.float32)
    ov.translation = np.ascontiguousarray(tr.reshape(-1))
    ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv())
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000))
    nodes = np.concatenate([tr.reshape(-1, 3), np.zeros((n ** 3, 3), np.float32)], axis=1)
    gv.set_deformation(nodes)
    gv.integrate(d, W, H, cam)
    assert_same_floats(gv.get_distance_data(), ov.dist, "custom nodes")
    assert_same_floats(gv.get_weight_data(), ov.weight, "custom nodes")


def test_clear_resets_and_set_get_round_trip(oracle):
    gv = tsdf_amd.TSDFVolume((20, 21, 22), (2000, 2100, 2200))
    ov = oracle.Volume((20, 21, 22), (2000, 2100, 2200))
    assert gv.size() == (20, 21, 22)
    assert np.array_equal(gv.voxel_size(), ov.voxel_size())
    assert gv.truncation_distance() == ov.truncation_distance()
    rng = np.random.RandomState(1)
    a = rng.rand(20 * 21 * 22).astype(np.float32)
    gv.set_distance_data(a)
    gv.set_weight_data(a * 2)
    assert np.array_equal(gv.get_distance_data(), a) and np.array_equal(gv.get_weight_data(), a * 2)
    gv.clear()
    assert np.all(gv.get_weight_data() == 0)
    assert np.all(gv.get_distance_data() == np.float32(gv.truncation_distance()))


def test_slab_volumes_integrate_their_planes_plus_one_halo_plane(oracle):
    # multi-GPU sharding unit: two Z-slabs on one device equal the whole volume, plane for plane
    n = 48
    d, cam = synth.depth_frame(3, 8, seed=2)
    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv())
    full = ov.dist.reshape(n, n * n)
    for zb, ze in ((0, 24), (24, 48), (10, 11)):
        s = tsdf_amd.TSDFVolume((n, n, n), (3000, 3000, 3000), slab=(zb, ze))
        lo, hi = s.resident_planes()
        assert (lo, hi) == (zb, min(ze + 1, n))
        s.integrate(d, W, H, cam)
        assert_same_floats(s.get_distance_data(), full[lo:hi], "slab [%d,%d)" % (zb, ze))


@pytest.mark.parametrize("planes", [32, 33, 36, 37, 64, 65, 68, 70])
def test_depths_around_the_brick_layers(oracle, planes):
    """The integrate kernel walks 32-plane bricks; up to 4 left-over planes (a slab's halo plane, typically) ride on the
    last full layer, more get a layer of their own.  Whole volumes and slabs of every such depth, two frames each."""
    frames = [synth.depth_frame(i, 8, seed=5) for i in (1, 4)]
    size, phys = (72, 20, planes), (2700.0, 750.0, planes * 37.5)
    gv, ov, up = run_both(oracle, size, phys, frames)
    check(gv, ov, up, "%d planes" % planes)
    if planes >= 36:    # the same planes as a slab [2, planes - 1) plus its halo plane
        s = tsdf_amd.TSDFVolume(size, phys, slab=(2, planes - 1))
        for d, cam in frames:
            s.integrate(d, W, H, cam)
        lo, hi = s.resident_planes()
        assert_same_floats(s.get_distance_data(), ov.dist.reshape(planes, -1)[lo:hi], "slab of %d resident planes" % (hi - lo))
        assert_same_floats(s.get_weight_data(), ov.weight.reshape(planes, -1)[lo:hi], "slab weights")


def test_committed_golden_vectors(oracle):
    """tests/golden/oracle_integrate_raycast.npz (generated by tests/golden/make_golden.py)."""
    from tests.helpers import Cam
    f = np.load(os.path.join(GOLD, "oracle_integrate_raycast.npz"))
    k2, kinv2 = oracle.camera_k(591.1 / 4, 590.1 / 4, 331.0 / 4, 234.6 / 4)
    pose = oracle.identity_pose((1500, 1500, -1000))
    cam = Cam(pose, oracle.mat4_inverse(pose), k2, kinv2)
    gv = tsdf_amd.TSDFVolume((32, 32, 32), (3000, 3000, 3000))
    gv.integrate(np.full(160 * 120, 2500, np.uint16), 160, 120, cam)
    assert_same_floats(gv.get_distance_data(), f["wall32_dist"], "golden wall32 dist")
    assert_same_floats(gv.get_weight_data(), f["wall32_weight"], "golden wall32 weight")
    V, N = gv.raycast(160, 120, cam)
    assert_same_floats(V, f["wall32_vertices"], "golden wall32 vertices")
    assert_same_floats(N, f["wall32_normals"], "golden wall32 normals")


def test_non_standard_intrinsics_and_projective_pose_take_the_general_path(oracle):
    # skewed K (k12 != 0) and an inverse pose whose last row is not (0,0,0,1): the standard-camera shortcuts of
    # the kernel (image.z == cam.z, w == 1, surface z == depth) must not be taken
    from tests.helpers import Cam
    d, cam = synth.depth_frame(1, 5, seed=17)
    k = cam.k().copy()
    k[3] = 3.5                      # K(0,1): skew
    kinv = oracle.mat3_inverse(k)
    skew = Cam(cam.pose(), cam.inverse_pose(), k, kinv)
    gv, ov, up = run_both(oracle, (64, 64, 64), (3000, 3000, 3000), [(d, skew)])
    check(gv, ov, up, "skewed intrinsics")
    assert up[0][1] > 0
    ip = cam.inverse_pose().copy()
    ip[3] = 1.0e-5                  # inv_pose(3,0): w = 1 + 1e-5 * x
    ip[15] = 0.98
    proj = Cam(cam.pose(), ip, cam.k(), cam.kinv())
    gv, ov, up = run_both(oracle, (64, 64, 64), (3000, 3000, 3000), [(d, proj)])
    check(gv, ov, up, "projective inverse pose")
    assert up[0][1] > 0


def _cropped(depth, width, height, new_w, new_h):
    """Top-left new_w x new_h window of a width x height depth image (same intrinsics: the principal point stays where it was)."""
    return np.ascontiguousarray(depth.reshape(height, width)[:new_h, :new_w]).reshape(-1)


@pytest.mark.parametrize("size", [(639, 480), (321, 241), (638, 479)])
def test_odd_image_widths_take_the_single_pixel_staging(oracle, size):
    """The brick's pixel box is staged two pixels per lane only for an even image width (integrate.hip: pair_loads); an odd
    width, and an even one with odd boxes at the right edge, must give the same bits."""
    w, h = size
    frames = []
    for i in (1, 4):
        d, cam = synth.depth_frame(i, 8, seed=9)
        frames.append((_cropped(d, W, H, w, h), cam))
    gv, ov, up = run_both(oracle, (80, 72, 64), (3000, 2700, 2400), frames, width=w, height=h)
    check(gv, ov, up, "%dx%d image" % (w, h))
    assert up[0][1] > 0


def test_depth_pointer_that_is_not_4_byte_aligned(oracle):
    """integrate_device accepts any device pointer to uint16: one that sits 2 bytes into a buffer takes the single-pixel staging."""
    import torch
    d, cam = synth.depth_frame(3, 8, seed=11)
    buf = torch.zeros(W * H + 1, dtype=torch.int16, device="cuda")
    buf[1:] = torch.from_numpy(d.view(np.int16)).cuda()
    assert (buf.data_ptr() + 2) % 4 == 2
    gv = tsdf_amd.TSDFVolume((96, 96, 96), (3000.0,) * 3)
    ov = oracle.Volume((96, 96, 96), (3000.0,) * 3)
    gv.integrate_device(buf.data_ptr() + 2, W, H, cam)
    torch.cuda.synchronize()
    ov.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
    assert_same_floats(gv.get_weight_data(), ov.weight, "misaligned depth weights")
    assert_same_floats(gv.get_distance_data(), ov.dist, "misaligned depth distances")


def test_image_with_more_tiles_than_the_cull_kernel_keeps_in_lds(oracle):
    """1600 x 1200 pixels are 7 500 depth tiles; brick_cull_kernel stages at most 4 096 maxima in LDS and reads the rest from memory."""
    w, h = 1600, 1200
    cam = camera_at((1500, 1500, -800))
    k = cam.k().copy(); k[0] *= 2.5; k[4] *= 2.5; k[6] = w / 2.0; k[7] = h / 2.0
    from tests.helpers import Cam
    big = Cam(cam.pose(), cam.inverse_pose(), k, oracle.mat3_inverse(k))
    rng = np.random.default_rng(5)
    depth = (2200 + 60 * np.sin(np.arange(w * h) * 1e-3) + rng.integers(-3, 4, size=w * h)).astype(np.uint16)
    depth[rng.random(w * h) < 0.02] = 0
    gv, ov, up = run_both(oracle, (64, 64, 64), (3000, 3000, 3000), [(depth, big)], width=w, height=h)
    check(gv, ov, up, "1600x1200 image")
    assert up[0][1] > 0


def test_tile_maxima_brought_by_the_caller_give_the_same_volume(oracle):
    """tsdf_integrate_device_tiles skips depth_tile_max_kernel and culls on the maxima the bilateral filter left; calls with and
    without them alternate on one volume (the brick list's two length words change sides every integration)."""
    import torch
    s = torch.cuda.current_stream().cuda_stream
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    gv = tsdf_amd.TSDFVolume((96, 96, 96), (3000.0,) * 3)
    gv.set_stream(s)
    ov = oracle.Volume((96, 96, 96), (3000.0,) * 3)
    tmax = torch.empty((30 * 40,), dtype=torch.int16, device="cuda")
    for n, with_tiles in enumerate((True, True, False, True, False, False, True)):
        d, cam = synth.depth_frame(n, 12, seed=21)
        src = torch.from_numpy(d.view(np.int16).copy()).cuda()
        dst = torch.empty_like(src)
        bil.filter_device(src.data_ptr(), dst.data_ptr(), W, H, bits=16, stream=s, tile_max_ptr=tmax.data_ptr() if with_tiles else None)
        gv.integrate_device(dst.data_ptr(), W, H, cam, tile_max_ptr=tmax.data_ptr() if with_tiles else None)
        torch.cuda.synchronize()
        filt = dst.cpu().numpy().view(np.uint16)
        ov.integrate(filt, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
        assert_same_floats(gv.get_weight_data(), ov.weight, "frame %d weights" % n)
        assert_same_floats(gv.get_distance_data(), ov.dist, "frame %d distances" % n)


def test_slab_with_caller_tile_maxima(oracle):
    import torch
    s = torch.cuda.current_stream().cuda_stream
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    whole = tsdf_amd.TSDFVolume((64, 64, 96), (2000.0, 2000.0, 3000.0))
    slab = tsdf_amd.TSDFVolume((64, 64, 96), (2000.0, 2000.0, 3000.0), slab=(30, 71))
    tmax = torch.empty((30 * 40,), dtype=torch.int16, device="cuda")
    for n in range(3):
        d, cam = synth.depth_frame(n, 12, seed=22)
        src = torch.from_numpy(d.view(np.int16).copy()).cuda()
        dst = torch.empty_like(src)
        bil.filter_device(src.data_ptr(), dst.data_ptr(), W, H, bits=16, stream=s, tile_max_ptr=tmax.data_ptr())
        whole.integrate_device(dst.data_ptr(), W, H, cam)
        slab.integrate_device(dst.data_ptr(), W, H, cam, tile_max_ptr=tmax.data_ptr())
        torch.cuda.synchronize()
    a, b = whole.get_distance_data().reshape(96, 64, 64), slab.get_distance_data().reshape(-1, 64, 64)
    z0 = 30
    assert_same_floats(a[z0:z0 + b.shape[0]].reshape(-1), b.reshape(-1), "slab distances")


# ---- round 4: bricks that straddle the camera plane are culled against the double cone of the image's side planes ----------------

@pytest.mark.parametrize("ypr", [(0.0, 0.0, 0.0), (0.35, -0.2, 0.15), (1.2, 0.4, -0.7), (3.0, 0.1, 0.0), (-1.57, 1.3, 2.0)])
def test_camera_inside_a_grid_of_many_bricks(oracle, ypr):
    # 192 x 96 x 128 voxels = 3 x 24 x 4 integrate bricks; the camera plane cuts through most of them at every angle
    d, _ = synth.depth_frame(1, 7, seed=21)
    cam = camera_at((1310.0, 1490.0, 1720.0), yaw_pitch_roll=ypr)
    gv, ov, up = run_both(oracle, (192, 96, 128), (3000, 3000, 3000), [(d, cam), (synth.wall_depth(700), cam)])
    check(gv, ov, up, "camera inside, ypr %s" % (ypr,))
    assert up[0][1] > 0 and up[1][1] > 0
    # voxels BEHIND the camera are updated too (Q2): the oracle's weights say so, and the GPU agreed above
    w = ov.weight.reshape(128, 96, 192)
    ip = cam.inverse_pose().astype(np.float64).reshape(4, 4).T
    zz, yy, xx = np.nonzero(w > 0)
    vs = np.array([3000 / 192, 3000 / 96, 3000 / 128])
    camz = ip[2, 0] * (xx + 0.5) * vs[0] + ip[2, 1] * (yy + 0.5) * vs[1] + ip[2, 2] * (zz + 0.5) * vs[2] + ip[2, 3]
    assert (camz < 0).any(), "no voxel behind the camera was updated: the case does not reach the back half of the cone"


def test_camera_exactly_on_a_voxel_centre(oracle):
    # the voxel under the camera projects to 0 / 0: NaN -> pixel (0, 0) (Q3); its brick straddles the camera plane
    n, phys = 128, 3200.0            # voxel size 25: centres are exact floats
    depth = synth.wall_depth(900)
    for voxel in ((70, 40, 50), (64, 4, 32), (0, 0, 0), (127, 127, 127)):
        pos = tuple((v + 0.5) * 25.0 for v in voxel)
        cam = camera_at(pos)
        gv, ov, up = run_both(oracle, (n, n, n), (phys, phys, phys), [(depth, cam)])
        check(gv, ov, up, "camera on voxel %s" % (voxel,))
        assert ov.weight.reshape(n, n, n)[voxel[2], voxel[1], voxel[0]] == 1.0, "the voxel under the camera takes pixel (0, 0)"


def test_camera_inside_with_a_general_projection(oracle):
    # skewed intrinsics and a projective last row: the straddlers' test is on the signs of affine forms and holds for any matrices
    d, _ = synth.depth_frame(2, 7, seed=23)
    cam0 = camera_at((1500.0, 1400.0, 1600.0), yaw_pitch_roll=(0.5, 0.3, -0.2))
    k = cam0.k().copy(); k[3] = 7.5; k[1] = -3.0          # column-major: m12, m21
    ip = cam0.inverse_pose().copy(); ip[3] = 1.0e-5; ip[7] = -2.0e-5
    from tests.helpers import Cam
    cam = Cam(cam0.pose(), ip, k, oracle.mat3_inverse(k))
    gv, ov, up = run_both(oracle, (192, 64, 96), (3000, 3000, 3000), [(d, cam)])
    check(gv, ov, up, "camera inside, general projection")
    assert up[0][1] > 0


class _Matrices:
    """A camera given by its four matrices (column-major), as the reference's integrate receives them (TSDFVolume.cu:867-877)."""

    def __init__(self, pose, inverse_pose, k, kinv):
        self._m = [np.ascontiguousarray(m, np.float32).reshape(-1) for m in (pose, inverse_pose, k, kinv)]

    def pose(self): return self._m[0]
    def inverse_pose(self): return self._m[1]
    def k(self): return self._m[2]
    def kinv(self): return self._m[3]


def test_integrate_equals_the_loop_around_the_reference_s_compiled_transforms(oracle):
    """tests/golden/ref_transforms.npz: distances and weights left by the loop of integrate_kernel around the REFERENCE's own compiled
    world_to_pixel / pixel_to_camera / world_to_camera (oracle/ref_transforms_wrap.cpp, generated in the build container): the HIP
    kernels' projection, rounding and gating decisions of every voxel against reference code, not against the restatement -- and,
    where oracle/_ref travelled to this box, against that library run here."""
    f = np.load(os.path.join(GOLD, "ref_transforms.npz"))
    k, kinv, poses = f["integrate_k"], f["integrate_kinv"], f["integrate_poses"]
    for i in range(int(f["n_integrate"])):
        size, phys = tuple(int(s) for s in f["integrate%d_size" % i]), tuple(float(p) for p in f["integrate%d_phys" % i])
        off0, off1 = f["integrate%d_offsets" % i]
        name = str(f["integrate%d_name" % i])
        for storage in (8, 32):
            gv = tsdf_amd.TSDFVolume(size, phys)
            gv.set_weight_storage(storage)
            gv.offset(*off0); gv.clear(); gv.offset(*off1)      # Q1: the offset of clear() is baked in, the offset of now added on top
            gv.set_counting(True)
            live = oracle.have_ref_transforms()
            if live:
                vs = (np.array(phys, np.float32) / np.array(size, np.float32)).astype(np.float32)
                rd = np.full(size[0] * size[1] * size[2], gv.truncation_distance(), np.float32)
                rw = np.zeros_like(rd)
            for d, pi, u in zip(f["integrate%d_depths" % i], f["integrate%d_pose_index" % i], f["integrate%d_updates" % i]):
                cam = _Matrices(poses[pi], oracle.mat4_inverse(poses[pi]), k, kinv)
                gv.integrate(d, 160, 120, cam)
                assert gv.last_updated_voxels() == int(u), name
                if live:
                    assert oracle.ref_integrate_composed(rd, rw, size, vs, gv.truncation_distance(), cam.inverse_pose(), k, kinv, d, 160, 120, off0, off1) == int(u)
            assert_same_floats(gv.get_weight_data(), f["integrate%d_weight" % i], name + " weights (storage %d)" % storage)
            assert_same_floats(gv.get_distance_data(), f["integrate%d_dist" % i], name + " distances (storage %d)" % storage)
            if live:
                assert_same_floats(gv.get_distance_data(), rd, name + " distances against the library run here")
                assert_same_floats(gv.get_weight_data(), rw, name + " weights against the library run here")
