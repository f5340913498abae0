"""The kinfu loop closed with ICP (BASELINE configs[4] minus the mesh): frame-to-model tracking over the synthetic stream
must follow the ground-truth trajectory.  The parts are pinned against the oracle elsewhere (integrate, ray cast,
bilateral, ICP); this checks their composition -- the pose convention between ICPOdometry and the volume, metres vs
millimetres, model = render of the previous pose -- on a moving camera."""
import numpy as np
import pytest

import tsdf_amd
from tsdf_amd import synth
from tsdf_amd.tracking import FrameToModelTracker

pytestmark = pytest.mark.gpu
W, H = synth.WIDTH, synth.HEIGHT


def rotation_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))


def test_vertices_to_depth_matches_the_class_surface(oracle):
    import torch
    n = 64
    d, cam = synth.depth_frame(1, 30, seed=0x5EED0001)
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    vol.integrate(d, W, H, cam)
    V, _ = vol.raycast(W, H, cam)
    Vd = torch.from_numpy(V).cuda()
    out = torch.empty((W * H,), dtype=torch.int16, device="cuda")
    tsdf_amd.vertices_to_depth_device(W, H, Vd.data_ptr(), cam, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint16)
    # Camera::world_to_camera(v).z rounded half away from zero, in fp32, misses -> 0
    ip = cam.inverse_pose().reshape(4, 4).T.astype(np.float32)
    x, y, z = V[:, 0], V[:, 1], V[:, 2]
    cz = ((ip[2, 0] * x + ip[2, 1] * y) + ip[2, 2] * z) + ip[2, 3] * np.float32(1.0)
    cw = ((ip[3, 0] * x + ip[3, 1] * y) + ip[3, 2] * z) + ip[3, 3] * np.float32(1.0)
    with np.errstate(invalid="ignore"):
        q = cz / cw
        r = np.where(q >= 0, np.floor(q + np.float32(0.5)), np.ceil(q - np.float32(0.5)))   # q has few fraction bits: exact
        exp = np.where(np.isfinite(r) & (r > 0) & (r < 65536), r, 0).astype(np.uint16)
    assert np.array_equal(got, exp)
    assert (got > 0).mean() > 0.5


def test_tracking_follows_the_synthetic_trajectory():
    n, frames, stream_len = 256, 12, 200
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    tracker = FrameToModelTracker(vol, W, H)
    worst_t, worst_r = 0.0, 0.0
    moved = 0.0
    first = None
    for i in range(frames):
        depth, cam = synth.depth_frame(i, stream_len, seed=0x5EED0005)
        truth = cam.pose().astype(np.float64).reshape(4, 4).T
        pose = tracker.process(depth, initial_pose=truth if i == 0 else None)
        if first is None:
            first = truth
        worst_t = max(worst_t, float(np.linalg.norm(pose[:3, 3] - truth[:3, 3])))
        worst_r = max(worst_r, rotation_angle(pose[:3, :3], truth[:3, :3]))
        moved = float(np.linalg.norm(truth[:3, 3] - first[:3, 3]))
        assert np.allclose(pose[:3, :3] @ pose[:3, :3].T, np.eye(3), atol=1e-5)
        if i > 0:
            assert tracker.last_inliers > 0.3 * W * H
    assert moved > 50.0                      # the camera really moved (mm)
    assert worst_t < 15.0, worst_t           # tracked to within 1.5 cm (voxels are 11.7 mm, depth noise +-3 mm)
    assert worst_r < 0.01, worst_r
