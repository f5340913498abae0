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
    # ... and the one-call variant (tsdf_raycast_depth_device: the depth formed in the resolve kernel, no vertex map in between)
    rc = tsdf_amd.GPURaycaster(W, H)
    direct = torch.empty((W * H,), dtype=torch.int16, device="cuda")
    verts = torch.empty((W * H, 3), dtype=torch.float32, device="cuda")
    for vp in (None, verts.data_ptr()):
        direct.zero_()
        rc.render_to_depth_device(vol, cam, direct.data_ptr(), vp)
        vol.synchronize()
        assert np.array_equal(direct.cpu().numpy().view(np.uint16), exp)
    same = (verts.cpu().numpy().view(np.uint32) == V.view(np.uint32)) | (np.isnan(verts.cpu().numpy()) & np.isnan(V))
    assert same.all()


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


def test_the_two_stream_loop_gives_the_poses_of_the_sequential_one_bit_for_bit():
    """tsdf_tracker_* with TSDF_PIPELINE_OVERLAP (the new frame's filter and ICP maps on a second stream beside the model ray cast,
    frames queued on the device without a host round trip but the one for T) against the same calls on one stream: the same poses,
    the same volume, bit for bit -- scheduling only."""
    import torch
    n, frames, stream_len = 128, 7, 200
    data = [synth.depth_frame(i, stream_len, seed=0x5EED0005) for i in range(frames)]
    dev = [torch.from_numpy(d.view(np.int16).copy()).cuda() for d, _ in data]
    torch.cuda.synchronize()
    runs = []
    for overlap in (False, True):
        vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
        tracker = FrameToModelTracker(vol, W, H, overlap=overlap)
        poses = []
        for i, (_, cam) in enumerate(data):
            truth = cam.pose().astype(np.float64).reshape(4, 4).T
            poses.append(tracker.process_device(dev[i].data_ptr(), initial_pose=truth if i == 0 else None))
        tracker.synchronize()
        runs.append((np.stack(poses), vol.get_distance_data(), vol.get_weight_data()))
        tracker.close()
        vol.close()
    assert np.array_equal(runs[0][0], runs[1][0]), np.max(np.abs(runs[0][0] - runs[1][0]))
    assert np.array_equal(runs[0][1].view(np.uint32), runs[1][1].view(np.uint32))
    assert np.array_equal(runs[0][2].view(np.uint32), runs[1][2].view(np.uint32))
