"""BASELINE configs[4] at its full size: 512^3 volume, closed tracking loop (bilateral filter, model image rendered from the
previous pose, ICP replacing third_party/ICP_CUDA, integrate) over 24 frames of the synthetic fr1/desk surrogate, then the
marching-cubes mesh of the tracked model.  Asserted here, not just run inside bench.py:
  * the tracked trajectory stays on the true one;
  * the ICP answer of a frame pair taken out of the running loop equals the CPU oracle's for the same two images;
  * the mesh: device extract_surface == host marching cubes == the oracle's restatement of the reference's loop, bit for bit.
"""
import numpy as np
import pytest

import tsdf_amd
from tsdf_amd import synth
from tsdf_amd.tracking import FrameToModelTracker

pytestmark = pytest.mark.gpu
W, H = synth.WIDTH, synth.HEIGHT
N, FRAMES, STREAM, SEED = 512, 24, 200, 0x5EED0003


def rotation_angle(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))


@pytest.fixture(scope="module")
def tracked():
    vol = tsdf_amd.TSDFVolume((N, N, N), (3000.0,) * 3)
    tracker = FrameToModelTracker(vol, W, H)
    log, pair = [], None
    for i in range(FRAMES):
        depth, cam = synth.depth_frame(i, STREAM, seed=SEED)
        truth = cam.pose().astype(np.float64).reshape(4, 4).T
        pose = tracker.process(depth, initial_pose=truth if i == 0 else None)
        log.append((pose, truth, tracker.last_inliers))
        if i == 13:
            pair = (tracker.last_icp_inputs(), tracker.last_T.copy())
    return vol, log, pair


def test_the_tracked_trajectory_follows_the_true_one(tracked):
    _, log, _ = tracked
    worst_t = max(float(np.linalg.norm(p[:3, 3] - t[:3, 3])) for p, t, _ in log)
    worst_r = max(rotation_angle(p[:3, :3], t[:3, :3]) for p, t, _ in log)
    moved = float(np.linalg.norm(log[-1][1][:3, 3] - log[0][1][:3, 3]))
    assert moved > 50.0                       # the camera really moved (mm)
    assert worst_t < 10.0, worst_t            # voxels are 5.9 mm, depth noise +-3 mm
    assert worst_r < 0.006, worst_r
    for p, _, inliers in log[1:]:
        assert np.allclose(p[:3, :3] @ p[:3, :3].T, np.eye(3), atol=1e-5)
        assert inliers > 0.3 * W * H


def test_icp_of_a_frame_pair_from_the_loop_equals_the_oracle(tracked, oracle):
    _, _, ((model, current), T) = tracked
    assert (model > 0).mean() > 0.5 and (current > 0).mean() > 0.5
    To, _, inliers = oracle.icp_incremental_transformation(current, model, W, H, 331.0, 234.6, 591.1, 590.1)
    # the 29 sums are fp32 partial sums in a fixed order on the device, double in the oracle: the poses agree far inside 1e-4
    assert np.max(np.abs(T - To)) < 1e-6, np.max(np.abs(T - To))
    assert inliers > 0.3 * W * H


def test_the_mesh_of_the_tracked_model(tracked, oracle):
    vol, _, _ = tracked
    mesh_dev = vol.extract_surface()
    dist = vol.get_distance_data()
    mesh_host = tsdf_amd.marching_cubes(dist, (N, N, N), (3000.0 / N,) * 3)
    assert mesh_dev.shape == mesh_host.shape and mesh_dev.shape[0] % 3 == 0
    assert mesh_dev.shape[0] // 3 > 500_000           # a room-sized surface at 5.9 mm voxels
    assert np.array_equal(mesh_dev.view(np.uint32), mesh_host.view(np.uint32))
    mesh_orc = oracle.marching_cubes(dist, (N, N, N), (3000.0 / N,) * 3, (0.0, 0.0, 0.0), nthreads=oracle.max_threads())
    assert mesh_orc.shape == mesh_dev.shape
    assert np.array_equal(mesh_dev.view(np.uint32), mesh_orc.view(np.uint32))
