"""GPU parity of the ICP tracking row (tsdf_amd/csrc/icp.hip) against the CPU oracle (oracle/icp_oracle.c).

Pyramid, vertex and normal maps are per-pixel arithmetic in the reference's order: bit-exact.  The 29 sums of a step are
accumulated in fp32 by the reference in an order that depends on its launch configuration, so they can only agree within
a tolerance: 1e-4 relative to the largest entry (north_star's figure), against the oracle's double sums."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import assert_same_floats
from tsdf_amd import synth

pytestmark = pytest.mark.gpu
W, H = synth.WIDTH, synth.HEIGHT
CX, CY, FX, FY = 331.0, 234.6, 591.1, 590.1


def two_views(i0=0, i1=2):
    # noise-free frames: raw one-pixel normals of a +-3 mm noisy depth map fail the 20 degree gate almost everywhere
    d0, cam0 = synth.depth_frame(i0, 200, seed=0x5EED0005, noise=False)
    d1, cam1 = synth.depth_frame(i1, 200, seed=0x5EED0005, noise=False)
    return d0, cam0, d1, cam1


def test_pyramid_and_maps_are_bit_exact(oracle):
    d0, _, d1, _ = two_views()
    icp = tsdf_amd.ICPOdometry(W, H, CX, CY, FX, FY)
    icp.init_icp_model(d0)
    icp.init_icp(d1)
    level_depth = d1.reshape(H, W)
    for level in range(3):
        rows, cols, div = H >> level, W >> level, 1 << level
        if level > 0:
            level_depth = oracle.icp_pyr_down(level_depth, rows * 2, cols * 2)
        assert np.array_equal(icp.get_depth_level(level), level_depth), "depth pyramid level %d" % level
        vo = oracle.icp_vmap(level_depth, rows, cols, np.float32(FX) / div, np.float32(FY) / div, np.float32(CX) / div,
                             np.float32(CY) / div)
        no = oracle.icp_nmap(vo, rows, cols)
        assert_same_floats(icp.get_map("vmap_curr", level), vo, "vmap level %d" % level)
        assert_same_floats(icp.get_map("nmap_curr", level), no, "nmap level %d" % level)
    # the model side goes through the same kernels
    vo = oracle.icp_vmap(d0.reshape(H, W), H, W, FX, FY, CX, CY)
    assert_same_floats(icp.get_map("vmap_prev", 0), vo, "model vmap")


def test_depth_cutoff_and_invalid_pixels(oracle):
    d = np.full((H, W), 1500, np.uint16)
    d[::7, ::5] = 0            # dropouts
    d[100:200, 100:300] = 30000  # 30 m: beyond the default 20 m cut-off
    icp = tsdf_amd.ICPOdometry(W, H, CX, CY, FX, FY)
    icp.init_icp(d, depth_cutoff=20.0)
    vo = oracle.icp_vmap(d, H, W, FX, FY, CX, CY, 20.0)
    v = icp.get_map("vmap_curr", 0)
    assert_same_floats(v, vo, "vmap with invalid pixels")
    assert np.isnan(v[:H][d == 0]).all() and np.isnan(v[:H][d == 30000]).all()
    assert_same_floats(icp.get_map("nmap_curr", 0), oracle.icp_nmap(vo, H, W), "nmap with invalid pixels")


@pytest.mark.parametrize("level", [0, 1, 2])
def test_one_step_matches_the_oracle_sums(oracle, level):
    d0, _, d1, _ = two_views()
    icp = tsdf_amd.ICPOdometry(W, H, CX, CY, FX, FY)
    icp.init_icp_model(d0)
    icp.init_icp(d1)
    rows, cols, div = H >> level, W >> level, 1 << level
    # a small non-trivial pose
    T = oracle.se3_exp([0.004, -0.003, 0.002, 0.002, -0.001, 0.0015])
    R, t = T[:3, :3].astype(np.float32), T[:3, 3].astype(np.float32)
    A, b, res, inl = icp.estimate_step(level, R, t)
    maps = [icp.get_map(k, level) for k in ("vmap_curr", "nmap_curr", "vmap_prev", "nmap_prev")]
    Ao, bo, reso, inlo, sums = oracle.icp_step(R.T.reshape(-1), t, *maps, rows, cols, np.float32(FX) / div, np.float32(FY) / div,
                                               np.float32(CX) / div, np.float32(CY) / div, 0.10, icp_angle())
    assert inl == inlo and inl > 0.3 * rows * cols                 # the same pixels pass the gates
    tol = 1e-4
    assert np.max(np.abs(A - Ao)) <= tol * np.max(np.abs(Ao))
    assert np.max(np.abs(b - bo)) <= tol * max(np.max(np.abs(bo)), 1e-6 * np.max(np.abs(Ao)))
    assert abs(res - reso) <= tol * reso
    assert np.array_equal(A, A.T)
    # deterministic: a second run gives the same bits
    A2, b2, res2, inl2 = icp.estimate_step(level, R, t)
    assert np.array_equal(A, A2) and np.array_equal(b, b2) and res == res2 and inl == inl2


def icp_angle():
    import math
    return float(np.float32(math.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))


def test_incremental_transformation_matches_the_oracle_and_the_motion(oracle):
    d0, cam0, d1, cam1 = two_views(0, 3)
    icp = tsdf_amd.ICPOdometry(W, H, CX, CY, FX, FY)
    icp.init_icp_model(d0)
    icp.init_icp(d1)
    T = icp.get_incremental_transformation()
    To, erro, inlo = oracle.icp_incremental_transformation(d1, d0, W, H, CX, CY, FX, FY)
    # 19 Gauss-Newton steps, each within 1e-4 of the oracle's sums: poses agree far inside a millimetre / 1e-4 rad
    assert np.max(np.abs(T[:3, 3] - To[:3, 3])) < 2e-4, (T, To)
    assert np.max(np.abs(T[:3, :3] - To[:3, :3])) < 1e-4
    assert abs(icp.last_inliers - inlo) <= 0.002 * inlo
    # and both recover the camera motion: T_prev_curr = inv(P_model) * P_current (metres)
    P0 = cam0.pose().astype(np.float64).reshape(4, 4).T
    P1 = cam1.pose().astype(np.float64).reshape(4, 4).T
    E = np.linalg.inv(P0) @ P1
    E[:3, 3] /= 1000.0
    assert np.max(np.abs(T[:3, 3] - E[:3, 3])) < 5e-3
    assert np.max(np.abs(T[:3, :3] - E[:3, :3])) < 5e-3
    # rigid
    assert np.allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-12) and np.allclose(T[3], [0, 0, 0, 1])


def test_bad_arguments():
    with pytest.raises(ValueError):
        tsdf_amd.ICPOdometry(0, 480, CX, CY, FX, FY)
    icp = tsdf_amd.ICPOdometry(W, H, CX, CY, FX, FY)
    with pytest.raises(ValueError):
        icp.init_icp(np.zeros(10, np.uint16))
    with pytest.raises(ValueError):
        icp.estimate_step(3, np.eye(3), np.zeros(3))


_PERSIST_PROBE = r"""
import sys, numpy as np
import tsdf_amd
from tsdf_amd import synth
W, H = synth.WIDTH, synth.HEIGHT
out = []
for (i0, i1) in ((0, 2), (5, 6), (10, 14)):
    d0, _ = synth.depth_frame(i0, 200, seed=0x5EED0005, noise=False)
    d1, _ = synth.depth_frame(i1, 200, seed=0x5EED0005, noise=False)
    icp = tsdf_amd.ICPOdometry(W, H, 331.0, 234.6, 591.1, 590.1)
    for rep in range(2):                      # twice on one object: the barrier's counter carries on from launch to launch
        icp.init_icp_model(d0); icp.init_icp(d1)
        T = icp.get_incremental_transformation()
        out.append(np.asarray(T, np.float64).reshape(-1))
        out.append(np.array([icp.last_error, icp.last_inliers], np.float64))
np.save(sys.argv[1], np.concatenate(out))
"""


def test_one_persistent_launch_gives_the_chain_of_launches_bit_for_bit(tmp_path):
    """Round 4: getIncrementalTransformation as ONE launch whose 256 workgroups stay through all 19 iterations and meet in a grid
    barrier (icp_persistent_kernel), in its two variants (TSDF_ICP_PERSISTENT=1 / 2), against the chain of 20 launches of rounds 1-3 (0, the default: the persistent
    launches measured slower): the same fixed-order
    sums, the same solve, so the same pose, residual and inlier count to the last bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for mode in ("1", "2", "0"):
        out = str(tmp_path / ("icp_%s.npy" % mode))
        e = dict(os.environ, TSDF_ICP_PERSISTENT=mode)
        e["PYTHONPATH"] = root + os.pathsep + e.get("PYTHONPATH", "")
        subprocess.run([sys.executable, "-c", _PERSIST_PROBE, out], check=True, env=e, cwd=root, timeout=600)
        got[mode] = np.load(out)
    assert got["1"].shape == got["0"].shape == got["2"].shape and got["1"].size == 3 * 2 * 18
    assert np.array_equal(got["1"].view(np.uint64), got["0"].view(np.uint64))          # every workgroup finishes each step
    assert np.array_equal(got["2"].view(np.uint64), got["0"].view(np.uint64))          # workgroup 0 finishes it and publishes the pose
    assert np.abs(got["0"][:16].reshape(4, 4)[:3, 3]).max() > 1e-4      # (a real motion was estimated)
