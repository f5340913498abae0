"""The product's C++ Camera (tsdf_amd/host, reached through the Python mirror) against the known answers of
the reference's Test_Camera.cpp.  Host-only code: runs without a GPU."""
import math

import numpy as np
import pytest

import tsdf_amd

EPS = 1e-6
WORLD = [(0, 0, 0), (100, 0, 0), (100, 100, 0), (0, 100, 0), (0, 100, 100), (0, 0, 100), (100, 0, 100), (100, 100, 100)]


def cam():
    return tsdf_amd.Camera.default_depth_camera()


@pytest.mark.parametrize("target,expect", [
    (None, lambda w: (w[0], w[1], w[2])),                 # Test_Camera.cpp:35-52
    ((-1, 0, 0), lambda w: (w[2], w[1], -w[0])),          # :54-71
    ((0, -1, 0), lambda w: (w[0], w[2], -w[1])),          # :73-89
    ((0, 1, 0), lambda w: (w[0], -w[2], w[1])),           # :91-107
    ((1, 0, 0), lambda w: (-w[2], w[1], w[0])),           # :109-125
    ((0, 0, -1), lambda w: (-w[0], w[1], -w[2])),         # :127-143
])
def test_world_to_camera_for_axis_aligned_views(target, expect):
    c = cam()
    if target:
        c.look_at(*target)
    for w in WORLD:
        assert np.allclose(c.world_to_camera(w), expect(w), atol=EPS)


@pytest.mark.parametrize("pos", [(100, 0, 0), (0, 100, 0), (0, 0, 100)])     # :146-197
def test_world_to_camera_translation_only(pos):
    c = cam()
    c.move_to(*pos)
    for w in WORLD:
        assert np.allclose(c.world_to_camera(w), np.subtract(w, pos), atol=EPS)


def test_pixel_image_plane_round_trip_at_the_image_corners():      # :226-323
    c = cam()
    for px, ex, ey in [((0, 0), -0.5, -0.5), ((640, 0), 0.5, -0.5), ((0, 480), -0.5, 0.5), ((640, 480), 0.5, 0.5)]:
        ip = c.pixel_to_image_plane(*px)
        assert abs(ip[0] - ex) < 0.1 and abs(ip[1] - ey) < 0.11
        assert c.image_plane_to_pixel(ip) == px


def test_default_pose_is_identity_and_move_to_sets_translation():  # :328-351
    c = cam()
    assert np.array_equal(c.pose().reshape(4, 4), np.eye(4, dtype=np.float32))
    c.move_to(100.0, 200.0, 300.0)
    assert tuple(c.pose()[12:15]) == (100.0, 200.0, 300.0)


def _y_rot(theta, pos):
    c, s = np.float32(math.cos(theta)), np.float32(math.sin(theta))
    return np.array([[c, 0, s, pos[0]], [0, 1, 0, pos[1]], [-s, 0, c, pos[2]], [0, 0, 0, 1]], np.float32)


def _x_rot(theta, pos):
    c, s = np.float32(math.cos(theta)), np.float32(math.sin(theta))
    return np.array([[1, 0, 0, pos[0]], [0, c, -s, pos[1]], [0, s, c, pos[2]], [0, 0, 0, 1]], np.float32)


@pytest.mark.parametrize("pos,expected", [
    ((0, 0, 100), _y_rot(-math.pi, (0, 0, 100))),          # :355-374
    ((100, 0, 0), _y_rot(-math.pi / 2, (100, 0, 0))),      # :377-396
    ((-100, 0, 0), _y_rot(math.pi / 2, (-100, 0, 0))),     # :398-416
    ((0, 0, -100), _y_rot(0, (0, 0, -100))),               # :418-436
    ((0, 100, 0), _x_rot(math.pi / 2, (0, 100, 0))),       # :438-456
    ((0, -100, 0), _x_rot(-math.pi / 2, (0, -100, 0))),    # :458-476
])
def test_look_at_origin_builds_the_expected_pose(pos, expected):
    c = cam()
    c.move_to(*pos)
    c.look_at(0, 0, 0)
    assert np.allclose(c.pose().reshape(4, 4).T, expected, atol=EPS)


def test_set_pose_stores_the_matrix_verbatim():                    # :480-493
    c = cam()
    rows = np.arange(1, 17, dtype=np.float32).reshape(4, 4)
    c.set_pose_rows(rows)
    assert np.array_equal(c.pose().reshape(4, 4).T, rows)


def test_inverse_pose_times_pose_is_identity_for_rigid_poses():
    c = cam()
    c.move_to(1234.5, -250.0, 777.0)
    c.look_at(100, 200, 3000)
    P = c.pose().reshape(4, 4).T.astype(np.float64)
    I = c.inverse_pose().reshape(4, 4).T.astype(np.float64)
    assert np.allclose(P @ I, np.eye(4), atol=2e-4)
    # bottom row of a rigid inverse is exactly (0,0,0,1): integrate's w-divide is then exact
    assert tuple(c.inverse_pose()[3::4]) == (0.0, 0.0, 0.0, 1.0)


def test_product_camera_agrees_with_the_oracle_camera(oracle):
    c = cam()
    k, kinv = oracle.camera_k()
    assert np.array_equal(c.k(), k) and np.array_equal(c.kinv(), kinv)
    c.move_to(1500, 1300, -600)
    c.look_at(1500, 1400, 1900)
    p = oracle.look_at(oracle.identity_pose((1500, 1300, -600)), (1500, 1400, 1900))
    assert np.array_equal(c.pose(), p)
    assert np.array_equal(c.inverse_pose(), oracle.mat4_inverse(p))
