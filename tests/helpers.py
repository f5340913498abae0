"""Shared builders for the parity tests (inputs only; no expectations live here)."""
import numpy as np

import tsdf_amd
from tsdf_amd import synth

W, H = 640, 480


class Cam:
    """Camera stand-in built from raw matrices (anything with pose/inverse_pose/k/kinv works)."""

    def __init__(self, pose, inverse_pose, k, kinv):
        self._p, self._ip, self._k, self._ki = (np.asarray(a, np.float32).reshape(-1) for a in (pose, inverse_pose, k, kinv))

    def pose(self):
        return self._p

    def inverse_pose(self):
        return self._ip

    def k(self):
        return self._k

    def kinv(self):
        return self._ki


def camera_at(position, look_at=None, yaw_pitch_roll=None):
    cam = tsdf_amd.Camera.default_depth_camera()
    if yaw_pitch_roll is not None:
        y, p, r = yaw_pitch_roll
        cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        M = np.eye(4)
        M[:3, :3] = Ry @ Rx @ Rz
        M[:3, 3] = position
        cam.set_pose_rows(M)
    else:
        cam.move_to(*position)
        if look_at is not None:
            cam.look_at(*look_at)
    return cam


def sphere_depth_map(width, height, radius, min_depth, max_depth):
    """Same construction as the reference's make_sphere_depth_map fixture generator
    (src/Tests/TestTSDF/TestHelpers.cpp:144-183): a hemisphere bulging towards the camera, 0 elsewhere."""
    cx, cy = width / 2.0, height / 2.0
    xs, ys = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
    dx2, dy2 = (cx - xs) ** 2, (cy - ys) ** 2
    r2 = np.float32(radius * radius)
    half = (max_depth - min_depth) / 2.0
    centre = min_depth + half
    inside = dx2 + dy2 < r2
    dz = np.sqrt(np.maximum(r2 - (dx2 + dy2), 0))
    d = np.where(inside, np.clip((centre - dz).astype(np.int64), min_depth, max_depth), 0)
    return d.astype(np.uint16).reshape(-1)


def sphere_tsdf(O, size, physical, radius, offset=(0, 0, 0)):
    """Analytic sphere TSDF, the reference's create_sphere_in_TSDF (TestHelpers.cpp:17-60):
    d = clamp(|centre - voxel_centre| - r, +-trunc), centre = (offset + physical) / 2."""
    n = size
    g = O.Volume((n, n, n), (physical,) * 3)
    vs = g.voxel_size()
    trunc = np.float32(g.truncation_distance())
    off = np.asarray(offset, np.float32)
    centre = (off + np.float32(physical)) / np.float32(2.0)
    idx = (np.arange(n, dtype=np.float32) + np.float32(0.5))
    cx = off[0] + vs[0] * idx
    cy = off[1] + vs[1] * idx
    cz = off[2] + vs[2] * idx
    dx = (centre[0] - cx)[None, None, :]
    dy = (centre[1] - cy)[None, :, None]
    dz = (centre[2] - cz)[:, None, None]
    dist = np.sqrt(dx * dx + dy * dy + dz * dz).astype(np.float32) - np.float32(radius)
    return np.minimum(np.maximum(dist, -trunc), trunc).astype(np.float32).reshape(-1)


def assert_same_floats(a, b, what=""):
    """Bit-for-bit equality of float arrays (NaN patterns included)."""
    a = np.ascontiguousarray(a, np.float32).reshape(-1)
    b = np.ascontiguousarray(b, np.float32).reshape(-1)
    assert a.shape == b.shape, what
    ai, bi = a.view(np.uint32), b.view(np.uint32)
    same = (ai == bi) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        bad = np.flatnonzero(~same)
        raise AssertionError("%s: %d of %d values differ; first at %d: %r vs %r" %
                             (what, bad.size, a.size, bad[0], a[bad[0]], b[bad[0]]))
