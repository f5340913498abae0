"""Synthetic RGB-D stream: the TUM-dataset surrogate of SURVEY.md section 8d (no TUM data in the image).

An analytic scene (back wall, sphere, box) inside the default 3000 mm volume is ray traced
exactly from a camera moving on a small closed trajectory; the result is quantised to uint16
millimetres with integer-hash sensor noise and dropouts (0 = invalid), i.e. what
TUMDataLoader + DepthImage::scale_depth(0.2) hand to TSDFVolume::integrate in the reference
(src/DataLoader/TUMDataLoader.cpp:84-99).  Everything is input DATA for the path; numpy only.
"""
import math

import numpy as np

from .api import Camera

WIDTH, HEIGHT = 640, 480

# scene, world millimetres (volume spans 0..3000 on every axis)
WALL_Z = 2400.0
SPHERE_C = np.array([1500.0, 1400.0, 1800.0])
SPHERE_R = 350.0
BOX_MIN = np.array([700.0, 1700.0, 1900.0])
BOX_MAX = np.array([1200.0, 2300.0, 2400.0])
LOOK_AT = (1500.0, 1400.0, 1900.0)


def splitmix64(x):
    """splitmix64 finaliser on uint64 arrays (explicit integer hash so every platform agrees)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def camera_position(i, n):
    """Closed trajectory in front of the volume."""
    a = 2.0 * math.pi * i / n
    return (1500.0 + 200.0 * math.sin(a), 1300.0 + 100.0 * math.sin(2.0 * a), -600.0 + 150.0 * math.cos(a))


def camera_position_inside(i, n):
    """Closed trajectory INSIDE the volume, one metre in front of the wall (BASELINE configs[3]: at 1024^3 a ray covers only
    4402 * 0.279 mm = 1228 mm, Q8, so the scene has to be that close)."""
    a = 2.0 * math.pi * i / n
    return (1500.0 + 120.0 * math.sin(a), 1350.0 + 60.0 * math.sin(2.0 * a), 1400.0 + 80.0 * math.cos(a))


def camera_for_frame(i, n, camera=None, inside=False):
    cam = camera or Camera.default_depth_camera()
    cam.set_pose(np.eye(4, dtype=np.float32).reshape(-1))
    cam.move_to(*(camera_position_inside(i, n) if inside else camera_position(i, n)))
    cam.look_at(*LOOK_AT)
    return cam


def trace_depth(camera, width=WIDTH, height=HEIGHT):
    """Exact camera-z depth (float64 mm, inf = nothing hit) of the analytic scene."""
    kinv = camera.kinv().astype(np.float64).reshape(3, 3).T
    pose = camera.pose().astype(np.float64).reshape(4, 4).T
    R, o = pose[:3, :3], pose[:3, 3]
    xs, ys = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    pix = np.stack([xs, ys, np.ones_like(xs)], axis=-1)               # (H, W, 3)
    d_cam = pix @ kinv.T                                              # camera-space ray, z == 1
    d = d_cam @ R.T                                                   # world-space, t == camera z
    best = np.full(xs.shape, np.inf)

    # wall z = WALL_Z
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (WALL_Z - o[2]) / d[..., 2]
    ok = (t > 0) & np.isfinite(t)
    best = np.where(ok & (t < best), t, best)

    # sphere
    oc = o - SPHERE_C
    a = np.sum(d * d, axis=-1)
    b = 2.0 * (d @ oc)
    c = float(oc @ oc) - SPHERE_R ** 2
    disc = b * b - 4.0 * a * c
    with np.errstate(invalid="ignore"):
        t = (-b - np.sqrt(disc)) / (2.0 * a)
    ok = (disc >= 0) & (t > 0)
    best = np.where(ok & (t < best), t, best)

    # box (slab method)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (BOX_MIN - o) / d
        t1 = (BOX_MAX - o) / d
    tn = np.max(np.minimum(t0, t1), axis=-1)
    tf = np.min(np.maximum(t0, t1), axis=-1)
    ok = (tn <= tf) & (tn > 0)
    best = np.where(ok & (tn < best), tn, best)
    return best


def depth_frame(i, n, seed, width=WIDTH, height=HEIGHT, camera=None, noise=True, inside=False):
    """-> (depth uint16 (H*W,), camera) for frame i of an n-frame stream (inside: the config-4 trajectory within the volume)."""
    cam = camera_for_frame(i, n, camera, inside)
    z = trace_depth(cam, width, height)
    mm = np.where(np.isfinite(z), np.rint(z), 0.0)
    if noise:
        idx = (np.arange(width * height, dtype=np.uint64).reshape(height, width)
               + np.uint64(i) * np.uint64(width * height))
        h = splitmix64(idx ^ np.uint64(seed))
        jitter = (h % np.uint64(7)).astype(np.int64) - 3                 # +-3 mm
        drop = ((h >> np.uint64(20)) % np.uint64(1000)) < np.uint64(15)  # 1.5 % dropouts
        mm = np.where(mm > 0, mm + jitter, 0.0)
        mm = np.where(drop, 0.0, mm)
    depth = np.clip(mm, 0, 65535).astype(np.uint16).reshape(-1)
    return depth, cam


def wall_depth(mm=2500, width=WIDTH, height=HEIGHT):
    """The survey probe's frame: a wall at constant depth everywhere."""
    return np.full(width * height, mm, np.uint16)


def config1_depth(seed=1, width=WIDTH, height=HEIGHT):
    """BASELINE config 1 frame: wall at 2500 mm, a centred bulge of radius 120 px, hash dropouts."""
    xs, ys = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    dx, dy = xs - width / 2.0, ys - height / 2.0
    r2 = dx * dx + dy * dy
    d = np.full((height, width), 2500.0)
    inside = r2 < 120.0 ** 2
    bulge = 2000.0 - np.sqrt(np.maximum(120.0 ** 2 - r2, 0.0)) * 2.0
    d = np.where(inside, np.maximum(bulge, 400.0), d)
    h = splitmix64((np.arange(width * height, dtype=np.uint64).reshape(height, width)) ^ np.uint64(seed))
    d = np.where(h % np.uint64(50) == 0, 0.0, d)
    return np.rint(d).astype(np.uint16).reshape(-1)


def _write_png16(path, image):
    """16-bit greyscale PNG (big-endian samples, filter 0), enough for TUM-style depth maps."""
    import struct
    import zlib
    h, w = image.shape
    raw = b"".join(b"\x00" + image[y].astype(">u2").tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        f.write(chunk(b"IEND", b""))


def write_tum_directory(directory, n_frames, seed, stream_frames=None, width=WIDTH, height=HEIGHT):
    """Writes the layout the reference's TUMDataLoader expects (src/DataLoader/TUMDataLoader.cpp:20,111-128):
    <dir>/depth/<stem>.png (uint16, 5 units per mm) and <dir>/ground_truth.txt with lines
    '<stem> tx ty tz qx qy qz qw' (metres, TUM quaternion order).  Returns the per-frame (depth_mm, pose16)."""
    import os
    os.makedirs(os.path.join(directory, "depth"), exist_ok=True)
    frames = []
    lines = ["# synthetic TUM surrogate (tsdf_amd.synth), seed 0x%X" % seed]
    total = stream_frames or n_frames
    for i in range(n_frames):
        depth, cam = depth_frame(i, total, seed, width, height)
        stem = "%010.6f" % (1305031102.0 + i / 30.0)
        _write_png16(os.path.join(directory, "depth", stem + ".png"), (depth.astype(np.uint32) * 5).clip(0, 65535)
                     .astype(np.uint16).reshape(height, width))
        P = cam.pose().astype(np.float64).reshape(4, 4).T
        R, t = P[:3, :3], P[:3, 3] / 1000.0
        # rotation -> quaternion (w largest-branch free form is enough for these small rotations)
        qw = math.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
        qx = (R[2, 1] - R[1, 2]) / (4.0 * qw)
        qy = (R[0, 2] - R[2, 0]) / (4.0 * qw)
        qz = (R[1, 0] - R[0, 1]) / (4.0 * qw)
        lines.append("%s %.9f %.9f %.9f %.9f %.9f %.9f %.9f" % (stem, t[0], t[1], t[2], qx, qy, qz, qw))
        frames.append((depth, cam.pose().copy()))
    with open(os.path.join(directory, "ground_truth.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return frames
