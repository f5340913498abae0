"""Generates tests/golden/ref_transforms.npz from the REFERENCE's own code.  Run from the repo root in the build container:

    python -m tests.golden.make_ref_transform_vectors

The outputs come from oracle/_ref/libref_transforms.so, i.e. /root/reference/src/Utilities/cuda_coordinate_transforms.cu (world_to_pixel,
pixel_to_camera, world_to_camera) and the helpers of src/include/cuda_utilities.hpp compiled where they lie with g++ against the CUDA
toolkit headers of this image (oracle/Makefile target "ref"; oracle/ref_transforms_wrap.cpp has the C entry points).  Data only: inputs
and the reference's outputs.  The integrate cases are the loop of integrate_kernel restated around those compiled functions
(ref_integrate_composed): they pin the projection, rounding and gating of every voxel on reference code, not its blend lines.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cameras(O):
    """(name, pose, k, kinv): the default depth camera and some that exercise every matrix entry."""
    out = []
    k, kinv = O.camera_k()
    out.append(("default_identity", O.identity_pose((1500, 1500, -1000)), k, kinv))
    out.append(("default_look_at", O.look_at(O.identity_pose((400, 2100, -650)), (1500, 1400, 1600)), k, kinv))
    k2, kinv2 = O.camera_k(525.0, 400.0, 319.5, 239.5)
    out.append(("kinect_rgb_like", O.look_at(O.identity_pose((2900, 300, 3300)), (1500, 1500, 1500)), k2, kinv2))
    # a general (sheared, scaled) pose and a full 3 x 3 K: nothing in the reference's arithmetic assumes rigidity
    rng = np.random.RandomState(77)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = (np.eye(3) + 0.2 * rng.randn(3, 3)).astype(np.float32)
    pose[:3, 3] = (1200, 1700, -300)
    pose[3, :] = (1e-5, -2e-5, 3e-5, 1.001)     # a projective last row: world_to_camera divides by w (:118-120)
    kf = (np.array([[500, 3, 300], [-2, 480, 250], [1e-4, -1e-4, 1.0]]) ).astype(np.float32)
    out.append(("general_matrices", pose.T.reshape(-1).copy(), kf.T.reshape(-1).copy(), np.linalg.inv(kf.astype(np.float64)).astype(np.float32).T.reshape(-1).copy()))
    return out


def points(rng, n):
    p = (rng.rand(n, 3) * 3600 - 300).astype(np.float32)                     # in and around a 3 m volume
    p[: n // 20] = (rng.randn(n // 20, 3) * 5000).astype(np.float32)         # far away, behind the camera
    p[n // 20: n // 10, 2] = (rng.randn(n // 10 - n // 20) * 2 - 1000).astype(np.float32)   # next to the first camera's plane: huge quotients
    return p


def integrate_cases(O):
    """(name, size, physical, offset_at_clear, offset_now, frames[(depth, width, height, camera index)])"""
    rng = np.random.RandomState(9)
    W, H = 160, 120
    cases = []
    yy, xx = np.mgrid[0:H, 0:W]
    wall = np.full((H, W), 2500, np.uint16)
    bumpy = [(2200 + 150 * np.sin(xx / 17.0 + i) + 120 * np.cos(yy / 11.0)).astype(np.uint16) for i in range(3)]
    for b in bumpy:
        b[rng.rand(H, W) < 0.02] = 0
    cases.append(("wall32", (32, 32, 32), (3000.0, 3000.0, 3000.0), (0, 0, 0), (0, 0, 0), [(wall, 0)]))
    cases.append(("bumpy32x3", (32, 32, 32), (3000.0, 3000.0, 3000.0), (0, 0, 0), (0, 0, 0), [(bumpy[0], 0), (bumpy[1], 1), (bumpy[2], 0)]))
    cases.append(("offset_q1_24x40x32", (24, 40, 32), (2400.0, 3000.0, 2800.0), (100.0, -50.0, 25.0), (130.0, -50.0, -10.0), [(bumpy[1], 1), (bumpy[2], 0)]))
    return cases, (W, H)


def main():
    import oracle as O
    O.build()
    assert O.have_ref_transforms(), "oracle/_ref/libref_transforms.so is missing: make -C oracle ref (needs /root/reference)"
    rng = np.random.RandomState(20261001)
    out = {}
    cams = cameras(O)
    out["n_cameras"] = len(cams)
    for ci, (name, pose, k, kinv) in enumerate(cams):
        pose = np.asarray(pose, np.float32).reshape(-1)
        inv_pose = O.mat4_inverse(pose)
        p = points(rng, 5000)
        out["cam%d_name" % ci] = name
        out["cam%d_pose" % ci], out["cam%d_inv_pose" % ci], out["cam%d_k" % ci], out["cam%d_kinv" % ci] = pose, inv_pose, k, kinv
        out["cam%d_points" % ci] = p
        out["cam%d_world_to_pixel" % ci] = O.ref_world_to_pixel(p, inv_pose, k)
        out["cam%d_world_to_camera" % ci] = O.ref_world_to_camera(p, inv_pose)
        pix = np.stack([rng.randint(-50, 700, 5000), rng.randint(-50, 530, 5000)], 1).astype(np.int32)
        depth = rng.randint(0, 8000, 5000).astype(np.float32)
        out["cam%d_pixels" % ci], out["cam%d_depth" % ci] = pix, depth
        out["cam%d_pixel_to_camera" % ci] = O.ref_pixel_to_camera(pix, depth, kinv)
        upix = np.stack([rng.randint(0, 640, 5000), rng.randint(0, 480, 5000)], 1).astype(np.uint16)
        out["cam%d_ray_pixels" % ci] = upix
        rot = pose.reshape(4, 4).T[:3, :3].T.reshape(-1).copy()    # column-major 3 x 3 of the pose (GPURaycaster.cu:449-455)
        out["cam%d_rot" % ci] = rot
        out["cam%d_ray_direction" % ci] = O.ref_ray_direction(upix, rot, kinv)
    cases, (W, H) = integrate_cases(O)
    k4, kinv4 = O.camera_k(591.1 / 4, 590.1 / 4, 331.0 / 4, 234.6 / 4)
    poses = [O.identity_pose((1500, 1500, -1000)), O.look_at(O.identity_pose((700, 1900, -800)), (1500, 1500, 1500))]
    out["integrate_k"], out["integrate_kinv"] = k4, kinv4
    out["integrate_poses"] = np.stack(poses)
    out["n_integrate"] = len(cases)
    for i, (name, size, phys, off0, off1, frames) in enumerate(cases):
        vs = (np.array(phys, np.float32) / np.array(size, np.float32)).astype(np.float32)   # TSDFVolume.cu:690
        v = O.Volume(size, phys)
        trunc = v.truncation_distance()
        n = size[0] * size[1] * size[2]
        dist = np.full(n, trunc, np.float32)
        weight = np.zeros(n, np.float32)
        updates = []
        for depth, pi in frames:
            updates.append(O.ref_integrate_composed(dist, weight, size, vs, trunc, O.mat4_inverse(poses[pi]), k4, kinv4, depth, W, H, off0, off1))
        out["integrate%d_name" % i] = name
        out["integrate%d_size" % i], out["integrate%d_phys" % i] = np.array(size, np.int32), np.array(phys, np.float32)
        out["integrate%d_offsets" % i] = np.array([off0, off1], np.float32)
        out["integrate%d_depths" % i] = np.stack([d for d, _ in frames])
        out["integrate%d_pose_index" % i] = np.array([pi for _, pi in frames], np.int32)
        out["integrate%d_updates" % i] = np.array(updates, np.int64)
        out["integrate%d_dist" % i], out["integrate%d_weight" % i] = dist, weight
    np.savez_compressed(os.path.join(HERE, "ref_transforms.npz"), **out)
    print("wrote ref_transforms.npz:", len(out), "arrays,", os.path.getsize(os.path.join(HERE, "ref_transforms.npz")), "bytes")


if __name__ == "__main__":
    main()
