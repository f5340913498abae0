"""Generates the committed fixtures under tests/golden/.  Run from the repo root IN THE BUILD CONTAINER:

    python -m tests.golden.make_golden

  bilateral_ref_u8.npz         inputs + outputs of the REFERENCE's own BilateralFilter (oracle/_ref/libref_bilateral.so, built
                               from /root/reference/src/BilateralFilter.cpp where it lies).  Data only.
  oracle_integrate_raycast.npz small integrate + raycast cases produced by the CPU oracle (oracle/), so the
                               GPU box can check the HIP path against stored vectors as well as live.

The reference's CUDA integrate/raycast cannot be executed here (no nvcc / CUDA headers / Eigen), so those
vectors are oracle-generated and say so in their name.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def bilateral_cases():
    rng = np.random.RandomState(1234)
    imgs = []
    ramp = (np.arange(16)[None, :] * 12 + np.arange(8)[:, None] * 5).astype(np.uint8)
    noisy = np.clip(np.tile(np.linspace(10, 240, 64), (48, 1)) + rng.randint(-15, 16, (48, 64)), 0, 255).astype(np.uint8)
    blocks = rng.randint(0, 256, (6, 8)).astype(np.uint8).repeat(7, 0).repeat(9, 1)[:40, :70]
    speck = np.full((21, 19), 128, np.uint8)
    speck[rng.randint(0, 21, 30), rng.randint(0, 19, 30)] = rng.randint(0, 256, 30)
    for img in (ramp, noisy, np.ascontiguousarray(blocks), speck):
        for sig in ((3.0, 2.0), (30.0, 4.5)):
            imgs.append((img, sig))
    return imgs


def integrate_raycast_case(O, name):
    """Small deterministic cases; returns dist/weight after integration and the raycast from the same pose."""
    k, kinv = O.camera_k()
    W, H = 160, 120
    k2, kinv2 = O.camera_k(591.1 / 4, 590.1 / 4, 331.0 / 4, 234.6 / 4)
    v = O.Volume((32, 32, 32), (3000, 3000, 3000))
    if name == "wall32":
        pose = O.identity_pose((1500, 1500, -1000))
        depth = np.full(W * H, 2500, np.uint16)
        frames = [(depth, pose)]
    elif name == "rot32":
        rng = np.random.RandomState(5)
        frames = []
        for i in range(3):
            pose = O.look_at(O.identity_pose((900 + 300 * i, 1700 - 200 * i, -700)), (1500, 1500, 1500))
            yy, xx = np.mgrid[0:H, 0:W]
            depth = (2200 + 150 * np.sin(xx / 17.0 + i) + 120 * np.cos(yy / 11.0)).astype(np.uint16)
            depth[rng.rand(H, W) < 0.02] = 0
            frames.append((depth.reshape(-1), pose))
    elif name == "spheredepth32":
        # hemispherical bulge in front of the camera, like make_sphere_depth_map (TestHelpers.cpp:144-183)
        pose = O.identity_pose((1500, 1500, -500))
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        r2 = (W / 2.0 - xx) ** 2 + (H / 2.0 - yy) ** 2
        depth = np.where(r2 < 50.0 ** 2, np.clip(1700.0 - np.sqrt(np.maximum(50.0 ** 2 - r2, 0)) * 8, 1200, 2200), 0)
        frames = [(depth.astype(np.uint16).reshape(-1), pose)]
    else:
        raise KeyError(name)
    for depth, pose in frames:
        v.integrate(depth, W, H, O.mat4_inverse(pose), k2, kinv2)
    V, N = v.raycast(W, H, frames[0][1], kinv2, nthreads=O.max_threads())
    return {"dist": v.dist.copy(), "weight": v.weight.copy(), "vertices": V, "normals": N}


def sphere_raycast_case(O, cam_pos):
    """Analytic sphere TSDF (create_sphere_in_TSDF, TestHelpers.cpp:17-60) in a 256 mm cube of 64^3 voxels, ray cast at
    160x120 from one of the two poses of Test_TSDF_RayCast.cpp:430-431 / :580-581."""
    n, phys, radius = 64, 256.0, 80.0
    v = O.Volume((n, n, n), (phys,) * 3)
    vs, trunc = v.voxel_size(), np.float32(v.truncation_distance())
    c = (np.arange(n, dtype=np.float32) + np.float32(0.5)) * vs[0]
    centre = np.float32(phys) / np.float32(2.0)
    dx, dy, dz = (centre - c)[None, None, :], (centre - c)[None, :, None], (centre - c)[:, None, None]
    d = np.sqrt(dx * dx + dy * dy + dz * dz).astype(np.float32) - np.float32(radius)
    v.set_distance_data(np.minimum(np.maximum(d, -trunc), trunc))
    k2, kinv2 = O.camera_k(591.1 / 4, 590.1 / 4, 331.0 / 4, 234.6 / 4)
    pose = O.look_at(O.identity_pose(cam_pos), (128, 128, 128))
    V, N = v.raycast(160, 120, pose, kinv2, nthreads=O.max_threads())
    return {"dist": v.dist.copy(), "pose": pose, "kinv": kinv2, "vertices": V, "normals": N}


def main():
    import oracle as O
    O.build(force=True)
    assert O.have_ref(), "reference build missing: run `make -C oracle ref` with /root/reference mounted"
    out = {}
    cases = bilateral_cases()
    out["count"] = np.int32(len(cases))
    for i, (img, sig) in enumerate(cases):
        h, w = img.shape
        out["in_%d" % i] = img
        out["sigmas_%d" % i] = np.array(sig, np.float32)
        out["out_%d" % i] = O.ref_bilateral_u8(img, w, h, *sig)
    np.savez_compressed(os.path.join(HERE, "bilateral_ref_u8.npz"), **out)

    out = {}
    for name in ("wall32", "rot32", "spheredepth32"):
        for key, val in integrate_raycast_case(O, name).items():
            out[name + "_" + key] = val
    np.savez_compressed(os.path.join(HERE, "oracle_integrate_raycast.npz"), **out)
    out = {}
    for tag, pos in (("a", (450, 150, 150)), ("b", (-150, 150, 450))):
        for key, val in sphere_raycast_case(O, pos).items():
            out[tag + "_" + key] = val
    np.savez_compressed(os.path.join(HERE, "oracle_sphere_raycast.npz"), **out)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
