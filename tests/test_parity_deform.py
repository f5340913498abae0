"""GPU parity of TSDFVolume::deform_mesh (next row f5; reference src/TSDF/TSDFVolume.cu:101-291) against the oracle's
restatement, bit for bit: the trilinear blend of the deformation nodes' translations (with the reference's swapped
coefficients 6/7 and offset-free node lattice), the global rotation built from cos / sin products and the global
translation."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import assert_same_floats

pytestmark = pytest.mark.gpu


def sample_points(rng, dims, phys, offset, n=4000):
    lo = np.array(offset, np.float64)
    hi = lo + np.array(phys, np.float64)
    p = rng.uniform(lo - 0.02 * (hi - lo), hi + 0.02 * (hi - lo), size=(n, 3))
    # exact boundary values, the epsilon bands and the first / last half voxel
    p[:8] = [lo, hi, lo - 0.0005, hi + 0.0005, lo - 0.002, hi + 0.002, lo + 1e-4, hi - 1e-4]
    return p.astype(np.float32)


@pytest.mark.parametrize("custom", [False, True])
def test_deform_mesh_matches_the_oracle(oracle, custom):
    rng = np.random.default_rng(11 + custom)
    dims, phys, offset = (20, 17, 23), (400.0, 510.0, 345.0), (30.0, -20.0, 5.0)
    v = tsdf_amd.TSDFVolume(dims, phys)
    v.offset(*offset)
    v.clear()                                   # nodes: voxel centre + offset
    v.offset(12.0, 7.0, -3.0)                   # ... and the offset moved afterwards (Q1)
    rot, tr = (0.31, -0.2, 1.1), (15.0, -8.0, 120.0)
    v.set_global_transform(rot, tr)
    nodes = None
    if custom:
        n = dims[0] * dims[1] * dims[2]
        zz, yy, xx = np.mgrid[0:dims[2], 0:dims[1], 0:dims[0]]
        vs = v.voxel_size()
        t = np.stack([(xx + 0.5) * vs[0], (yy + 0.5) * vs[1], (zz + 0.5) * vs[2]], axis=-1).reshape(n, 3)
        t = t + 5.0 * np.sin(t / 60.0)          # a smooth non-rigid warp
        nodes = np.concatenate([t, rng.normal(size=(n, 3))], axis=1).astype(np.float32)
        v.set_deformation(nodes)
    pts = sample_points(rng, dims, phys, (12.0, 7.0, -3.0))
    got = v.deform_mesh(pts)
    exp = oracle.deform_points(dims, v.voxel_size(), (12.0, 7.0, -3.0), offset, nodes, rot, tr, pts)
    assert_same_floats(got, exp, "deform_mesh custom=%s" % custom)
    moved = np.any(got != pts, axis=1)
    assert 0.8 < moved.mean() < 1.0             # points outside the volume stay as they were
    assert np.array_equal(got[~moved], pts[~moved])


def test_identity_field_and_zero_transform_reproduce_interior_points(oracle):
    dims, phys = (16, 16, 16), (320.0, 320.0, 320.0)
    v = tsdf_amd.TSDFVolume(dims, phys)
    rng = np.random.default_rng(2)
    pts = rng.uniform(20.0, 300.0, size=(1000, 3)).astype(np.float32)   # away from the outer half voxel
    got = v.deform_mesh(pts)
    assert_same_floats(got, oracle.deform_points(dims, v.voxel_size(), (0, 0, 0), (0, 0, 0), None, (0, 0, 0), (0, 0, 0), pts), "identity")
    # the blend of a regular lattice with the swapped 6/7 coefficients is the identity only in y and z; x shifts by
    # (swap) * voxel -- what matters here is parity, and that the result stays within one voxel of the input
    assert np.max(np.abs(got - pts)) <= 20.0 + 1e-3


def test_cpp_class_and_empty_input():
    v = tsdf_amd.TSDFVolume((8, 8, 8), (80.0, 80.0, 80.0))
    assert v.deform_mesh(np.zeros((0, 3), np.float32)).shape == (0, 3)


def test_sphere_mesh_through_a_twist_field_like_the_reference_fixture(oracle):
    """The scenario of the reference's src/Tests/test_MC_main.cpp:12-152 (a sphere SDF, a 'banana' twist written into the
    deformation nodes, extract_surface, then the mesh pushed through the field), at 48^3 instead of 200^3: mesh of the
    sphere by the host marching cubes, deformed on the GPU, equal to the oracle's deformation of the same vertices."""
    n, phys = 48, 2000.0
    v = tsdf_amd.TSDFVolume((n, n, n), (phys,) * 3)
    vs = phys / n
    zz, yy, xx = np.mgrid[0:n, 0:n, 0:n]
    c = [((a + np.float32(0.5)) * np.float32(vs)).astype(np.float32) for a in (xx, yy, zz)]
    d = np.sqrt((c[0] - phys / 2) ** 2 + (c[1] - phys / 2) ** 2 + (c[2] - phys / 2) ** 2) - phys / 2.5
    v.set_distance_data(d.astype(np.float32).reshape(-1))
    # build_twist_translation_data: rotate every node about an axis at x = 1.5 * extent by twice its polar angle
    cx, cy = 1.5 * phys, 0.5 * phys
    theta = np.arctan2(c[1] - cy, c[0] - cx) * 2
    tx = (np.cos(theta) * (c[0] - cx) - np.sin(theta) * (c[1] - cy)) + cx
    ty = (np.sin(theta) * (c[0] - cx) + np.cos(theta) * (c[1] - cy)) + cy
    nodes = np.stack([tx, ty, c[2], 0 * tx, 0 * tx, 0 * tx], axis=-1).reshape(-1, 6).astype(np.float32)
    v.set_deformation(nodes)
    mesh = tsdf_amd.marching_cubes(v.get_distance_data(), (n, n, n), (vs,) * 3)
    assert mesh.shape[0] > 3000 and np.all(np.abs(np.linalg.norm(mesh - phys / 2, axis=1) - phys / 2.5) < 0.1 * vs)
    bent = v.deform_mesh(mesh)
    exp = oracle.deform_points((n, n, n), v.voxel_size(), (0, 0, 0), (0, 0, 0), nodes, (0, 0, 0), (0, 0, 0), mesh)
    assert_same_floats(bent, exp, "twisted sphere")
    assert np.max(np.linalg.norm(bent - mesh, axis=1)) > 100.0       # it really bends
