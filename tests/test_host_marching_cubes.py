"""Host marching cubes (tsdf_amd/host/src/MarkAndSweepMC.cpp; next row f2).  The 256-case table is built from base
configurations + the cube's rotations and must equal the reference's TRIANGLE_TABLE entry for entry: its SHA-256 is
compared with the digest tools/mc_table_sha.py took from the reference's file (tests/golden/mc_tables.sha256.json).  On top:
what any marching-cubes table must satisfy, the oracle's restatement of the reference's loop (same vertex array, bit for
bit), an independent numpy evaluation of every sign-changing lattice edge, and geometry with a known answer."""
import hashlib
import json
import os

import numpy as np
import pytest

import tsdf_amd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mc_tables.sha256.json")

CORNER = np.array([[0, 0, 1], [1, 0, 1], [1, 0, 0], [0, 0, 0], [0, 1, 1], [1, 1, 1], [1, 1, 0], [0, 1, 0]])   # MarkAndSweepMC.cu:80-97
EDGE = [(0, 1), (2, 1), (3, 2), (3, 0), (4, 5), (6, 5), (7, 6), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]        # :291-302


def rows_of(table):
    return [[int(e) for e in row if e >= 0] for row in table]


def test_table_is_the_reference_table_entry_for_entry():
    gold = json.load(open(GOLD))
    t = tsdf_amd.marching_cubes_table()
    assert t.shape == (256, 32) and t.dtype == np.int8 and (t[:, 16:] == -1).all()
    assert hashlib.sha256(np.ascontiguousarray(t[:, :16]).tobytes()).hexdigest() == gold["TRIANGLE_TABLE[256][16] int8"]
    counts = (t >= 0).sum(axis=1).astype(np.uint8)
    assert hashlib.sha256(counts.tobytes()).hexdigest() == gold["VERTICES_FOR_CUBE_TYPE[256] uint8"]
    assert hashlib.sha256(np.array(EDGE, np.uint8).tobytes()).hexdigest() == gold["EDGE_VERTICES[12][2] uint8"]


def test_table_equals_the_oracle_table(oracle):
    t, counts = oracle.mc_tables()
    assert np.array_equal(tsdf_amd.marching_cubes_table()[:, :16], t)
    assert np.array_equal((t >= 0).sum(axis=1), counts)


@pytest.mark.parametrize("size", [(2, 2, 2), (3, 2, 5), (17, 9, 11), (40, 33, 21)])
def test_vertex_array_equals_the_oracle_restatement_of_the_reference_loop(oracle, size):
    rng = np.random.default_rng(size[0] * 100 + size[2])
    n = size[0] * size[1] * size[2]
    D = rng.uniform(-1.0, 1.0, n).astype(np.float32)
    D[rng.random(n) < 0.5] = 1.0
    D[rng.integers(0, n, 3)] = [0.0, -0.0, np.nan]         # zeros and NaN are "not negative"
    vs, off = (2.0, 3.0, 1.5), (10.0, -4.0, 0.25)
    got = tsdf_amd.marching_cubes(D, size, vs, off)
    exp = oracle.marching_cubes(D, size, vs, off, nthreads=2)
    assert got.shape == exp.shape
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))


def test_table_is_a_marching_cubes_table():
    rows = rows_of(tsdf_amd.marching_cubes_table())
    assert rows[0] == [] and rows[255] == []
    assert rows[1] == [0, 8, 3]                    # only corner 0 negative: the classic orientation
    for c, row in enumerate(rows):
        assert len(row) % 3 == 0 and len(row) <= 15
        crossing = {e for e, (a, b) in enumerate(EDGE) if ((c >> a) & 1) != ((c >> b) & 1)}
        assert set(row) == crossing, c             # exactly the sign-changing edges are used
        assert set(rows[255 - c]) == crossing      # the complement cuts the same edges
        # every configuration's patch is closed within the cube's faces: each directed triangle edge that is not on a
        # cube face appears once in each direction
        seen = {}
        for i in range(0, len(row), 3):
            t = row[i:i + 3]
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                seen[(a, b)] = seen.get((a, b), 0) + 1
        for (a, b), n in seen.items():
            assert n == 1
            on_face = any(set(EDGE[a]) | set(EDGE[b]) <= set(f) for f in
                          ([0, 1, 2, 3], [4, 5, 6, 7], [0, 1, 5, 4], [3, 2, 6, 7], [0, 3, 7, 4], [1, 2, 6, 5]))
            assert on_face or (b, a) in seen, (c, a, b)


def reference_interpolate(v0, v1, w0, w1):
    """interpolate() of MarkAndSweepMC.cu:47-63 in fp32, vectorised."""
    swap = (w0 > 0) & (w1 < 0)
    v0, v1 = np.where(swap[:, None], v1, v0), np.where(swap[:, None], v0, v1)
    w0, w1 = np.where(swap, w1, w0), np.where(swap, w0, w1)
    delta = (v1 - v0).astype(np.float32)
    ratio = (-(w0) / (w1 - w0)).astype(np.float32)
    return ((ratio[:, None] * delta).astype(np.float32) + v0).astype(np.float32)


def lattice_crossings(D, vs, off):
    """Every lattice edge whose end points differ in (d < 0), interpolated as the cube code does (from the cube's point of
    view the end points come in the order of EDGE; both orders give the same point after the swap rule unless a value is
    exactly 0 -- avoided by the callers)."""
    Z, Y, X = D.shape
    pts = []
    zz, yy, xx = np.mgrid[0:Z, 0:Y, 0:X]
    C = np.stack([(xx.astype(np.int32) + np.float32(0.5)) * np.float32(vs[0]) + np.float32(off[0]),
                  (yy.astype(np.int32) + np.float32(0.5)) * np.float32(vs[1]) + np.float32(off[1]),
                  (zz.astype(np.int32) + np.float32(0.5)) * np.float32(vs[2]) + np.float32(off[2])], axis=-1).astype(np.float32)
    for axis in range(3):
        a = [slice(None)] * 3
        b = [slice(None)] * 3
        a[axis] = slice(0, -1)
        b[axis] = slice(1, None)
        a, b = tuple(a), tuple(b)
        m = (D[a] < 0) != (D[b] < 0)
        # an edge only appears in the mesh if it belongs to at least one cube (all lattice edges do when X, Y, Z >= 2)
        pts.append(reference_interpolate(C[a][m], C[b][m], D[a][m], D[b][m]))
    return np.concatenate(pts)


def as_keys(P):
    return {p.tobytes() for p in np.ascontiguousarray(P, np.float32)}


@pytest.mark.parametrize("seed", range(6))
def test_vertices_are_the_reference_interpolation_of_every_sign_change_and_the_mesh_is_closed(seed):
    rng = np.random.default_rng(100 + seed)
    X, Y, Z = (int(v) for v in rng.integers(4, 14, size=3))
    D = rng.uniform(-1, 1, size=(Z, Y, X)).astype(np.float32)
    D[D == 0] = 0.5
    # positive shell: the surface cannot leave the grid, so the mesh must be closed
    D[0], D[-1], D[:, 0], D[:, -1], D[:, :, 0], D[:, :, -1] = (1.0,) * 6
    vs, off = (2.0, 3.0, 1.5), (10.0, -4.0, 0.25)
    V = tsdf_amd.marching_cubes(D.reshape(-1), (X, Y, Z), vs, off)
    assert V.shape[0] % 3 == 0 and V.shape[0] > 0
    assert as_keys(V) == as_keys(lattice_crossings(D, vs, off))
    # closed and consistently oriented: every directed edge is matched by its reverse (a fan diagonal may lie in a cube
    # face that has four crossings and then be shared by four triangles, so 'exactly once' would be too strict)
    T = V.reshape(-1, 3, 3)
    edges = {}
    for t in T:
        for i in range(3):
            k = (t[i].tobytes(), t[(i + 1) % 3].tobytes())
            edges[k] = edges.get(k, 0) + 1
    assert all(edges.get((b, a), 0) == n for (a, b), n in edges.items())
    assert max(edges.values()) <= 2


def test_sphere_area_volume_and_orientation():
    n, r = 48, 15.0
    vs = 1.0
    zz, yy, xx = np.mgrid[0:n, 0:n, 0:n]
    c = n / 2.0
    D = (np.sqrt((xx + 0.5 - c) ** 2 + (yy + 0.5 - c) ** 2 + (zz + 0.5 - c) ** 2) - r).astype(np.float32)   # negative inside
    V = tsdf_amd.marching_cubes(D.reshape(-1), (n, n, n), (vs,) * 3).astype(np.float64)
    T = V.reshape(-1, 3, 3)
    assert np.all(np.abs(np.linalg.norm(V - c, axis=1) - r) < 0.05)           # vertices on the sphere
    # the reference wires vertices (i, i+2, i+1) (MarkAndSweepMC.cu:549)
    a, b, cc = T[:, 0], T[:, 2], T[:, 1]
    area = 0.5 * np.linalg.norm(np.cross(b - a, cc - a), axis=1).sum()
    volume = np.einsum("ij,ij->i", a - c, np.cross(b - c, cc - c)).sum() / 6.0
    assert abs(area - 4 * np.pi * r * r) < 0.01 * 4 * np.pi * r * r
    assert abs(abs(volume) - 4.0 / 3.0 * np.pi * r ** 3) < 0.01 * 4.0 / 3.0 * np.pi * r ** 3
    assert volume > 0                                                          # normals point out of the negative region


def test_empty_and_degenerate_inputs():
    assert tsdf_amd.marching_cubes(np.ones(27, np.float32), (3, 3, 3), (1, 1, 1)).shape == (0, 3)
    assert tsdf_amd.marching_cubes(-np.ones(27, np.float32), (3, 3, 3), (1, 1, 1)).shape == (0, 3)
    assert tsdf_amd.marching_cubes(np.array([1, -1], np.float32), (2, 1, 1), (1, 1, 1)).shape == (0, 3)   # no cube at all
    with pytest.raises(ValueError):
        tsdf_amd.marching_cubes(np.ones(5, np.float32), (2, 2, 2), (1, 1, 1))


def test_large_grids_are_marched_by_threads_in_the_same_order():
    """Above 2^22 cubes the z range is cut over host threads; the concatenation must be the sequential cube order: compare
    with two half volumes marched on their own (unit voxels, so the shifted offsets are exact)."""
    n, h = 176, 88
    rng = np.random.default_rng(5)
    zz, yy, xx = np.mgrid[0:n, 0:n, 0:n]
    D = (np.sqrt((xx - 80.0) ** 2 + (yy - 90.0) ** 2 + (zz - 88.0) ** 2) - 60.0 + rng.uniform(-0.3, 0.3, (n, n, n))).astype(np.float32)
    whole = tsdf_amd.marching_cubes(D.reshape(-1), (n, n, n), (1, 1, 1))
    lower = tsdf_amd.marching_cubes(D[:h + 1].reshape(-1), (n, n, h + 1), (1, 1, 1))
    upper = tsdf_amd.marching_cubes(D[h:].reshape(-1), (n, n, n - h), (1, 1, 1), offset=(0, 0, float(h)))
    assert whole.shape[0] > 100000
    assert np.array_equal(whole, np.concatenate([lower, upper]))
