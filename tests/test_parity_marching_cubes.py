"""GPU: extract_surface on the device (tsdf_volume_marching_cubes) returns exactly what the oracle's restatement of the
reference's loop (oracle/mc_oracle.c: MarkAndSweepMC.cu:133-152, 219-312, 506-555 with the reference's TRIANGLE_TABLE) returns
on the downloaded distances -- the same vertices in the same order, bit for bit -- and what the host code returns."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import assert_same_floats
from tsdf_amd import synth

pytestmark = pytest.mark.gpu


def both(gv, size):
    """(device mesh, oracle mesh); the host implementation is checked on the way."""
    import oracle as O
    dev = gv.extract_surface()
    vs = gv.voxel_size()
    D = gv.get_distance_data()
    host = tsdf_amd.marching_cubes(D, size, vs, gv.offset())
    orc = O.marching_cubes(D, size, vs, gv.offset(), nthreads=O.max_threads())
    assert host.shape == orc.shape
    assert_same_floats(host, orc, "host marching cubes vs oracle %s" % (size,))
    return dev, orc


@pytest.mark.parametrize("size", [(2, 2, 2), (3, 2, 5), (17, 9, 11), (64, 64, 64), (65, 33, 70), (130, 20, 7)])
def test_random_fields_of_odd_sizes(size):
    rng = np.random.default_rng(size[0] * 1000 + size[1] * 10 + size[2])
    gv = tsdf_amd.TSDFVolume(size, (size[0] * 10.0, size[1] * 12.5, size[2] * 9.0))
    gv.offset(100.0, -50.0, 25.0)
    n = size[0] * size[1] * size[2]
    D = rng.uniform(-1.0, 1.0, n).astype(np.float32)
    D[rng.random(n) < 0.6] = 1.0                       # mostly outside, so that every configuration turns up
    D[rng.integers(0, n, 3)] = [0.0, -0.0, np.nan]     # zeros are "not negative", a NaN neither
    gv.set_distance_data(D)
    dev, host = both(gv, size)
    assert dev.shape == host.shape and dev.shape[0] % 3 == 0 and (size == (2, 2, 2) or dev.shape[0] > 0)
    assert_same_floats(dev, host, "marching cubes %s" % (size,))


def test_integrated_scene_and_empty_volume():
    n = 96
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    dev, host = both(gv, (n, n, n))
    assert dev.shape == (0, 3) and host.shape == (0, 3)          # cleared: the distance is +trunc everywhere
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    for i in range(3):
        d, cam = synth.depth_frame(i, 4, seed=0x5EED0002)
        f = d.copy(); bil.filter(f, synth.WIDTH, synth.HEIGHT)
        gv.integrate(f, synth.WIDTH, synth.HEIGHT, cam)
    dev, host = both(gv, (n, n, n))
    assert dev.shape[0] > 10000
    assert_same_floats(dev, host, "mesh of the integrated scene")


def test_slabs_concatenate_to_the_whole_volume():
    size, phys = (40, 24, 50), (400.0, 240.0, 500.0)
    rng = np.random.default_rng(7)
    D = rng.uniform(-1.0, 1.0, size[0] * size[1] * size[2]).astype(np.float32)
    D[rng.random(D.size) < 0.5] = 1.0
    whole = tsdf_amd.TSDFVolume(size, phys)
    whole.set_distance_data(D)
    ref = whole.extract_surface()
    planes = D.reshape(size[2], -1)
    for bounds in ((0, 17, 18, 49, 50), (0, 1, 25, 50), (0, 50)):
        parts = []
        for zb, ze in zip(bounds[:-1], bounds[1:]):
            s = tsdf_amd.TSDFVolume(size, phys, slab=(zb, ze))
            lo, hi = s.resident_planes()
            s.set_distance_data(planes[lo:hi].ravel())
            parts.append(s.extract_surface())
        assert_same_floats(np.concatenate(parts), ref, "slabs %s" % (bounds,))


def test_a_buffer_that_is_too_small_is_refused_and_a_bad_table_too():
    import ctypes as C
    from tsdf_amd import _capi
    gv = tsdf_amd.TSDFVolume((8, 8, 8), (80.0,) * 3)
    D = np.ones(512, np.float32); D[200] = -1.0
    gv.set_distance_data(D)
    table = tsdf_amd.marching_cubes_table()
    n = C.c_uint64(0)
    assert _capi.lib.tsdf_volume_marching_cubes(gv._h, table.ctypes.data, C.byref(n), None, 0) == 0 and n.value > 0
    small = np.zeros((n.value - 1, 3), np.float32)
    rc = _capi.lib.tsdf_volume_marching_cubes(gv._h, table.ctypes.data, C.byref(n), small.ctypes.data, n.value - 1)
    assert rc != 0 and b"do not fit" in _capi.lib.tsdf_last_error()
    bad = table.copy(); bad[1, 0] = 12                      # no such edge
    assert _capi.lib.tsdf_volume_marching_cubes(gv._h, bad.ctypes.data, C.byref(n), None, 0) != 0
    bad = table.copy(); bad[1, 3] = 0                       # four vertices: not whole triangles
    assert _capi.lib.tsdf_volume_marching_cubes(gv._h, bad.ctypes.data, C.byref(n), None, 0) != 0
