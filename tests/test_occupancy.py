"""The brick flags the ray caster skips on (tsdf_amd/csrc/volume.hip: occupancy_rebuild, reach_mip_kernel) against
a direct numpy evaluation of their definition.  They are an internal acceleration structure -- every parity test of
the ray caster depends on them being conservative -- so they are pinned separately here: exact equality after a
rebuild, and 'never less than exact' for the marks integrate leaves between rebuilds."""
import numpy as np
import pytest

import tsdf_amd
from tsdf_amd import synth

pytestmark = pytest.mark.gpu


def _flat_band(trunc):
    """[flat_lo, flat_hi] of tsdf_amd/csrc/common.hpp (OccGrid): the band every voxel in reach of a brick at the grid boundary
    must lie in for the brick to be clear (there the reference extrapolates, Q10: positive taps alone prove nothing)."""
    t = np.float32(trunc)
    return np.float32(0.9375) * t, (np.float32(1.0) + np.float32(1.0) / np.float32(1024.0)) * t


def _expected(D, dims, tau, planes=None, trunc=None):
    X, Y, Z = dims
    low = ~(D.reshape(Z, Y, X) > tau)                       # also NaN
    flat_lo, flat_hi = _flat_band(trunc if trunc is not None else np.float32(tau) / np.float32(0.01))
    not_flat = ~((D.reshape(Z, Y, X) >= flat_lo) & (D.reshape(Z, Y, X) <= flat_hi))   # also NaN
    if planes is not None:                                   # slab: only resident planes are scanned
        keep = np.zeros(Z, bool)
        keep[planes[0]:planes[1]] = True
        low &= keep[:, None, None]
        not_flat &= keep[:, None, None]
    nbx, nby, nbz = (X + 3) // 4, (Y + 3) // 4, (Z + 3) // 4
    fine = np.zeros((nbz, nby, nbx), np.uint8)
    cell = np.zeros((nbz, nby, nbx), np.uint8)
    for bz in range(nbz):
        for by in range(nby):
            for bx in range(nbx):
                boundary = bx == 0 or by == 0 or bz == 0 or bx == nbx - 1 or by == nby - 1 or bz == nbz - 1
                g = (not_flat if boundary else low)[max(4 * bz - 2, 0):4 * bz + 6, max(4 * by - 2, 0):4 * by + 6, max(4 * bx - 2, 0):4 * bx + 6]
                fine[bz, by, bx] = 1 if g.any() else 0
                c = low[4 * bz:4 * bz + 5, 4 * by:4 * by + 5, 4 * bx:4 * bx + 5]
                partial = 4 * bx + 4 > X - 1 or 4 * by + 4 > Y - 1 or 4 * bz + 4 > Z - 1
                cell[bz, by, bx] = 1 if (partial or c.any()) else 0
    return fine, cell


def _expected_reach(fine):
    nbz, nby, nbx = fine.shape
    reach = np.zeros_like(fine)
    for bz in range(nbz):
        for by in range(nby):
            for bx in range(nbx):
                level = 0
                for l in range(1, 6):
                    s = 1 << (l - 1)
                    z0, y0, x0 = (bz // s) * s, (by // s) * s, (bx // s) * s
                    if z0 + s > nbz or y0 + s > nby or x0 + s > nbx:
                        break                                 # blocks sticking out of the grid are never empty
                    if fine[z0:z0 + s, y0:y0 + s, x0:x0 + s].any():
                        break
                    level = l
                reach[bz, by, bx] = level
    return reach


# (brick counts that are multiples of 16 on every axis take reach_mip_wave_kernel, the others reach_mip_kernel)
@pytest.mark.parametrize("dims", [(48, 40, 44), (37, 30, 41), (72, 72, 72), (128, 64, 64), (64, 128, 192)])
def test_rebuilt_flags_equal_their_definition(dims):
    X, Y, Z = dims
    rng = np.random.default_rng(X * 1000 + Y)
    v = tsdf_amd.TSDFVolume(dims, (X * 10.0, Y * 10.0, Z * 10.0))
    trunc = v.truncation_distance()
    tau = np.float32(0.01) * np.float32(trunc)
    D = np.full(X * Y * Z, trunc, np.float32)
    # a sprinkle of isolated low voxels, values straddling tau, one NaN, and a small solid block
    idx = rng.choice(D.size, size=max(6, D.size // 4000), replace=False)
    D[idx] = rng.choice(np.array([-trunc, 0.0, tau, np.nextafter(tau, np.float32(1e9)), 0.5 * tau], np.float32), idx.size)
    D[idx[0]] = np.nan
    Dv = D.reshape(Z, Y, X)
    # values round the flat band, in reach of bricks at the grid boundary and elsewhere
    flat_lo, flat_hi = _flat_band(trunc)
    band = np.array([flat_lo, np.nextafter(flat_lo, np.float32(0)), flat_hi, np.nextafter(flat_hi, np.float32(1e9)), 0.5 * trunc, 2.0 * trunc], np.float32)
    for i, val in enumerate(band):
        Dv[(5 * i + 1) % Z, (3 * i) % 6, (7 * i + 2) % X] = val          # near the y = 0 face
        Dv[Z - 1 - (i % 6), (4 * i + 9) % Y, (5 * i + 3) % X] = val      # near the far z face
        Dv[(Z // 3 + i) % Z, (Y // 3 + 2 * i) % Y, X - 1 - (i % 6)] = val  # near the far x face
        Dv[Z // 2 + 5, Y // 2 + 7 + i, X // 2 + 6] = val                 # interior: only values <= tau matter
    Dv[Z // 2:Z // 2 + 3, Y // 2:Y // 2 + 5, X // 2:X // 2 + 4] = -1.0
    v.set_distance_data(D)
    fine, cell, reach = v.occupancy_data(force_rebuild=True)
    ef, ec = _expected(D, dims, tau, trunc=trunc)
    assert np.array_equal(fine, ef)
    assert np.array_equal(cell, ec)
    assert np.array_equal(reach, _expected_reach(ef))


def test_marks_left_by_integrate_cover_the_exact_flags_and_rebuild_tightens_them():
    n = 96
    v = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    tau = np.float32(0.01) * np.float32(v.truncation_distance())
    for i in range(3):
        d, cam = synth.depth_frame(i * 5, 200, seed=0x5EED0003)
        v.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
    sticky_fine, sticky_cell, _ = v.occupancy_data()
    D = v.get_distance_data()
    ef, ec = _expected(D, (n, n, n), tau, trunc=v.truncation_distance())
    assert np.all(sticky_fine >= ef) and np.all(sticky_cell >= ec)
    fine, cell, _ = v.occupancy_data(force_rebuild=True)
    assert np.array_equal(fine, ef) and np.array_equal(cell, ec)
    assert fine.sum() <= sticky_fine.sum()


@pytest.mark.parametrize("slab", [None, (32, 80), (30, 71)])
def test_second_rebuild_reads_only_what_integrate_touched_and_still_equals_the_definition(slab):
    """After the first rebuild a rebuild scans only the integrate bricks marked since (volume.hip: touched); the flags must
    equal the definition on the current distances all the same.  Slab (30, 71) does not start on a brick boundary: full scan."""
    n = 96
    v = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3, slab=slab)
    tau = np.float32(0.01) * np.float32(v.truncation_distance())
    lo, hi = v.resident_planes()

    def definition():
        D = np.full((n, n, n), v.truncation_distance(), np.float32)     # (planes outside the slab: never low)
        D[lo:hi] = v.get_distance_data().reshape(hi - lo, n, n)
        return _expected(D.reshape(-1), (n, n, n), tau, trunc=v.truncation_distance())

    def resident(a):
        # bricks wholly inside the resident planes (the others depend on planes this volume does not hold)
        return a[(lo + 3) // 4 + 1:hi // 4 - 1]

    for rnd, first in enumerate((0, 60, 120)):
        for i in range(3):
            d, cam = synth.depth_frame(first + i * 7, 200, seed=0x5EED0003)
            v.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
        fine, cell, _ = v.occupancy_data(force_rebuild=True)
        ef, ec = definition()
        assert np.array_equal(resident(fine), resident(ef)), "fine flags, rebuild %d" % rnd
        assert np.array_equal(resident(cell), resident(ec)), "cell flags, rebuild %d" % rnd


@pytest.mark.parametrize("seed", [3, 4])
def test_incremental_rebuild_equals_a_full_scan(seed):
    """The incremental rebuild skips every brick whose own `fine` flag is clear and trusts its summary bits (volume.hip,
    occupancy_scan_kernel; the invariant is stated beside tsdf_volume::occ_dirty).  Pinned on streams that put surface into the rim zone
    -- walls that leave the grid through its faces (a wall right behind the entry face, the camera inside the volume) -- over several
    rounds of integrate + rebuild: the flags after each incremental rebuild equal the definition evaluated on ALL distances, i.e. what a
    full scan gives (definition == full scan: test_rebuilt_flags_equal_their_definition)."""
    n = 64
    v = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    tau = np.float32(0.01) * np.float32(v.truncation_distance())
    rng = np.random.RandomState(seed)
    for rnd in range(5):
        for i in range(1 + rnd % 3):
            if rng.rand() < 0.5:
                d, cam = synth.depth_frame(int(rng.randint(0, 100)), 100, seed=0x5EED0004, inside=True)
            else:
                # a slanted wall that crosses the grid's faces: depth grows across the image, near enough to start inside the first voxels
                cam = tsdf_amd.Camera.default_depth_camera()
                cam.move_to(float(rng.uniform(200, 2800)), float(rng.uniform(200, 2800)), -60.0 - 40.0 * rnd)
                cam.look_at(1500.0 + float(rng.uniform(-800, 800)), 1500.0, 1500.0)
                xs = np.arange(synth.WIDTH, dtype=np.float32)[None, :].repeat(synth.HEIGHT, 0)
                d = (80.0 + 20.0 * rnd + xs * float(rng.uniform(0.2, 4.0))).astype(np.uint16).reshape(-1)
            v.integrate(d, synth.WIDTH, synth.HEIGHT, cam)
        fine, cell, _ = v.occupancy_data(force_rebuild=True)     # (occ_dirty only: the incremental path from the second round on)
        ef, ec = _expected(v.get_distance_data(), (n, n, n), tau, trunc=v.truncation_distance())
        assert np.array_equal(fine, ef), "fine flags, round %d" % rnd
        assert np.array_equal(cell, ec), "cell flags, round %d" % rnd
        assert fine[0].any() or fine[:, 0].any() or fine[:, :, 0].any() or rnd < 1, "no surface in the rim zone: the test would prove nothing"
