"""A long stream: all 200 frames of the config-3 surrogate (bilateral filter + integrate every frame, ray cast every 10th)
at 128^3, bit for bit against the oracle.  The occupancy flags the ray caster skips on are refreshed from the distances
after 2, 4, 8, 16 frames and then every 16 (tsdf_amd/csrc/integrate.hip, launch_integrate): a 200-frame stream crosses that
schedule a dozen times, with marks set by integrate in between, and the weights grow to 200 (uncapped, Q4)."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, assert_same_floats
from tsdf_amd import synth

pytestmark = pytest.mark.gpu


def test_200_frames_of_config3_against_the_oracle(oracle):
    n, frames, seed = 128, 200, 0x5EED0003
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    threads = oracle.max_threads()
    casts = hits = 0
    for i in range(frames):
        d, cam = synth.depth_frame(i, frames, seed=seed)
        f = d.copy()
        bil.filter(f, W, H)
        if i % 25 == 0:
            fo = oracle.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=threads).reshape(-1)
            assert np.array_equal(f, fo), "bilateral, frame %d" % i
        gv.integrate(f, W, H, cam)
        ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
        if i % 10 == 9 or i in (1, 2, 3, 4, 5, 15, 16, 17):          # around the refreshes too
            V, Nn = gv.raycast(W, H, cam)
            Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=threads)
            assert_same_floats(V, Vo, "vertices after frame %d" % i)
            assert_same_floats(Nn, No, "normals after frame %d" % i)
            casts += 1
            hits += int((~np.isnan(V[:, 0])).sum())
        if i % 50 == 49:
            assert_same_floats(gv.get_weight_data(), ov.weight, "weights after frame %d" % i)
            assert_same_floats(gv.get_distance_data(), ov.dist, "distances after frame %d" % i)
    assert casts == 28 and hits > casts * 0.5 * W * H
    assert ov.weight.max() > 150.0
