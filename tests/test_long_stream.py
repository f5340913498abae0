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


def test_200_frames_of_config3_at_512_cubed_on_the_device_path(oracle):
    """BASELINE configs[2] at its full size and length, driven the way bench.py drives it: frames resident in HBM, the filter
    leaving its tile maxima for integrate, a ray cast after EVERY frame (so the occupancy flags go through their whole schedule
    of marks and incremental rebuilds at 512^3: after 2, 4, 8, 16 frames, then every 16).  The oracle integrates the same
    filtered frames; pictures are compared after frames 18, 80, 160 and 200, the volume after frame 200."""
    import torch
    n, frames, seed = 512, 200, 0x5EED0003
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    rc = tsdf_amd.GPURaycaster(W, H)
    s = torch.cuda.current_stream()
    gv.set_stream(s.cuda_stream)
    threads = oracle.max_threads()
    filt = torch.empty((H * W,), dtype=torch.int16, device="cuda")
    tmax = torch.empty((((W + 15) // 16) * ((H + 15) // 16),), dtype=torch.int16, device="cuda")
    vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    norm = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    compared = 0
    for i in range(frames):
        d, cam = synth.depth_frame(i, frames, seed=seed)
        src = torch.from_numpy(d.view(np.int16).copy()).cuda()
        bil.filter_device(src.data_ptr(), filt.data_ptr(), W, H, bits=16, stream=s.cuda_stream, tile_max_ptr=tmax.data_ptr())
        gv.integrate_device(filt.data_ptr(), W, H, cam, tile_max_ptr=tmax.data_ptr())
        rc.raycast_device(gv, cam, vert.data_ptr(), norm.data_ptr())
        f = filt.cpu().numpy().view(np.uint16)          # (synchronises)
        ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
        if i in (17, 79, 159, 199):
            Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=threads)
            assert_same_floats(vert.cpu().numpy().reshape(-1), np.asarray(Vo).reshape(-1), "vertices after frame %d" % i)
            assert_same_floats(norm.cpu().numpy().reshape(-1), np.asarray(No).reshape(-1), "normals after frame %d" % i)
            compared += 1
        if i == frames - 1:
            assert_same_floats(gv.get_weight_data(), ov.weight, "weights after frame %d" % i)
            assert_same_floats(gv.get_distance_data(), ov.dist, "distances after frame %d" % i)
    assert compared == 4 and ov.weight.max() > 150.0
