"""tsdf_amd.pipeline.FusionPipeline: the next frame's bilateral filter on a second, lower-priority stream during this frame's
ray cast.  Scheduling only -- volume and pictures must be the bits of the strictly sequential step and of the oracle."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, assert_same_floats
from tsdf_amd import synth

pytestmark = pytest.mark.gpu


def _run(overlap, release_after_integrate, frames, n):
    import torch
    from tsdf_amd.pipeline import FusionPipeline
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    pipe = FusionPipeline(vol, tsdf_amd.BilateralFilter(30.0, 4.5), tsdf_amd.GPURaycaster(W, H), W, H, overlap=overlap,
                          release_after_integrate=release_after_integrate)
    depth = torch.from_numpy(np.stack([d for d, _ in frames]).view(np.int16)).cuda()
    vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    norm = torch.empty_like(vert)
    pictures = []
    for i, (_, cam) in enumerate(frames):
        # (the third frame is NOT announced: the pipeline must filter it itself when it arrives)
        nxt = depth[i + 1].data_ptr() if i + 1 < len(frames) and i != 1 else None
        pipe.step(depth[i].data_ptr(), cam, vert.data_ptr(), norm.data_ptr(), nxt)
        pipe.synchronize()
        pictures.append((vert.cpu().numpy().copy(), norm.cpu().numpy().copy()))
    out = (vol.get_distance_data(), vol.get_weight_data(), pictures)
    vol.close()
    return out


def test_filter_ahead_gives_the_bits_of_the_sequential_step_and_of_the_oracle(oracle):
    n = 96
    frames = [synth.depth_frame(i * 3, 200, seed=0x5EED0003) for i in range(7)]
    seq = _run(False, True, frames, n)
    for overlap, gate in ((True, True), (True, False)):
        got = _run(overlap, gate, frames, n)
        assert_same_floats(got[0], seq[0], "distances (overlap, release after integrate = %s)" % gate)
        assert_same_floats(got[1], seq[1], "weights")
        for i, ((v, nn), (vs, ns)) in enumerate(zip(got[2], seq[2])):
            assert_same_floats(v, vs, "vertices of frame %d" % i)
            assert_same_floats(nn, ns, "normals of frame %d" % i)
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    threads = oracle.max_threads()
    for d, cam in frames:
        f = oracle.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=threads).reshape(-1)
        ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
    assert_same_floats(seq[0], ov.dist, "distances vs oracle")
    assert_same_floats(seq[1], ov.weight, "weights vs oracle")
    Vo, No = ov.raycast(W, H, frames[-1][1].pose(), frames[-1][1].kinv(), nthreads=threads)
    assert_same_floats(seq[2][-1][0], Vo, "last picture vs oracle")
    assert_same_floats(seq[2][-1][1], No, "last normals vs oracle")
