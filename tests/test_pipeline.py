"""tsdf_pipeline_* (C++ behind the C ABI; tsdf_amd.pipeline.FusionPipeline is the ctypes mirror): the next frame's bilateral
filter and brick culling on a second, lower-priority stream during this frame's ray cast.  Scheduling only -- volume and
pictures must be the bits of the strictly sequential step and of the oracle."""
import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, assert_same_floats
from tsdf_amd import synth

pytestmark = pytest.mark.gpu


def _run(overlap, equal_priority, frames, n, prepare=False, unannounced=(1,), wrong_announcement=(), tighten_ahead=True, sync=True,
         flags_after=None):
    import torch
    from tsdf_amd.pipeline import FusionPipeline
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    pipe = FusionPipeline(vol, tsdf_amd.BilateralFilter(30.0, 4.5), tsdf_amd.GPURaycaster(W, H), W, H, overlap=overlap,
                          equal_priority=equal_priority, tighten_ahead=tighten_ahead)
    depth = torch.from_numpy(np.stack([d for d, _ in frames]).view(np.int16)).cuda()
    vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    norm = torch.empty_like(vert)
    pictures = []
    for i, (_, cam) in enumerate(frames):
        # (the third frame is NOT announced: the pipeline must filter it itself when it arrives; a WRONG announcement -- another
        # frame arrives than the one filtered and culled ahead -- must be waited for and dropped, round-2 advisor finding)
        j = i + 1
        if i in wrong_announcement and i + 2 < len(frames):
            j = i + 2
        nxt = depth[j].data_ptr() if j < len(frames) and i not in unannounced else None
        pipe.step(depth[i].data_ptr(), cam, vert.data_ptr(), norm.data_ptr(), nxt,
                  frames[j][1] if (prepare and nxt is not None) else None)
        if sync:
            pipe.synchronize()
            pictures.append((vert.cpu().numpy().copy(), norm.cpu().numpy().copy()))
        else:   # (no host round trip between the steps: the streams really run ahead of each other)
            with torch.cuda.stream(pipe.main):   # (the copies are ordered behind the step on ITS stream)
                pictures.append((vert.clone(), norm.clone()))
        if flags_after is not None and i in flags_after:
            pipe.synchronize()
            flags_after[i] = vol.occupancy_data(force_rebuild=False)
    pipe.synchronize()
    pictures = [(v if isinstance(v, np.ndarray) else v.cpu().numpy(), m if isinstance(m, np.ndarray) else m.cpu().numpy()) for v, m in pictures]
    out = (vol.get_distance_data(), vol.get_weight_data(), pictures)
    pipe.close()
    vol.close()
    return out


def test_filter_ahead_gives_the_bits_of_the_sequential_step_and_of_the_oracle(oracle):
    n = 96
    frames = [synth.depth_frame(i * 3, 200, seed=0x5EED0003) for i in range(7)]
    seq = _run(False, False, frames, n)
    for overlap, equal, prepare, wrong in ((True, False, False, ()), (True, True, False, ()), (True, False, True, ()), (True, False, True, (3, 4))):
        got = _run(overlap, equal, frames, n, prepare, wrong_announcement=wrong)
        assert_same_floats(got[0], seq[0], "distances (overlap, equal priority = %s, wrong announcements = %s)" % (equal, wrong))
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


def test_flags_tightened_beside_the_ray_cast_change_nothing(oracle):
    """The periodic rebuild of the ray caster's flags (after 2, 4, 8, 16 integrations) runs on the second stream beside the ray
    cast of the frame that made it due (occupancy_tighten_on); the next integrate waits for it.  Volume and every picture must be
    the bits of the run that rebuilds in front of the ray cast, with and without a host round trip between the steps, and the
    flags the next frames see must still cover the distances (a flag the rebuild cleared after integrate had set it would not)."""
    n = 96
    frames = [synth.depth_frame(i * 2, 200, seed=0x5EED0003) for i in range(19)]
    ref = _run(True, False, frames, n, prepare=True, unannounced=(), tighten_ahead=False)
    flags = {8: None, 16: None, 18: None}
    for sync in (True, False):
        got = _run(True, False, frames, n, prepare=True, unannounced=(), tighten_ahead=True, sync=sync, flags_after=flags if sync else None)
        assert_same_floats(got[0], ref[0], "distances")
        assert_same_floats(got[1], ref[1], "weights")
        for i, ((v, nn), (vs, ns)) in enumerate(zip(got[2], ref[2])):
            assert_same_floats(v, vs, "vertices of frame %d (host round trips: %s)" % (i, sync))
            assert_same_floats(nn, ns, "normals of frame %d" % i)
    # the flags after frames 8, 16 (tightened beside their ray cast) and 18 must cover the final distances' low voxels of that time;
    # checked on the last state: every brick the definition flags for the final volume is flagged after frame 18
    tau = np.float32(0.01) * np.float32(tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3).truncation_distance())
    low = ~(got[0].reshape(n, n, n) > tau)
    zz, yy, xx = np.nonzero(low)
    fine18 = flags[18][0].reshape(n // 4, n // 4, n // 4)
    for dz in (-2, 2):
        for dy in (-2, 2):
            for dx in (-2, 2):
                bz, by, bx = np.clip((zz + dz) // 4, 0, n // 4 - 1), np.clip((yy + dy) // 4, 0, n // 4 - 1), np.clip((xx + dx) // 4, 0, n // 4 - 1)
                assert fine18[bz, by, bx].all(), "a low voxel's grown brick is not flagged after the deferred tightening"
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    threads = oracle.max_threads()
    for d, cam in frames:
        f = oracle.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=threads).reshape(-1)
        ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
    assert_same_floats(ref[0], ov.dist, "distances vs oracle")
    Vo, No = ov.raycast(W, H, frames[-1][1].pose(), frames[-1][1].kinv(), nthreads=threads)
    assert_same_floats(got[2][-1][0], Vo, "last picture vs oracle")
    assert_same_floats(got[2][-1][1], No, "last normals vs oracle")


def test_a_prepared_brick_list_is_used_only_by_the_matching_integrate(oracle):
    """tsdf_integrate_prepare_device_tiles builds a frame's brick list ahead; an integrate call with other arguments must ignore
    it (and cull for itself), the matching one must use it, and a second prepare replaces the first -- the volume equals the
    oracle's after every call."""
    import torch
    n = 80
    s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    gv.set_stream(s.cuda_stream)
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    fr = [synth.depth_frame(i * 5, 200, seed=0x5EED0003) for i in range(4)]
    filt = [torch.empty((H * W,), dtype=torch.int16, device="cuda") for _ in fr]
    tmax = [torch.empty((1200,), dtype=torch.int16, device="cuda") for _ in fr]
    for (d, _), f, t in zip(fr, filt, tmax):
        src = torch.from_numpy(d.view(np.int16).copy()).cuda()
        bil.filter_device(src.data_ptr(), f.data_ptr(), W, H, bits=16, stream=s.cuda_stream, tile_max_ptr=t.data_ptr())
    torch.cuda.synchronize()

    def check(i, what):
        torch.cuda.synchronize()
        ov.integrate(filt[i].cpu().numpy().view(np.uint16), W, H, fr[i][1].inverse_pose(), fr[i][1].k(), fr[i][1].kinv(),
                     nthreads=oracle.max_threads())
        assert_same_floats(gv.get_weight_data(), ov.weight, what + ": weights")
        assert_same_floats(gv.get_distance_data(), ov.dist, what + ": distances")

    gv.integrate_device(filt[0].data_ptr(), W, H, fr[0][1], tile_max_ptr=tmax[0].data_ptr())
    check(0, "first frame, nothing prepared")
    # prepared for frame 1 on another stream, used by frame 1
    gv.integrate_prepare_device(filt[1].data_ptr(), W, H, fr[1][1], tmax[1].data_ptr(), side.cuda_stream)
    side.synchronize()
    gv.integrate_device(filt[1].data_ptr(), W, H, fr[1][1], tile_max_ptr=tmax[1].data_ptr())
    check(1, "prepared and used")
    # prepared for frame 3, but frame 2 arrives: the list must be ignored
    gv.integrate_prepare_device(filt[3].data_ptr(), W, H, fr[3][1], tmax[3].data_ptr(), side.cuda_stream)
    side.synchronize()
    gv.integrate_device(filt[2].data_ptr(), W, H, fr[2][1], tile_max_ptr=tmax[2].data_ptr())
    check(2, "prepared for another frame")
    # two prepare calls, the second one counts; and the plain entry point (no tile maxima) ignores any preparation
    gv.integrate_prepare_device(filt[0].data_ptr(), W, H, fr[0][1], tmax[0].data_ptr(), side.cuda_stream)
    gv.integrate_prepare_device(filt[3].data_ptr(), W, H, fr[3][1], tmax[3].data_ptr(), side.cuda_stream)
    side.synchronize()
    gv.integrate_device(filt[3].data_ptr(), W, H, fr[3][1], tile_max_ptr=tmax[3].data_ptr())
    check(3, "second of two preparations")
    gv.integrate_prepare_device(filt[1].data_ptr(), W, H, fr[1][1], tmax[1].data_ptr(), side.cuda_stream)
    side.synchronize()
    gv.integrate_device(filt[1].data_ptr(), W, H, fr[1][1])
    check(1, "plain integrate after a preparation")


def test_a_prepared_brick_list_does_not_outlive_the_state_it_was_culled_for(oracle):
    """The prepared list and the per-plane constants bake in the volume's offset (now and at the last clear) and the truncation
    distance: clear(), a new offset and a rewritten image (tsdf_integrate_discard_prepared) must make the matching integrate call
    cull again.  After each, the volume equals the oracle's (round-2 advisor finding)."""
    import ctypes as C
    import torch
    from tsdf_amd._capi import check as ck, lib
    n = 72
    s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    gv = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    gv.set_stream(s.cuda_stream)
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    fr = [synth.depth_frame(i * 7, 200, seed=0x5EED0003) for i in range(2)]
    src = [torch.from_numpy(d.view(np.int16).copy()).cuda() for d, _ in fr]
    filt = torch.empty((H * W,), dtype=torch.int16, device="cuda")
    tmax = torch.empty((1200,), dtype=torch.int16, device="cuda")

    def refilter(i):
        bil.filter_device(src[i].data_ptr(), filt.data_ptr(), W, H, bits=16, stream=s.cuda_stream, tile_max_ptr=tmax.data_ptr())
        torch.cuda.synchronize()

    def check(i, what):
        torch.cuda.synchronize()
        ov.integrate(filt.cpu().numpy().view(np.uint16), W, H, fr[i][1].inverse_pose(), fr[i][1].k(), fr[i][1].kinv(),
                     nthreads=oracle.max_threads())
        assert_same_floats(gv.get_weight_data(), ov.weight, what + ": weights")
        assert_same_floats(gv.get_distance_data(), ov.dist, what + ": distances")

    def prepare(i):
        gv.integrate_prepare_device(filt.data_ptr(), W, H, fr[i][1], tmax.data_ptr(), side.cuda_stream)
        side.synchronize()

    refilter(0)
    # (1) a new offset between prepare and integrate: the list was culled for the old position of the grid
    prepare(0)
    gv.offset(400.0, -250.0, 300.0)
    ov.offset(400.0, -250.0, 300.0)
    gv.integrate_device(filt.data_ptr(), W, H, fr[0][1], tile_max_ptr=tmax.data_ptr())
    check(0, "offset changed after the preparation")
    # (2) clear() between prepare and integrate: offset_at_clear changes (Q1), the plane constants with it
    prepare(0)
    gv.clear()
    ov.clear()
    gv.integrate_device(filt.data_ptr(), W, H, fr[0][1], tile_max_ptr=tmax.data_ptr())
    check(0, "cleared after the preparation")
    # (3) same pointers and pose, other content: the caller says so
    prepare(0)
    refilter(1)
    ck(lib.tsdf_integrate_discard_prepared(gv._h))
    cam0_image1 = fr[0][1]
    gv.integrate_device(filt.data_ptr(), W, H, cam0_image1, tile_max_ptr=tmax.data_ptr())
    torch.cuda.synchronize()
    ov.integrate(filt.cpu().numpy().view(np.uint16), W, H, cam0_image1.inverse_pose(), cam0_image1.k(), cam0_image1.kinv(),
                 nthreads=oracle.max_threads())
    assert_same_floats(gv.get_weight_data(), ov.weight, "image rewritten after the preparation: weights")
    assert_same_floats(gv.get_distance_data(), ov.dist, "image rewritten after the preparation: distances")


def test_a_volume_takes_one_pipeline_or_tracker_at_a_time():
    """two attachments restored each other's (destroyed) streams.  A second one is refused now; once the
    first has gone the volume is back on the stream it had, and the next attachment works."""
    from tsdf_amd.pipeline import FusionPipeline
    from tsdf_amd.tracking import FrameToModelTracker
    vol = tsdf_amd.TSDFVolume((32, 32, 32), (3000.0,) * 3)
    bil, rc = tsdf_amd.BilateralFilter(30.0, 4.5), tsdf_amd.GPURaycaster(W, H)
    before = vol.stream_ptr()
    a = FusionPipeline(vol, bil, rc, W, H)
    assert vol.stream_ptr() == a.main.cuda_stream != before
    with pytest.raises((ValueError, tsdf_amd.TsdfError), match="attached"):
        FusionPipeline(vol, bil, rc, W, H)
    with pytest.raises((ValueError, tsdf_amd.TsdfError), match="attached"):
        FrameToModelTracker(vol, W, H)
    assert vol.stream_ptr() == a.main.cuda_stream          # the refused attempts changed nothing
    a.close()
    assert vol.stream_ptr() == before
    t = FrameToModelTracker(vol, W, H)
    with pytest.raises((ValueError, tsdf_amd.TsdfError), match="attached"):
        FusionPipeline(vol, bil, rc, W, H)
    t.close()
    assert vol.stream_ptr() == before
    b = FusionPipeline(vol, bil, rc, W, H)                  # rebinding after an explicit close is fine
    d, cam = synth.depth_frame(0, 10, seed=3)
    import torch
    dd = torch.from_numpy(d.view(np.int16)).cuda()
    v = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    b.step(dd.data_ptr(), cam, v.data_ptr())
    b.synchronize()
    b.close()
    vol.close()


KNOB_SCRIPT = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
import tsdf_amd
from tsdf_amd import synth
from tsdf_amd.pipeline import FusionPipeline
W, H, n = 640, 480, 96
frames = [synth.depth_frame(i * 3, 200, seed=0x5EED0003) for i in range(20)]   # (past the 16th integration: one flag rebuild beside a ray cast)
vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
pipe = FusionPipeline(vol, tsdf_amd.BilateralFilter(30.0, 4.5), tsdf_amd.GPURaycaster(W, H), W, H, overlap=True)
depth = torch.from_numpy(np.stack([d for d, _ in frames]).view(np.int16)).cuda()
vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
norm = torch.empty_like(vert)
h = hashlib.sha256()
for i, (_, cam) in enumerate(frames):   # (no host round trip between the steps, as bench.py drives it)
    nxt = depth[i + 1].data_ptr() if i + 1 < len(frames) else None
    pipe.step(depth[i].data_ptr(), cam, vert.data_ptr(), norm.data_ptr(), nxt, frames[i + 1][1] if nxt is not None else None)
    if i %% 5 == 4:
        pipe.synchronize()
        h.update(vert.cpu().numpy().tobytes()); h.update(norm.cpu().numpy().tobytes())
pipe.synchronize()
h.update(vol.get_distance_data().tobytes()); h.update(vol.get_weight_data().tobytes())
print(h.hexdigest())
"""


def test_the_schedule_knobs_of_the_pipeline_change_no_bit():
    """Where the side stream is released, how the step waits for it and what fence its events carry are scheduling: the same volume and
    the same pictures under every setting (each in a process of its own: the knobs are read once)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for name, env in (("default", {}), ("event scope 0", {"TSDF_EVENT_SCOPE": "0"}), ("event scope 1", {"TSDF_EVENT_SCOPE": "1"}),
                      ("release 1", {"TSDF_PIPE_RELEASE": "1"}), ("release 2", {"TSDF_PIPE_RELEASE": "2"}),
                      ("host wait", {"TSDF_PIPE_HOST_WAIT": "1"}),
                      ("word release", {"TSDF_PIPE_WORD_RELEASE": "1"}), ("word release, cells forced", {"TSDF_PIPE_WORD_RELEASE": "1", "TSDF_RAY_CELLS": "2"}),
                      ("word release, march", {"TSDF_PIPE_WORD_RELEASE": "1", "TSDF_RAY_CELLS": "0"})):
        out = subprocess.run([sys.executable, "-c", KNOB_SCRIPT % root], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (name, out.stderr[-2000:])
        seen[name] = out.stdout.split()[-1]
    assert len(set(seen.values())) == 1, seen


def test_one_rank_of_a_world_through_the_loopback_exchange(oracle):
    """tsdf_slab_exchange_create_loopback: "rank r of P" on one GPU (the all-gather copies the rank's own records into every rank's
    place): what tools/dbg_slab_pipeline.py times.  With a world of one and the whole grid as the slab the merged picture IS the
    volume's picture: the sharded step (slab cast, exchange, min-k merge + normals) against the oracle, bit for bit; and a rank of a
    world of four merges four copies of its own records into the picture of its slab alone -- the same min-k picture as one copy."""
    import torch
    from tsdf_amd.multi import LoopbackExchange
    from tsdf_amd.pipeline import FusionPipeline
    n = 96
    frames = [synth.depth_frame(i, 8, seed=0x5EED0002) for i in range(4)]
    depth = torch.from_numpy(np.stack([d for d, _ in frames]).view(np.int16)).cuda()
    vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda"); norm = torch.empty_like(vert)
    pictures = {}
    for world, slab in ((1, None), (4, (24, 48)), (1, (24, 48))):
        vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3, slab=slab if slab else (0, n))
        ex = LoopbackExchange(0 if world == 1 else 1, world)
        pipe = FusionPipeline(vol, tsdf_amd.BilateralFilter(30.0, 4.5), tsdf_amd.GPURaycaster(W, H), W, H, overlap=True, exchange=ex, exchange_stream=(world == 4))
        for i, (_, cam) in enumerate(frames):
            nxt = i + 1 if i + 1 < len(frames) else None
            pipe.step(depth[i].data_ptr(), cam, vert.data_ptr(), norm.data_ptr(), depth[nxt].data_ptr() if nxt is not None else None, frames[nxt][1] if nxt is not None else None)
        pipe.synchronize()
        torch.cuda.synchronize()
        pictures[(world, slab)] = (vert.cpu().numpy().copy(), norm.cpu().numpy().copy())
        assert ex.ranks_seen() == world
        pipe.close(); ex.close(); vol.close()
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    for d, cam in frames:
        f = d.copy(); bil.filter(f, W, H)
        ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
    Vo, No = ov.raycast(W, H, frames[-1][1].pose(), frames[-1][1].kinv(), nthreads=oracle.max_threads())
    assert_same_floats(pictures[(1, None)][0], Vo, "world of one: vertices")
    assert_same_floats(pictures[(1, None)][1], No, "world of one: normals")
    assert_same_floats(pictures[(4, (24, 48))][0], pictures[(1, (24, 48))][0], "four copies of one slab's records merge like one")
    assert_same_floats(pictures[(4, (24, 48))][1], pictures[(1, (24, 48))][1], "... and so do the normals")
