"""tools/kinfu_stream.cpp: BASELINE configs[2] driven from C++ through tsdf_pipeline_step (no Python in the loop) on a synthetic
TUM-layout directory.  Its final volume and last picture must be the oracle's, bit for bit, and its checksum the one the Python
mirror of the same entry points gets on the same directory."""
import json
import os
import subprocess

import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, assert_same_floats
from tsdf_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "kinfu_stream")


def test_kinfu_stream_is_built_and_checks_its_arguments(tmp_path):
    """No GPU needed: the binary exists (make cpptest) and refuses a call without a directory / with a missing one."""
    assert os.path.exists(BIN), "build/kinfu_stream missing: run `make cpptest` (build() does)"
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "-d <tum dir>" in r.stderr
    r = subprocess.run([BIN, "-d", str(tmp_path / "nothing_here")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and r.stdout == ""


def test_tum_directory_read_back_through_the_host_loader(tmp_path):
    """The loader both drivers use (TUMDataLoader + DepthImage + PNG codec of the host library): depth x 5 in the PNGs comes back
    as the millimetres that were written, poses to fp32 rounding of the quaternion round trip."""
    written = synth.write_tum_directory(str(tmp_path), 3, seed=0x5EED0002)
    frames, size = tsdf_amd.load_tum_directory(str(tmp_path))
    assert size == (W, H) and len(frames) == 3
    for (d, pose), (got, cam) in zip(written, frames):
        assert np.array_equal(d, got)
        assert np.abs(pose - cam.pose()).max() < 1e-3
    with pytest.raises(ValueError):
        tsdf_amd.load_tum_directory(str(tmp_path / "missing"))


def test_a_missing_or_corrupt_frame_is_an_error_not_the_end_of_the_stream(tmp_path):
    """the C wrapper of TUMDataLoader::next told exhaustion, a missing PNG and a corrupt PNG apart by nothing; a
    truncated stream would have become a different workload without a word."""
    synth.write_tum_directory(str(tmp_path), 4, seed=0x5EED0002)
    pngs = sorted(os.listdir(tmp_path / "depth"))
    os.rename(tmp_path / "depth" / pngs[2], tmp_path / "depth" / "moved_away.png")
    with pytest.raises(ValueError, match="record 2"):
        tsdf_amd.load_tum_directory(str(tmp_path))
    os.rename(tmp_path / "depth" / "moved_away.png", tmp_path / "depth" / pngs[2])
    with open(tmp_path / "depth" / pngs[1], "wb") as f:
        f.write(b"not a png at all")
    with pytest.raises(ValueError, match="record 1"):
        tsdf_amd.load_tum_directory(str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [True, False])
def test_kinfu_stream_matches_the_oracle_and_the_python_mirror(tmp_path, oracle, overlap):
    import torch
    from tsdf_amd.pipeline import FusionPipeline
    n, F, Wu, K = 96, 5, 2, 5                       # 7 steps over 5 frames: the stream wraps round
    d = tmp_path / "tum"
    synth.write_tum_directory(str(d), F, seed=0x5EED0003, stream_frames=40)
    out = tmp_path / "out"
    out.mkdir()
    cmd = [BIN, "-d", str(d), "-n", str(n), "-k", str(K), "-w", str(Wu), "--dump", str(out)] + ([] if overlap else ["--no-overlap"])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["steps"] == K and line["frames_in_directory"] == F and line["overlap"] is overlap and line["ms_per_step"] > 0

    frames, _ = tsdf_amd.load_tum_directory(str(d))
    poses = np.fromfile(str(out / "poses.f32"), np.float32).reshape(F, 16)
    for (_, cam), p in zip(frames, poses):
        assert np.array_equal(cam.pose().view(np.uint32), p.view(np.uint32))      # the same loader, the same Camera

    # the oracle on the same steps
    ov = oracle.Volume((n, n, n), (3000.0,) * 3)
    threads = oracle.max_threads()
    for i in range(Wu + K):
        depth, cam = frames[i % F]
        f = oracle.bilateral_u16(depth, W, H, 30.0, 4.5, nthreads=threads).reshape(-1)
        ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
    last = frames[(Wu + K - 1) % F][1]
    Vo, No = ov.raycast(W, H, last.pose(), last.kinv(), nthreads=threads)
    assert_same_floats(np.fromfile(str(out / "distances.f32"), np.float32), ov.dist, "C++ driver: distances vs oracle")
    assert_same_floats(np.fromfile(str(out / "weights.f32"), np.float32), ov.weight, "C++ driver: weights vs oracle")
    V = np.fromfile(str(out / "vertices.f32"), np.float32).reshape(-1, 3)
    N = np.fromfile(str(out / "normals.f32"), np.float32).reshape(-1, 3)
    assert_same_floats(V, Vo, "C++ driver: last picture vs oracle")
    assert_same_floats(N, No, "C++ driver: last normals vs oracle")
    assert line["last_frame_hits"] == int((~np.isnan(Vo[:, 0])).sum()) > 1000
    assert line["last_frame_vertex_bits"] == int(Vo.view(np.int32).astype(np.int64).sum())
    # (the normals' NaNs come out of arithmetic: the host's have another bit pattern than the device's -- their checksum is
    # compared with the Python mirror's below, device against device)

    # the Python mirror of the same entry points on the same directory: the same bits
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    pipe = FusionPipeline(vol, tsdf_amd.BilateralFilter(30.0, 4.5), tsdf_amd.GPURaycaster(W, H), W, H, overlap=overlap)
    depth = torch.from_numpy(np.stack([f for f, _ in frames]).view(np.int16)).cuda()
    vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    norm = torch.empty_like(vert)
    for i in range(Wu + K):
        a, b = i % F, (i + 1) % F
        pipe.step(depth[a].data_ptr(), frames[a][1], vert.data_ptr(), norm.data_ptr(), depth[b].data_ptr(), frames[b][1])
    pipe.synchronize()
    assert int(vert.cpu().numpy().view(np.int32).astype(np.int64).sum()) == line["last_frame_vertex_bits"]
    assert int(norm.cpu().numpy().view(np.int32).astype(np.int64).sum()) == line["last_frame_normal_bits"]
    assert_same_floats(vol.get_distance_data(), ov.dist, "Python mirror: distances vs oracle")
    pipe.close()
    vol.close()


@pytest.mark.gpu
def test_kinfu_stream_tracks_like_the_python_mirror(tmp_path):
    """kinfu_stream --track: BASELINE configs[4]'s loop from C++ (tsdf_tracker_filter / _align / _integrate + the Camera class) on a TUM
    directory.  The same entry points through the Python mirror (tsdf_amd.tracking.FrameToModelTracker) on the same frames: the same
    volume bit for bit when the poses agree bit for bit; the poses agree to the rounding of the one 4 x 4 double product per frame that the two hosts
    form in different orders (numpy's matmul, a plain loop) -- compared to 1e-4 mm / 1e-7 -- and follow the ground truth."""
    import torch
    from tsdf_amd.tracking import FrameToModelTracker
    n, F = 128, 8
    d = tmp_path / "tum"
    synth.write_tum_directory(str(d), F, seed=0x5EED0005, stream_frames=200)
    out = tmp_path / "out"
    out.mkdir()
    r = subprocess.run([BIN, "-d", str(d), "-n", str(n), "-k", str(F), "--track", "--dump", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["frames"] == F and line["ms_per_frame"] > 0 and line["last_icp_inliers"] > 0.3 * W * H
    assert line["last_pose_translation_error_mm"] < 25.0          # voxels are 23 mm here, depth noise +-3 mm
    poses = np.fromfile(str(out / "poses.f32"), np.float32).reshape(F, 4, 4).transpose(0, 2, 1)     # column-major -> usual notation

    frames, _ = tsdf_amd.load_tum_directory(str(d))
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    tracker = FrameToModelTracker(vol, W, H)
    dev = [torch.from_numpy(f.view(np.int16).copy()).cuda() for f, _ in frames]
    torch.cuda.synchronize()
    mine = []
    for i, (_, cam) in enumerate(frames):
        truth = cam.pose().astype(np.float64).reshape(4, 4).T
        mine.append(tracker.process_device(dev[i].data_ptr(), initial_pose=truth if i == 0 else None))
    tracker.synchronize()
    mine = np.stack(mine)
    assert np.array_equal(poses[0].astype(np.float64), mine[0])                     # the first frame: the loader's pose, both sides
    assert np.abs(poses[:, :3, 3] - mine[:, :3, 3]).max() < 1e-4 * F, np.abs(poses[:, :3, 3] - mine[:, :3, 3]).max()
    assert np.abs(poses[:, :3, :3] - mine[:, :3, :3]).max() < 1e-6
    if np.array_equal(poses.astype(np.float64), mine):
        assert_same_floats(np.fromfile(str(out / "distances.f32"), np.float32), vol.get_distance_data(), "C++ tracked volume vs the Python mirror's")
    tracker.close()
    vol.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_kinfu_stream_ranks_gives_the_single_volume_picture(tmp_path, ranks):
    """kinfu_stream --ranks P (round 4): the sharded step of SURVEY.md 8e from C++ -- one forked process per Z-slab, slab integrate +
    slab ray cast + all-gather of the hit records + min-k merge through tsdf_pipeline_step -- with no Python in it.  On a box with
    fewer GPUs than ranks the ranks share GPU 0 and the records travel through host shared memory (a tsdf_exchange_fn); with P GPUs
    the same command runs ncclAllGather.  The merged picture and the slabs' distances must be the single-volume run's, bit for bit."""
    n, F, Wu, K = 96, 5, 2, 5
    d = tmp_path / "tum"
    synth.write_tum_directory(str(d), F, seed=0x5EED0003, stream_frames=40)
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    base = [BIN, "-d", str(d), "-n", str(n), "-k", str(K), "-w", str(Wu)]
    r1 = subprocess.run(base + ["--dump", str(one)], capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    rp = subprocess.run(base + ["--ranks", str(ranks), "--dump", str(many)], capture_output=True, text=True, timeout=600)
    assert rp.returncode == 0, rp.stdout + rp.stderr
    lines = [l for l in rp.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    a, b = json.loads(r1.stdout.strip().splitlines()[-1]), json.loads(lines[0])
    assert b["ranks"] == ranks and b["ranks_hold_the_same_picture"] is True and b["steps"] == K
    assert b["last_frame_vertex_bits"] == a["last_frame_vertex_bits"] and b["last_frame_normal_bits"] == a["last_frame_normal_bits"]
    assert b["last_frame_hits"] == a["last_frame_hits"] > 1000
    assert_same_floats(np.fromfile(str(many / "vertices.f32"), np.float32), np.fromfile(str(one / "vertices.f32"), np.float32), "merged picture")
    whole = np.fromfile(str(one / "distances.f32"), np.float32).reshape(n, n * n)
    for r in range(ranks):
        z0, z1 = n * r // ranks, n * (r + 1) // ranks
        assert_same_floats(np.fromfile(str(many / ("distances.rank%d.f32" % r)), np.float32), whole[z0:z1], "slab %d distances" % r)


@pytest.mark.gpu
def test_kinfu_stream_eight_ranks_over_1024_planes_validate_their_merge(tmp_path):
    """The shape of BASELINE configs[3] -- 1024 planes in eight Z-slabs -- on a grid small enough for eight processes to share one GPU
    (48 x 48 x 1024 voxels over the 3 m cube): every rank's slab integrate + slab cast, the all-gather of the 8-byte records, the min-k
    merge, and then SURVEY.md 8e's mode B as the validator (--validate-merge): every rank all-gathers the DISTANCE slabs, casts the whole
    volume the ordinary way and must find the merged picture's bits.  The single-volume run of the same command is the second witness."""
    n, planes, F, Wu, K = 48, 1024, 4, 1, 3
    d = tmp_path / "tum"
    synth.write_tum_directory(str(d), F, seed=0x5EED0004, stream_frames=40)
    base = [BIN, "-d", str(d), "-n", str(n), "--planes", str(planes), "-k", str(K), "-w", str(Wu)]
    r1 = subprocess.run(base, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    r8 = subprocess.run(base + ["--ranks", "8", "--validate-merge"], capture_output=True, text=True, timeout=900)
    assert r8.returncode == 0, r8.stdout + r8.stderr
    a = json.loads(r1.stdout.strip().splitlines()[-1])
    b = json.loads([l for l in r8.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert b["ranks"] == 8 and b["ranks_seen"] == 8 and b["slab_planes"] == 128 and b["planes"] == planes
    assert b["merge_validated_mode_b"] is True and b["mode_b_differing_words"] == 0 and b["ranks_hold_the_same_picture"] is True
    assert b["last_frame_vertex_bits"] == a["last_frame_vertex_bits"] and b["last_frame_normal_bits"] == a["last_frame_normal_bits"]
    assert b["last_frame_hits"] == a["last_frame_hits"] > 1000
