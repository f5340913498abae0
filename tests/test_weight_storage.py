"""How a volume stores its weights (tsdf_amd/csrc/weights.hip) must not show in a single bit.

The reference keeps fp32 weights and only ever adds 1 to them (src/TSDF/TSDFVolume.cu:375-377); this library keeps them as 8-bit
counts, widens to 16 bits before a count could pass 255 and to fp32 before 65535, and takes the reference's fp32 layout at once for
anything that is not such a count or when the caller asks for the device pointer.  Every accessor speaks fp32.  Here: each mode and
each transition against the CPU oracle, bit for bit (distances and weights).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import tsdf_amd
from tests.helpers import H, W, Cam, assert_same_floats, camera_at
from tsdf_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE, PHYS = (96, 80, 72), (3000.0, 2500.0, 2250.0)    # (72 planes: two whole bricks of 32 + a part; 80 rows; x past one wave)


def frames(n, seed=0x5EED0401):
    out = []
    for i in range(n):
        d, cam = synth.depth_frame(i, max(n, 8), seed=seed)
        out.append((d, cam))
    return out


def pair(oracle, size=SIZE, phys=PHYS):
    return tsdf_amd.TSDFVolume(size, phys), oracle.Volume(size, phys)


def step(oracle, gv, ov, depth, cam):
    gv.integrate(depth, W, H, cam)
    ov.integrate(depth, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())


def same(gv, ov, what):
    assert_same_floats(gv.get_weight_data(), ov.weight, what + ": weights")
    assert_same_floats(gv.get_distance_data(), ov.dist, what + ": distances")


def test_a_new_volume_counts_in_bytes_and_matches_the_oracle(oracle):
    gv, ov = pair(oracle)
    assert gv.weight_storage() == (8, False)
    for i, (d, cam) in enumerate(frames(6)):
        step(oracle, gv, ov, d, cam)
        same(gv, ov, "frame %d" % i)
    assert gv.weight_storage() == (8, False)
    assert ov.weight.max() >= 5           # (the frames do overlap: counts beyond 1 are exercised)


def test_counts_widen_before_they_could_pass_255_and_again_before_65535(oracle):
    gv, ov = pair(oracle)
    fr = frames(4)
    n = gv.resident_voxels()
    # every voxel at 253, some lower: still bytes
    w = np.full(n, 253.0, np.float32)
    w[::7] = 3.0
    gv.set_weight_data(w); ov.set_weight_data(w)
    assert gv.weight_storage() == (8, False)
    assert_same_floats(gv.get_weight_data(), w, "uploaded counts come back")
    step(oracle, gv, ov, *fr[0]); same(gv, ov, "253 -> 254")
    assert gv.weight_storage()[0] == 8
    step(oracle, gv, ov, *fr[1]); same(gv, ov, "254 -> 255")
    assert gv.weight_storage()[0] == 8
    step(oracle, gv, ov, *fr[2]); same(gv, ov, "255 -> 256")
    assert gv.weight_storage()[0] == 16
    assert ov.weight.max() == 256.0
    step(oracle, gv, ov, *fr[3]); same(gv, ov, "256 -> 257")
    # ... and 16 bits -> fp32
    w = np.full(n, 65534.0, np.float32)
    w[::5] = 300.0
    gv.set_weight_data(w); ov.set_weight_data(w)
    assert gv.weight_storage()[0] == 16
    step(oracle, gv, ov, *fr[0]); same(gv, ov, "65534 -> 65535")
    assert gv.weight_storage()[0] == 16
    step(oracle, gv, ov, *fr[1]); same(gv, ov, "65535 -> 65536")
    assert gv.weight_storage()[0] == 32
    assert ov.weight.max() == 65536.0
    step(oracle, gv, ov, *fr[2]); same(gv, ov, "65536 -> 65537")
    # clear(): back to bytes
    gv.clear(); ov.clear()
    assert gv.weight_storage() == (8, False)
    assert np.all(gv.get_weight_data() == 0)
    step(oracle, gv, ov, *fr[3]); same(gv, ov, "after clear")


def test_a_camera_that_moves_on_keeps_the_bytes_beyond_255_frames(oracle):
    """The bound that triggers widening counts integrations; before widening the library looks at the counts themselves.  Two views
    of disjoint halves of a small grid, alternating for 270 frames: no voxel is updated more than 135 times, the weights stay bytes
    and equal the oracle's."""
    size, phys = (64, 40, 36), (1920.0, 1200.0, 1080.0)
    gv, ov = pair(oracle, size, phys)
    k_cam = tsdf_amd.Camera.default_depth_camera()
    views = []
    for x in (480.0, 1440.0):             # over the left half, over the right half, looking down +z with a narrow image
        cam = camera_at((x, 600.0, -700.0))
        d = np.zeros((H, W), np.uint16)
        d[:, W // 2 - 60:W // 2 + 60] = 1200
        views.append((d.reshape(-1), cam))
    for i in range(270):
        step(oracle, gv, ov, *views[i & 1])
    assert ov.weight.max() == 135.0
    assert gv.weight_storage() == (8, False)
    same(gv, ov, "270 alternating frames")


def test_counts_widen_in_the_middle_of_a_pipeline_run(oracle):
    """A camera that does not move: every voxel in view is updated by every frame, so the 256th frame needs 16-bit counts.  The
    widening happens inside tsdf_pipeline_step (two streams, no host round trip between the steps); volume and last picture are
    the oracle's."""
    import torch
    from tsdf_amd.pipeline import FusionPipeline
    size, phys = (64, 40, 36), (1920.0, 1200.0, 1080.0)
    gv, ov = pair(oracle, size, phys)
    cam = camera_at((960.0, 600.0, -700.0))
    d = np.zeros((H, W), np.uint16)
    d[120:360, 160:480] = 1250
    d = d.reshape(-1)
    threads = oracle.max_threads()
    f = oracle.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=threads).reshape(-1)
    n_frames = 262
    pipe = FusionPipeline(gv, tsdf_amd.BilateralFilter(30.0, 4.5), tsdf_amd.GPURaycaster(W, H), W, H, overlap=True)
    depth = torch.from_numpy(d.view(np.int16)).cuda()
    depth2 = depth.clone()                                   # (a distinct buffer per in-flight frame)
    vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    norm = torch.empty_like(vert)
    bufs = (depth, depth2)
    for i in range(n_frames):
        nxt = bufs[(i + 1) & 1].data_ptr() if i + 1 < n_frames else None
        pipe.step(bufs[i & 1].data_ptr(), cam, vert.data_ptr(), norm.data_ptr(), nxt, cam if nxt else None)
        ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
    pipe.synchronize()
    assert ov.weight.max() == float(n_frames)
    assert gv.weight_storage() == (16, False)
    same(gv, ov, "262 frames from one pose through the pipeline")
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=threads)
    assert_same_floats(vert.cpu().numpy(), Vo, "last picture")
    assert_same_floats(norm.cpu().numpy(), No, "last normals")
    pipe.close()


def test_sixteen_bit_counts_walk_every_plane_position(oracle):
    """Planes 2g and 2g + 1 share a dword in the 16-bit mode, 4g .. 4g + 3 in the 8-bit one: a weight pattern that differs plane by
    plane comes back, and integrates, exactly."""
    gv, ov = pair(oracle)
    n = gv.resident_voxels()
    planes = SIZE[2]
    per = n // planes
    for top, bits in ((200.0, 8), (40000.0, 16)):
        w = (np.arange(n, dtype=np.int64) // per * 17 + np.arange(n, dtype=np.int64) % 13).astype(np.float32)
        w = np.minimum(w * (top / w.max()), top).astype(np.float32)
        w = np.floor(w)
        gv.set_weight_data(w); ov.set_weight_data(w)
        assert gv.weight_storage()[0] == bits
        assert_same_floats(gv.get_weight_data(), w, "%d-bit pattern" % bits)
        for i, (d, cam) in enumerate(frames(3, seed=0x5EED0402)):
            step(oracle, gv, ov, d, cam)
            same(gv, ov, "%d-bit pattern, frame %d" % (bits, i))


def test_weights_that_are_not_counts_keep_the_reference_layout(oracle):
    gv, ov = pair(oracle)
    n = gv.resident_voxels()
    rng = np.random.default_rng(7)
    for what, w in (("fractions", rng.random(n, dtype=np.float32) * 9.0),
                    ("a negative zero", np.where(np.arange(n) == 11, np.float32(-0.0), np.float32(2.0)).astype(np.float32)),
                    ("a NaN", np.where(np.arange(n) == 5, np.float32(np.nan), np.float32(1.0)).astype(np.float32)),
                    ("beyond 16 bits", np.full(n, 70000.0, np.float32))):
        gv.clear(); ov.clear()
        assert gv.weight_storage()[0] == 8
        gv.set_weight_data(w); ov.set_weight_data(w)
        assert gv.weight_storage() == (32, False), what
        assert_same_floats(gv.get_weight_data(), w, what + " come back")
        for d, cam in frames(2, seed=0x5EED0403):
            step(oracle, gv, ov, d, cam)
        same(gv, ov, what)
    # counts uploaded into an fp32 volume are packed again
    w = np.full(n, 4.0, np.float32)
    gv.set_weight_data(w); ov.set_weight_data(w)
    assert gv.weight_storage()[0] == 8
    step(oracle, gv, ov, *frames(1)[0]); same(gv, ov, "counts after fractions")


def test_the_device_pointer_pins_fp32(oracle):
    import torch
    gv, ov = pair(oracle)
    fr = frames(3, seed=0x5EED0404)
    step(oracle, gv, ov, *fr[0])
    p = gv.weight_data()                  # the reference's weight_data()
    assert p and gv.weight_storage() == (32, True)
    same(gv, ov, "after pinning")
    n = gv.resident_voxels()
    # what the pointer shows is what the accessor returns
    class Raw:
        __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (p, False), "version": 2}
    gv.synchronize()
    host = torch.as_tensor(Raw(), device="cuda").cpu().numpy()
    assert_same_floats(host, ov.weight, "through the pointer")
    step(oracle, gv, ov, *fr[1]); same(gv, ov, "integrate on pinned fp32")
    assert gv.weight_data() == p          # the same array: the caller may have kept the pointer
    gv.clear(); ov.clear()
    assert gv.weight_storage() == (32, True) and gv.weight_data() == p
    step(oracle, gv, ov, *fr[2]); same(gv, ov, "pinned, after clear")


def test_a_general_camera_takes_the_general_kernel_and_fp32(oracle):
    gv, ov = pair(oracle)
    fr = frames(3, seed=0x5EED0405)
    step(oracle, gv, ov, *fr[0])
    assert gv.weight_storage()[0] == 8
    cam = fr[1][1]
    k = np.array(cam.k(), np.float32).copy()
    k[3] = 0.02                           # column-major: K[0][1], a skew term -- not the standard shape
    kinv = np.linalg.inv(k.reshape(3, 3).T.astype(np.float64)).T.astype(np.float32).reshape(-1)
    skew = Cam(cam.pose(), cam.inverse_pose(), k, kinv)
    step(oracle, gv, ov, fr[1][0], skew)
    assert gv.weight_storage() == (32, False)
    same(gv, ov, "skewed intrinsics")
    step(oracle, gv, ov, *fr[2])          # a standard camera again: stays fp32 until clear()
    assert gv.weight_storage()[0] == 32
    same(gv, ov, "standard camera on fp32 weights")


def test_custom_nodes_take_fp32(oracle):
    gv, ov = pair(oracle, (40, 36, 33), (1200.0, 1080.0, 990.0))
    d, cam = synth.depth_frame(0, 8, seed=0x5EED0406)
    cam = camera_at((600, 540, -900))
    step(oracle, gv, ov, d, cam)
    assert gv.weight_storage()[0] == 8
    assert gv.deformation()               # materialises the node array (the reference's deformation())
    step(oracle, gv, ov, d, cam)
    assert gv.weight_storage()[0] == 32
    same(gv, ov, "materialised nodes")


def test_a_slab_packs_from_its_first_resident_plane(oracle):
    size, phys = (64, 48, 70), (2000.0, 1500.0, 2187.5)
    whole = oracle.Volume(size, phys)
    fr = frames(3, seed=0x5EED0407)
    for d, cam in fr:
        whole.integrate(d, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
    wd, ww = whole.dist.reshape(size[2], -1), whole.weight.reshape(size[2], -1)
    for lo, hi in ((0, 23), (23, 37), (37, 70)):
        s = tsdf_amd.TSDFVolume(size, phys, slab=(lo, hi))
        info = s.info()
        for d, cam in fr:
            s.integrate(d, W, H, cam)
        assert s.weight_storage()[0] == 8
        a, b = info.z_store_begin, info.z_store_end
        assert_same_floats(s.get_weight_data().reshape(b - a, -1), ww[a:b], "slab [%d, %d) weights" % (lo, hi))
        assert_same_floats(s.get_distance_data().reshape(b - a, -1), wd[a:b], "slab [%d, %d) distances" % (lo, hi))


def test_the_short_division_by_a_count_is_the_division():
    """integrate_packed_kernel forms (d w + tsdf) / (w + 1) with rcp + mul + two fmas (div_by_count, proof in integrate_packed.hip);
    here against the division's own instruction sequence for every mantissa, both signs, three exponents, every divisor 1 .. 65536
    (16-bit counts + 1), and the special values that take the long sequence."""
    import ctypes as C
    from tsdf_amd import _capi
    bad = C.c_uint64(123)
    assert _capi.lib.tsdf_selftest_count_division(1, 65537, C.byref(bad)) == 0
    assert bad.value == 0


SCRIPT = r"""
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import tsdf_amd
from tsdf_amd import synth
v = tsdf_amd.TSDFVolume((96, 80, 72), (3000.0, 2500.0, 2250.0))
for i in range(5):
    d, cam = synth.depth_frame(i, 8, seed=0x5EED0408)
    v.integrate(d, 640, 480, cam)
print(v.weight_storage()[0], hashlib.sha256(v.get_distance_data().tobytes()).hexdigest(), hashlib.sha256(v.get_weight_data().tobytes()).hexdigest())
"""


def test_the_starting_mode_is_a_knob_that_changes_no_bit():
    seen = {}
    for mode in ("0", "8", "16"):
        env = dict(os.environ, TSDF_WEIGHT_PACK=mode)
        out = subprocess.run([sys.executable, "-c", SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        bits, hd, hw = out.stdout.split()
        assert int(bits) == (32 if mode == "0" else int(mode))
        seen[mode] = (hd, hw)
    assert seen["0"] == seen["8"] == seen["16"]
