"""GPU parity: BilateralFilter (HIP, through the C ABI) against the reference's own outputs (committed
fixtures made from the reference build), the CPU oracle and -- when present -- the reference build itself.  Byte/integer
work: bit-exact."""
import os

import numpy as np
import pytest

import tsdf_amd
from tsdf_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_u8_equals_the_reference_fixtures():
    f = np.load(os.path.join(GOLD, "bilateral_ref_u8.npz"))
    for i in range(int(f["count"])):
        img, out = f["in_%d" % i].copy(), f["out_%d" % i]
        sc, ss = (float(x) for x in f["sigmas_%d" % i])
        h, w = img.shape
        tsdf_amd.BilateralFilter(sc, ss).filter(img, w, h)       # in place, like the reference
        assert np.array_equal(img, out), "fixture %d (%dx%d, sigmas %g/%g)" % (i, w, h, sc, ss)


@pytest.mark.parametrize("shape", [(1, 1), (5, 3), (16, 16), (17, 33), (200, 150), (640, 480)])
def test_u8_equals_oracle_and_reference_build(oracle, shape):
    w, h = shape
    rng = np.random.RandomState(w * 1000 + h)
    img = rng.randint(0, 256, (h, w)).astype(np.uint8)
    for sc, ss in ((30.0, 4.5), (3.0, 2.0), (100.0, 0.5)):
        got = img.copy()
        tsdf_amd.BilateralFilter(sc, ss).filter(got, w, h)
        assert np.array_equal(got, oracle.bilateral_u8(img, w, h, sc, ss))
        if oracle.have_ref() and w * h <= 200 * 150:
            assert np.array_equal(got, oracle.ref_bilateral_u8(img, w, h, sc, ss))


def test_u16_depth_frame_equals_oracle(oracle):
    # config 3's filter: sigma_colour 30, sigma_space 4.5 (15x15) on a 640x480 uint16 depth frame
    d, _ = synth.depth_frame(0, 10, seed=0x5EED0003)
    got = d.copy()
    tsdf_amd.BilateralFilter(30.0, 4.5).filter(got, 640, 480)
    exp = oracle.bilateral_u16(d, 640, 480, 30.0, 4.5, nthreads=oracle.max_threads())
    assert np.array_equal(got.reshape(480, 640), exp)
    assert not np.array_equal(got, d)


def test_u16_full_range_values(oracle):
    rng = np.random.RandomState(9)
    img = rng.randint(0, 65536, (37, 53)).astype(np.uint16)
    got = img.copy()
    tsdf_amd.BilateralFilter(2000.0, 3.0).filter(got, 53, 37)
    assert np.array_equal(got, oracle.bilateral_u16(img, 53, 37, 2000.0, 3.0))


@pytest.mark.parametrize("sigma_colour", [30.0, 27.0, 5.0])
def test_u16_full_range_values_through_the_15x15_kernel(oracle, sigma_colour):
    # radius 7 = the staged kernel of the pipeline's filter.  Full-range values: nearly every tap's difference lies beyond the head of
    # the similarity table held in LDS (its look-up goes past the table and is replaced by the entry from memory).  sigma_colour 30:
    # the smallest weight is 1.9e-34 and the kernel carries 4 * sum; 27 and 5: the table reaches denormals / zero, the smallest
    # weight is below 2^-120 and the plain loops run instead (tsdf_bilateral::scale_exact) -- all three must equal the oracle.
    rng = np.random.RandomState(int(sigma_colour))
    img = rng.randint(0, 65536, (48, 64)).astype(np.uint16)
    img[10:20, 10:30] = 0                       # a hole: centre pixels of value 0 whose whole sum comes from far values
    img[30:40, 5:25] = rng.randint(900, 1100, (10, 20)).astype(np.uint16)   # and a smooth patch (differences inside the head)
    got = img.copy()
    tsdf_amd.BilateralFilter(sigma_colour, 4.5).filter(got, 64, 48)
    assert np.array_equal(got, oracle.bilateral_u16(img, 64, 48, sigma_colour, 4.5))


def test_device_variant_matches_host_variant():
    import torch
    d, _ = synth.depth_frame(1, 10, seed=5)
    host = d.copy()
    f = tsdf_amd.BilateralFilter(30.0, 4.5)
    f.filter(host, 640, 480)
    src = torch.from_numpy(d.astype(np.int16)).cuda()
    dst = torch.empty_like(src)
    f.filter_device(src.data_ptr(), dst.data_ptr(), 640, 480, bits=16,
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy().view(np.uint16), host)


@pytest.mark.parametrize("shape", [(640, 480), (53, 37), (16, 16), (641, 479)])
def test_tile_maxima_of_the_filtered_image(shape):
    """tsdf_bilateral_filter_u16_device_tiles: the same filtered image, plus the largest filtered value of every 16 x 16 tile
    (what integrate's depth_tile_max_kernel computes from it)."""
    import torch
    w, h = shape
    d, _ = synth.depth_frame(2, 10, seed=7)
    img = np.ascontiguousarray(d.reshape(480, 640)[:h, :w]) if (w <= 640 and h <= 480) else None
    if img is None:
        img = np.random.RandomState(3).randint(0, 4000, (h, w)).astype(np.uint16)
    f = tsdf_amd.BilateralFilter(30.0, 4.5)
    src = torch.from_numpy(img.view(np.int16).copy()).cuda()
    plain, tiled = torch.empty_like(src), torch.empty_like(src)
    tx, ty = (w + 15) // 16, (h + 15) // 16
    tmax = torch.full((ty * tx,), -1, dtype=torch.int16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    f.filter_device(src.data_ptr(), plain.data_ptr(), w, h, bits=16, stream=s)
    f.filter_device(src.data_ptr(), tiled.data_ptr(), w, h, bits=16, stream=s, tile_max_ptr=tmax.data_ptr())
    torch.cuda.synchronize()
    out = plain.cpu().numpy().view(np.uint16).reshape(h, w)
    assert np.array_equal(tiled.cpu().numpy().view(np.uint16).reshape(h, w), out)
    exp = np.zeros((ty, tx), np.uint16)
    for j in range(ty):
        for i in range(tx):
            exp[j, i] = out[j * 16:(j + 1) * 16, i * 16:(i + 1) * 16].max()
    assert np.array_equal(tmax.cpu().numpy().view(np.uint16).reshape(ty, tx), exp)
