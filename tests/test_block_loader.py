"""BlockTSDFLoader (text TSDF import; reference src/TSDF/BlockTSDFLoader.cpp:24-100, next row f3): the column-major text
layout -- per (y, x) column one line of size_z distances and one of size_z weights -- lands at index x + y*X + z*X*Y."""
import numpy as np

import tsdf_amd


def write_block_file(path, D, Wt, phys, trailing=""):
    Z, Y, X = D.shape
    with open(path, "w") as f:
        f.write("# a comment\n\nvoxel size=%d %d %d\n" % (X, Y, Z))
        f.write("physical size=%g %g %g\n" % phys)
        for y in range(Y):
            for x in range(X):
                f.write(" ".join(repr(float(v)) for v in D[:, y, x]) + "\n")
                f.write("# comments may appear anywhere\n" if (x + y) % 5 == 0 else "")
                f.write(" ".join(repr(float(v)) for v in Wt[:, y, x]) + "\n")
        f.write(trailing)


def test_round_trip_of_a_small_volume(tmp_path):
    rng = np.random.default_rng(3)
    X, Y, Z = 5, 4, 3
    D = rng.uniform(-30, 30, size=(Z, Y, X)).astype(np.float32)
    Wt = rng.integers(0, 9, size=(Z, Y, X)).astype(np.float32)
    p = tmp_path / "block.txt"
    write_block_file(p, D, Wt, (300.0, 240.5, 90.0))
    ok, size, phys, d, w = tsdf_amd.load_block_tsdf(p)
    assert ok and size == (X, Y, Z) and phys == (300.0, 240.5, 90.0)
    assert np.array_equal(d.reshape(Z, Y, X), D) and np.array_equal(w.reshape(Z, Y, X), Wt)


def test_incomplete_and_overlong_files_are_reported(tmp_path):
    D = np.ones((2, 2, 2), np.float32)
    p = tmp_path / "extra.txt"
    write_block_file(p, D, D, (1.0, 1.0, 1.0), trailing="1 2\n")
    assert tsdf_amd.load_block_tsdf(p)[0] is False          # data after the last column (the reference: state != done)
    q = tmp_path / "short.txt"
    q.write_text("size=2 2 2\nphysical=1 1 1\n1 1\n")
    ok, size, _, _, _ = tsdf_amd.load_block_tsdf(q)
    assert ok is False and size == (2, 2, 2)
    assert tsdf_amd.load_block_tsdf(tmp_path / "missing.txt")[0] is False
