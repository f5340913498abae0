"""Generates tests/golden/ref_io.npz from the REFERENCE's own code.  Run from the repo root in the build container:

    python -m tests.golden.make_ref_io_vectors

The outputs come from oracle/_ref/libref_io.so, i.e. /root/reference/src/Utilities/ply.cpp (write_to_ply) and
src/Utilities/PgmUtilities.cpp (read_pgm) compiled where they lie (oracle/Makefile target "ref"; oracle/ref_io_wrap.cpp has the C entry
points).  Data only: the meshes and PGM files given to the reference and what it wrote / read.
"""
import os
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def meshes():
    """(name, vertices (n, 3) float32, triangles (m, 3) int32): values that exercise the stream's default formatting (6 significant
    digits, exponents, negative zero, integers, subnormals, the largest float)."""
    rng = np.random.RandomState(31)
    out = []
    v = (rng.rand(200, 3) * 3000).astype(np.float32)
    t = rng.randint(0, 200, (120, 3)).astype(np.int32)
    out.append(("volume_like", v, t))
    special = np.array([[0.0, -0.0, 1.0], [1e-7, 123456.7, 1234567.0], [-1e10, 3.4028235e38, 1e-45], [0.1, 0.5, 999999.5],
                        [999999.4, 100000.0, 1e6], [1.5e-5, -2.25, 0.000123456789]], np.float32)
    out.append(("special_values", special, np.array([[0, 1, 2], [5, 4, 3], [2147483647, -1, 0]], np.int32)))
    out.append(("empty", np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)))
    out.append(("points_only", (rng.randn(7, 3) * 10).astype(np.float32), np.zeros((0, 3), np.int32)))
    return out


def pgm_files():
    """(name, the file's bytes)"""
    rng = np.random.RandomState(32)
    out = []
    w, h = 23, 9
    samples = rng.randint(0, 65536, (h, w)).astype(">u2")
    out.append(("p5_16bit", b"P5\n%d %d\n65535\n" % (w, h) + samples.tobytes()))
    out.append(("p5_16bit_tabs_and_crlf", b"P5\r\n%d\t%d \r\n 4095\n" % (w, h) + (samples & 4095).astype(">u2").tobytes()))
    small = rng.randint(0, 256, (5, 11)).astype(np.uint8)
    out.append(("p5_8bit", b"P5\n11 5\n255\n" + small.tobytes()))
    out.append(("p5_one_pixel", b"P5 1 1 65535 " + bytes([0x12, 0x34])))
    return out


def main():
    import oracle as O
    assert O.have_ref_io(), "make -C oracle ref (needs /root/reference and the CUDA toolkit headers) first"
    data = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, v, t in meshes():
            path = os.path.join(tmp, name + ".ply")
            O.ref_write_to_ply(path, v, t)
            data["ply_" + name + "_vertices"] = v
            data["ply_" + name + "_triangles"] = t
            data["ply_" + name + "_bytes"] = np.frombuffer(open(path, "rb").read(), np.uint8)
        for name, raw in pgm_files():
            path = os.path.join(tmp, name + ".pgm")
            open(path, "wb").write(raw)
            data["pgm_" + name + "_file"] = np.frombuffer(raw, np.uint8)
            data["pgm_" + name + "_read"] = O.ref_read_pgm(path)
    out = os.path.join(HERE, "ref_io.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out), "bytes;", len(data), "arrays")


if __name__ == "__main__":
    main()
