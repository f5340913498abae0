"""Generates tests/golden/ref_io.npz from the REFERENCE's own code.  Run from the repo root in the build container:

    python -m tests.golden.make_ref_io_vectors

The outputs come from oracle/_ref/libref_io.so, i.e. /root/reference/src/Utilities/ply.cpp (write_to_ply) and
src/Utilities/PgmUtilities.cpp (read_pgm) and src/Utilities/FileUtilities.cpp compiled where they lie (oracle/Makefile target "ref"; oracle/ref_io_wrap.cpp has the C entry
points).  Data only: the meshes, PGM files, file names and text files given to the reference and what it wrote / read / answered.
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


def name_cases():
    """(prefix, digits, suffix, extension, candidates)"""
    c = [b"depth_00012.png", b"depth_00012Xpng", b"depth_0001a.png", b"depth_000123.png", b"depth_0012.png", b"Depth_00012.png", b"depth_00012.pnG",
         b"color_00000.png", b"sflow_00007_results01.txt", b"sflow_00007_results02.txt", b"sflow_00007.xml", b"", b".", b"7.", b"12345", b"x1y.z"]
    return [(b"depth_", 5, b"", b"png", c), (b"color_", 5, b"", b"png", c), (b"sflow_", 5, b"_results01", b"txt", c), (b"sflow_", 5, b"", b"xml", c),
            (b"", 0, b"", b"", c), (b"", 1, b"", b"", c), (b"x", 1, b"y", b"z", c), (b"", 5, b"", b"", c)]


def text_files():
    """(name, bytes): what process_file_by_lines delivers and read_last_line finds"""
    return [("two_lines", b"first\nsecond\n"), ("no_final_newline", b"first\nsecond"), ("crlf", b"first\r\nsecond\r\n"), ("trailing_empty_lines", b"a\nbb\n\n\n"),
            ("only_line", b"alone\n"), ("only_line_no_newline", b"alone"), ("empty", b""), ("newlines_only", b"\n\n"), ("blank_last", b"a\n \n"),
            ("leading_empty", b"\nx\n"), ("tum_like", b"# timestamp tx ty tz qx qy qz qw\n1305031102.175304 1.3405 0.6266 1.6575 0.6574 0.6126 -0.2949 -0.3248\n")]


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
        R = O.ref_file_utilities()
        for i, (prefix, digits, suffix, ext, cands) in enumerate(name_cases()):
            data["names_%d_template" % i] = np.array([prefix, str(digits).encode(), suffix, ext], dtype=object).astype("S")
            data["names_%d_candidates" % i] = np.array(cands, dtype="S")
            data["names_%d_matches" % i] = np.array([R.match_file_name(prefix, digits, suffix, ext, c) for c in cands])
        data["n_name_cases"] = np.int32(len(name_cases()))
        for name, raw in text_files():
            path = os.path.join(tmp, name + ".txt")
            open(path, "wb").write(raw)
            ok, lines = R.process_file_by_lines(path)
            ok_last, last = R.read_last_line(path, b"untouched")
            data["text_" + name + "_file"] = np.frombuffer(raw, np.uint8)
            data["text_" + name + "_lines_ok"] = np.bool_(ok)
            data["text_" + name + "_lines"] = np.frombuffer(b"\x1e".join(lines + [b""]), np.uint8)
            data["text_" + name + "_last_ok"] = np.bool_(ok_last)
            data["text_" + name + "_last"] = np.frombuffer(last, np.uint8)
        ok, lines = R.process_file_by_lines(os.path.join(tmp, "missing.txt"))
        data["text_missing_lines_ok"] = np.bool_(ok)
        data["text_missing_last_ok"] = np.bool_(R.read_last_line(os.path.join(tmp, "missing.txt"))[0])
    out = os.path.join(HERE, "ref_io.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out), "bytes;", len(data), "arrays")


if __name__ == "__main__":
    main()
