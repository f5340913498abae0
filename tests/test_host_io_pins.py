"""The host library's mesh writer and NYU depth-map reader against the REFERENCE's own compiled code:
write_to_ply (reference src/Utilities/ply.cpp:6-30) byte for byte, read_nyu_depth_map (src/Utilities/DepthMapUtilities.cpp:20-33) =
the reference's read_pgm (src/Utilities/PgmUtilities.cpp:49-85) followed by its byte swap.  tests/golden/ref_io.npz holds what
the reference's build wrote / read (tests/golden/make_ref_io_vectors.py); where oracle/_ref/libref_io.so is present the same
comparison runs live on random inputs."""
import ctypes as C
import os

import numpy as np
import pytest

from tsdf_amd import _capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PLY_CASES = ["volume_like", "special_values", "empty", "points_only"]
PGM_CASES = ["p5_16bit", "p5_16bit_tabs_and_crlf", "p5_8bit", "p5_one_pixel"]


def host_write_ply(path, vertices, triangles):
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
    _capi.host.tsdf_host_write_ply(str(path).encode(), v.ctypes.data, len(v), t.ctypes.data, len(t))
    return open(path, "rb").read()


def host_read_nyu(path):
    size = (C.c_uint * 2)(0, 0)
    n = _capi.host.tsdf_host_read_nyu_depth_map(str(path).encode(), size, None, 0)
    if n == 0:
        return None
    out = np.empty(n, np.uint16)
    assert _capi.host.tsdf_host_read_nyu_depth_map(str(path).encode(), size, out.ctypes.data, n) == n
    return out.reshape(size[1], size[0])


def nyu_swap(samples):
    """src/Utilities/DepthMapUtilities.cpp:24-30: v = (v >> 8) + ((v & 0xFF) * 256)"""
    s = np.asarray(samples, np.uint16)
    return ((s >> 8) + ((s & 0xFF) << 8)).astype(np.uint16)


@pytest.mark.parametrize("name", PLY_CASES)
def test_write_to_ply_equals_the_bytes_the_reference_build_wrote(tmp_path, name):
    f = np.load(os.path.join(GOLD, "ref_io.npz"))
    got = host_write_ply(tmp_path / "mesh.ply", f["ply_%s_vertices" % name], f["ply_%s_triangles" % name])
    assert got == f["ply_%s_bytes" % name].tobytes()


@pytest.mark.parametrize("name", PGM_CASES)
def test_read_nyu_depth_map_equals_the_reference_s_read_pgm_and_swap(tmp_path, name):
    f = np.load(os.path.join(GOLD, "ref_io.npz"))
    p = tmp_path / "depth.pgm"
    p.write_bytes(f["pgm_%s_file" % name].tobytes())
    got = host_read_nyu(p)
    want = nyu_swap(f["pgm_%s_read" % name])
    assert got is not None and got.shape == want.shape and np.array_equal(got, want)


def test_a_file_that_is_not_a_pgm_is_refused(tmp_path):
    # (the reference asserts on the magic: src/Utilities/PgmUtilities.cpp:56-59; here a null map, as for a file that does not open)
    p = tmp_path / "not.pgm"
    p.write_bytes(b"P6\n2 2\n255\n" + bytes(12))
    assert host_read_nyu(p) is None
    assert host_read_nyu(tmp_path / "missing.pgm") is None


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_live_against_the_reference_build(tmp_path, oracle, seed):
    if not oracle.have_ref_io():
        pytest.skip("oracle/_ref/libref_io.so not present (built where /root/reference is mounted)")
    rng = np.random.RandomState(seed)
    n, m = int(rng.randint(1, 400)), int(rng.randint(0, 300))
    v = (rng.randn(n, 3) * 10.0 ** rng.randint(-8, 9, (n, 1))).astype(np.float32)
    t = rng.randint(0, n, (m, 3)).astype(np.int32)
    oracle.ref_write_to_ply(tmp_path / "ref.ply", v, t)
    assert host_write_ply(tmp_path / "host.ply", v, t) == open(tmp_path / "ref.ply", "rb").read()
    w, h = int(rng.randint(1, 80)), int(rng.randint(1, 60))
    maxval = [65535, 1000, 255][seed % 3]
    samples = rng.randint(0, maxval + 1, (h, w))
    raw = b"P5\n%d %d\n%d\n" % (w, h, maxval) + (samples.astype(">u2") if maxval > 255 else samples.astype(np.uint8)).tobytes()
    p = tmp_path / "depth.pgm"
    p.write_bytes(raw)
    assert np.array_equal(host_read_nyu(p), nyu_swap(oracle.ref_read_pgm(p)))


# ---- the loaders' file helpers (reference src/Utilities/FileUtilities.cpp) ----------------------------------------------------------
TEXT_CASES = ["two_lines", "no_final_newline", "crlf", "trailing_empty_lines", "only_line", "only_line_no_newline", "empty", "newlines_only",
              "blank_last", "leading_empty", "tum_like"]


def host_file_utilities(oracle):
    return oracle.FileUtilities(_capi.host, "tsdf_host")    # (the marshalling only: every call lands in libtsdf_host.so)


def test_match_file_name_answers_as_the_reference_build_did(oracle):
    f = np.load(os.path.join(GOLD, "ref_io.npz"))
    H = host_file_utilities(oracle)
    assert int(f["n_name_cases"]) >= 8
    for i in range(int(f["n_name_cases"])):
        prefix, digits, suffix, ext = [bytes(x) for x in f["names_%d_template" % i]]
        got = [H.match_file_name(prefix, int(digits), suffix, ext, bytes(c)) for c in f["names_%d_candidates" % i]]
        assert got == list(f["names_%d_matches" % i]), (prefix, digits, suffix, ext)
    # the character in front of the extension is counted, not compared (src/Utilities/FileUtilities.cpp:43-50)
    assert H.match_file_name(b"depth_", 5, b"", b"png", b"depth_00012Xpng")


@pytest.mark.parametrize("name", TEXT_CASES)
def test_lines_and_last_line_as_the_reference_build_read_them(tmp_path, oracle, name):
    f = np.load(os.path.join(GOLD, "ref_io.npz"))
    H = host_file_utilities(oracle)
    p = tmp_path / "file.txt"
    p.write_bytes(f["text_%s_file" % name].tobytes())
    ok, lines = H.process_file_by_lines(p)
    assert ok == bool(f["text_%s_lines_ok" % name]) and b"\x1e".join(lines + [b""]) == f["text_%s_lines" % name].tobytes()
    ok, last = H.read_last_line(p, b"untouched")
    assert ok == bool(f["text_%s_last_ok" % name]) and last == f["text_%s_last" % name].tobytes()


def test_a_missing_file_as_the_reference_build_reported_it(tmp_path, oracle, capfd):
    f = np.load(os.path.join(GOLD, "ref_io.npz"))
    H = host_file_utilities(oracle)
    ok, lines = H.process_file_by_lines(tmp_path / "missing.txt")
    assert ok == bool(f["text_missing_lines_ok"]) and lines == []     # (true: src/Utilities/FileUtilities.cpp:85-110 only prints)
    assert H.read_last_line(tmp_path / "missing.txt")[0] == bool(f["text_missing_last_ok"])
    capfd.readouterr()


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_file_helpers_live_against_the_reference_build(tmp_path, oracle, seed, capfd):
    if not oracle.have_ref_io():
        pytest.skip("oracle/_ref/libref_io.so not present (built where /root/reference is mounted)")
    H, R = host_file_utilities(oracle), oracle.ref_file_utilities()
    rng = np.random.RandomState(seed)
    alphabet = [b"a", b"b", b"0", b"1", b"7", b".", b"_", b"x"]
    word = lambda lo, hi: b"".join(alphabet[i] for i in rng.randint(0, len(alphabet), rng.randint(lo, hi)))
    for _ in range(3000):
        prefix, suffix, ext, digits = word(0, 3), word(0, 3), word(0, 3), int(rng.randint(0, 4))
        cand = prefix + word(0, 4) + suffix + word(0, 2) + ext if rng.rand() < 0.7 else word(0, 9)
        assert H.match_file_name(prefix, digits, suffix, ext, cand) == R.match_file_name(prefix, digits, suffix, ext, cand), (prefix, digits, suffix, ext, cand)
    pieces = [b"", b"", b"a", b"line two", b" ", b"\r", b"x\r", b"# comment", b"1 2 3"]
    for i in range(200):
        body = b"\n".join(pieces[j] for j in rng.randint(0, len(pieces), rng.randint(0, 7)))
        if rng.rand() < 0.5:
            body += b"\n"
        p = tmp_path / ("t%d.txt" % i)
        p.write_bytes(body)
        assert H.process_file_by_lines(p) == R.process_file_by_lines(p), body
        assert H.read_last_line(p) == R.read_last_line(p), body
    d = tmp_path / "frames"
    d.mkdir()
    (d / "sub_00001.png").mkdir()
    for i in rng.permutation(40):
        (d / (("depth_%05d.png" if i % 3 else "color_%05d.png") % i)).write_bytes(b"")
    (d / "depth_0000x.png").write_bytes(b"")
    for tmpl in [(b"depth_", 5, b"", b"png"), (b"color_", 5, b"", b"png"), (b"sub_", 5, b"", b"png"), (b"none_", 5, b"", b"png")]:
        got = H.files_in_directory(d, *tmpl)
        assert got == R.files_in_directory(d, *tmpl)      # (the directory's own order on both sides)
    assert len(H.files_in_directory(d, b"depth_", 5, b"", b"png")) == 26
    assert H.files_in_directory(tmp_path / "no_such_dir", b"", 0, b"", b"") == R.files_in_directory(tmp_path / "no_such_dir", b"", 0, b"", b"") == []
    for path in [d, d / "depth_00001.png", tmp_path / "nothing", "/dev/null"]:
        for preset in (False, True):
            assert H.file_exists(path, preset) == R.file_exists(path, preset), path
    capfd.readouterr()
