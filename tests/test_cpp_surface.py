"""The C++ class surface (libtsdf_host.so: TSDFVolume / Camera / GPURaycaster / BilateralFilter / DepthImage /
TUMDataLoader / extract_surface ...) on the GPU.

  * build/test_surface (tests/cpp/test_surface.cpp) drives the classes the way the reference's kinfu.cpp does and
    dumps raw results; they must be bit-identical to what the oracle computes from the same inputs.
  * where $TSDF_REF_BUILD/linkcheck/bin/kinfu exists (the REFERENCE's unchanged src/Tools/kinfu.cpp compiled against this
    repo's headers by tools/linkcheck.sh -- outside the tree, so only on a machine that has both the reference and a GPU), it is
    run end to end on a synthetic TUM-layout directory.
"""
import os
import subprocess

import numpy as np
import pytest

from tests.helpers import H, W, assert_same_floats
from tsdf_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "test_surface")
REF_BUILD = os.environ.get("TSDF_REF_BUILD") or "/tmp/tsdf_ref_build"
KINFU = os.path.join(REF_BUILD, "linkcheck", "bin", "kinfu")


def test_headers_of_the_class_surface_compile_standalone():
    """Every public header compiles on its own with g++ (no GPU, no HIP headers needed by a caller)."""
    inc = os.path.join(ROOT, "tsdf_amd", "host", "include")
    eigen = os.path.join(ROOT, "tsdf_amd", "host", "eigen_compat")
    for h in sorted(os.listdir(inc)):
        if h.endswith(".hpp"):
            src = '#include "%s"\nint main(){return 0;}\n' % h
            subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I" + inc, "-I" + eigen, "-I" + os.path.join(ROOT, "include"),
                            "-x", "c++", "-"], input=src.encode(), check=True)


def test_file_utilities_on_the_host(tmp_path):
    """The small host helpers of src/include/FileUtilities.hpp (no GPU involved): a throw-away program linked against
    libtsdf_host.so."""
    src = r"""
#include <cstdio>
#include <cstdlib>
#include <string>
#include "FileUtilities.hpp"
int main(int argc, char **argv) {
    if (!match_file_name("depth_", 4, "_x", "png", "depth_0012_x.png")) return 1;
    if (match_file_name("depth_", 4, "_x", "png", "depth_001a_x.png")) return 2;      // not all digits
    if (match_file_name("depth_", 4, "_x", "png", "depth_00123_x.png")) return 3;     // wrong length
    if (match_file_name("depth_", 4, "", "png", "depth_0012.pgm")) return 4;          // wrong extension
    std::string last;
    if (!read_last_line(argv[1], last) || last != "the last line") return 5;
    if (read_last_line(std::string(argv[1]) + ".missing", last)) return 6;
    setenv("HOME", "/somewhere", 1);
    if (std::string(get_home_directory()) != "/somewhere") return 7;
    if (path_to_file_on_desktop("a.png") != "/somewhere/Desktop/a.png") return 8;
    bool is_dir = false;
    if (!file_exists(argv[1], is_dir) || is_dir) return 9;
    std::puts("file utilities ok");
    return 0;
}
"""
    (tmp_path / "t.cpp").write_text(src)
    (tmp_path / "lines.txt").write_text("first\n\nthe last line\n\n\n")
    lib = os.path.join(ROOT, "tsdf_amd", "lib")
    exe = str(tmp_path / "t")
    subprocess.run(["g++", "-std=c++11", "-I" + os.path.join(ROOT, "tsdf_amd", "host", "include"), str(tmp_path / "t.cpp"), "-o", exe,
                    "-L" + lib, "-ltsdf_host", "-ltsdf_hip", "-Wl,-rpath," + lib], check=True)
    r = subprocess.run([exe, str(tmp_path / "lines.txt")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "file utilities ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_cpp_surface_matches_the_oracle(tmp_path, oracle):
    if not os.path.exists(BIN):
        pytest.fail("build/test_surface missing: run `make cpptest` (build() does)")
    n = 64
    depth, cam = synth.depth_frame(2, 30, seed=0x5EED0001)
    depth.tofile(str(tmp_path / "depth.u16"))
    cam.pose().astype(np.float32).tofile(str(tmp_path / "pose.f32"))
    # a text-format volume for BlockTSDFLoader::to_tsdf
    from tests.test_block_loader import write_block_file
    rng = np.random.default_rng(9)
    Db = rng.uniform(-20, 20, size=(6, 5, 7)).astype(np.float32)
    write_block_file(tmp_path / "block.txt", Db, np.ones_like(Db), (700.0, 500.0, 600.0))
    r = subprocess.run([BIN, str(tmp_path / "depth.u16"), str(tmp_path / "pose.f32"), str(tmp_path), str(n)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "test_surface ok" in r.stdout
    assert np.array_equal(np.fromfile(str(tmp_path / "block_distances.f32"), np.float32).reshape(6, 5, 7), Db)
    assert list(np.fromfile(str(tmp_path / "block_dims.f32"), np.float32)) == [7.0, 5.0, 6.0, 700.0]

    filtered = np.fromfile(str(tmp_path / "filtered.u16"), np.uint16)
    exp_f = oracle.bilateral_u16(depth, W, H, 30.0, 4.5, nthreads=oracle.max_threads()).reshape(-1)
    assert np.array_equal(filtered, exp_f)

    ov = oracle.Volume((n, n, n), (3000, 3000, 3000))
    ov.integrate(exp_f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=oracle.max_threads())
    V = np.fromfile(str(tmp_path / "vertices.f32"), np.float32).reshape(-1, 3)
    N = np.fromfile(str(tmp_path / "normals.f32"), np.float32).reshape(-1, 3)
    assert_same_floats(V, Vo, "C++ raycast vertices")
    assert_same_floats(N, No, "C++ raycast normals")
    # save_to_file -> file constructor -> raycast gives the same picture
    V2 = np.fromfile(str(tmp_path / "vertices_loaded.f32"), np.float32).reshape(-1, 3)
    assert_same_floats(V2, Vo, "raycast of the reloaded volume")
    # the .tsdf file has the reference's layout: 68-byte header + 4 N + 4 N + 3 N + 24 N bytes
    assert os.path.getsize(str(tmp_path / "volume.tsdf")) == 68 + n ** 3 * (4 + 4 + 3 + 24)
    hdr = np.fromfile(str(tmp_path / "volume.tsdf"), np.uint32, 3)
    assert tuple(hdr) == (n, n, n)
    # checkpoints of volumes that left the constructor's state: (a) offset() + clear() + offset() -- the loaded volume keeps
    # both offsets (Q1) and integrates like the original, which integrates like the oracle
    da = np.fromfile(str(tmp_path / "offset_dist_original.f32"), np.float32)
    db = np.fromfile(str(tmp_path / "offset_dist_loaded.f32"), np.float32)
    assert_same_floats(db, da, "integrate after save/load of a volume cleared under an offset")
    oq = oracle.Volume((32, 32, 32), (3000, 3000, 3000))
    oq.offset(37.5, -20.25, 11.0)
    oq.clear()
    oq.offset(-3.0, 4.5, 0.75)
    oq.integrate(exp_f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=oracle.max_threads())
    assert_same_floats(da, oq.dist, "integrate under both offsets vs oracle")
    assert (da != da[0]).any()
    # (b) edited deformation nodes are written and read back verbatim
    ns = np.fromfile(str(tmp_path / "warp_nodes_set.f32"), np.float32)
    assert ns.size == 12 ** 3 * 6
    assert np.array_equal(ns, np.fromfile(str(tmp_path / "warp_nodes_in_file.f32"), np.float32))
    assert np.array_equal(ns, np.fromfile(str(tmp_path / "warp_nodes_loaded.f32"), np.float32))
    # render_to_depth_image: camera-space z of the vertices, rounded (misses -> 0)
    rd = np.fromfile(str(tmp_path / "rendered_depth.u16"), np.uint16)
    hit = ~np.isnan(Vo[:, 0])
    ip = cam.inverse_pose().reshape(4, 4).T
    z = (Vo[hit].astype(np.float64) @ ip[2, :3].astype(np.float64)) + float(ip[2, 3])
    assert np.all(np.abs(rd[hit].astype(np.float64) - z) <= 0.51)
    assert np.all(rd[~hit] == 0)
    # the mesh lies on the zero level set: every vertex is within a voxel diagonal of a ray-cast surface depth range
    mv = np.fromfile(str(tmp_path / "mesh_vertices.f32"), np.float32).reshape(-1, 3)
    assert mv.shape[0] > 1000 and mv.shape[0] % 3 == 0
    assert mv.min() >= 0 and mv.max() <= 3000
    # ICPOdometry (the flow of src/Tools/tsdf_icp.cpp: model = input depth, current = depth rendered from the volume):
    # same pose as the oracle's ICP on the same two images, within the tolerance of the fp32 sums (tests/test_parity_icp.py)
    T = np.fromfile(str(tmp_path / "icp_transform.f64"), np.float64).reshape(4, 4).T
    stats = np.fromfile(str(tmp_path / "icp_stats.f32"), np.float32)
    To, erro, inlo = oracle.icp_incremental_transformation(rd, filtered, W, H, 331.0, 234.6, 591.1, 590.1)
    assert np.max(np.abs(T - To)) < 2e-4, (T, To)
    assert abs(stats[1] - inlo) <= 0.002 * inlo and inlo > 1000


@pytest.mark.gpu
def test_reference_kinfu_runs_end_to_end_against_this_library(tmp_path):
    if not os.path.exists(KINFU):
        pytest.skip("linkcheck/bin/kinfu not present (compiled reference code stays in the build container)")
    d = tmp_path / "tum"
    synth.write_tum_directory(str(d), 3, seed=0x5EED0002)
    r = subprocess.run([KINFU, "-m", "3", "-d", str(d)], capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out
    for line in ("Integrating frame 2", "Raycasting", "Extracting ISO surface"):
        assert line in out, out
    # kinfu prints how many mesh vertices it wrote (its output paths are hard-coded to the author's desktop)
    import re
    m = re.search(r"Writing (\d+) vertices and (\d+) triangles", out)
    assert m and int(m.group(1)) > 1000


@pytest.mark.gpu
def test_reference_tsdf_icp_runs_against_this_library(tmp_path):
    """src/Tools/tsdf_icp.cpp of the reference, compiled unchanged (tools/linkcheck.sh): loads a .tsdf volume and a depth
    PNG, renders the volume from the pose stored in it and runs ICPOdometry between the two images."""
    tool = os.path.join(REF_BUILD, "linkcheck", "bin", "tsdf_icp")
    if not os.path.exists(tool) or not os.path.exists(BIN):
        pytest.skip("linkcheck/bin/tsdf_icp not present (compiled reference code stays in the build container)")
    # a volume file written by the class surface (test_surface saves <out>/volume.tsdf) ...
    depth, cam = synth.depth_frame(2, 30, seed=0x5EED0001)
    depth.tofile(str(tmp_path / "depth.u16"))
    cam.pose().astype(np.float32).tofile(str(tmp_path / "pose.f32"))
    r = subprocess.run([BIN, str(tmp_path / "depth.u16"), str(tmp_path / "pose.f32"), str(tmp_path), "64"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # ... and a 16-bit depth PNG
    synth._write_png16(str(tmp_path / "depth.png"), depth.reshape(H, W))
    r = subprocess.run([tool, "-v", str(tmp_path / "volume.tsdf"), "-d", str(tmp_path / "depth.png")],
                       capture_output=True, text=True, timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out
    assert "trans :" in out and "rot :" in out, out
    import re
    nums = [float(x) for x in re.findall(r"[-+]?\d*\.?\d+(?:[eE][-+]?\d+)?", out.split("trans :")[1])]
    assert len(nums) >= 12 and np.all(np.isfinite(nums[:12]))
    R = np.array(nums[3:12]).reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-4)          # a rotation (printed in float)
