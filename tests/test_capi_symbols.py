"""The C-ABI library loads and exports every symbol include/tsdf_amd.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "tsdf_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(tsdf_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_whole_path():
    names = declared_functions()
    for needed in ("tsdf_volume_create", "tsdf_integrate", "tsdf_integrate_device", "tsdf_raycast",
                   "tsdf_raycast_device", "tsdf_normals_device", "tsdf_bilateral_filter_u16",
                   "tsdf_raycast_slab_device", "tsdf_merge_hits_device"):
        assert needed in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "tsdf_amd", "lib", "libtsdf_hip.so"))
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "declared in include/tsdf_amd.h but not exported: %s" % missing
    assert lib.tsdf_build_arch is not None


def test_python_binding_covers_every_declared_symbol():
    from tsdf_amd import _capi
    assert sorted(_capi.EXPORTS) == declared_functions()
    arch = _capi.lib.tsdf_build_arch()
    assert arch == b"gfx950"


def test_invalid_sizes_are_rejected_before_touching_the_device():
    # same contract as the reference's constructors (src/TSDF/TSDFVolume.cu:431-436): std::invalid_argument
    import pytest
    import tsdf_amd
    with pytest.raises(ValueError):
        tsdf_amd.TSDFVolume((0, 8, 8))
    with pytest.raises(ValueError):
        tsdf_amd.TSDFVolume((8, 8, 8), (3000.0, 0.0, 3000.0))
    with pytest.raises(ValueError):
        tsdf_amd.TSDFVolume((70000, 8, 8))
    with pytest.raises(ValueError, match="too large"):      # the ray caster indexes 4^3-voxel bricks with 32 bits
        tsdf_amd.TSDFVolume((65535, 65535, 65535), (1e6, 1e6, 1e6))
    with pytest.raises(ValueError):
        tsdf_amd.BilateralFilter(0.0, 2.0)


def test_product_never_imports_the_oracle():
    """Nothing under tsdf_amd/ may reference oracle/ (the oracle is test infrastructure)."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "tsdf_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(d, f), errors="replace").read()
                if re.search(r"import\s+oracle|from\s+oracle|tsdf_oracle|libtsdf_oracle|oracle/", src):
                    bad.append(os.path.join(d, f))
    assert not bad, bad
