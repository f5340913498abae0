"""ctypes binding of the C ABI declared in include/tsdf_amd.h (tsdf_amd/lib/libtsdf_hip.so).

There is no fallback: if the HIP library is missing or fails to load, importing this module
raises.  The CPU checker used by the tests lives outside this package and is never loaded from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (TSDF_HIP_LIB: another build of the same library, for A/B timing of kernel variants on one box -- tools/ab_variants.sh)
LIB_PATH = os.environ.get("TSDF_HIP_LIB") or os.path.join(_HERE, "lib", "libtsdf_hip.so")

TSDF_OK, TSDF_ERR_INVALID, TSDF_ERR_DEVICE, TSDF_ERR_NOMEM = 0, 1, 2, 3


class TsdfError(RuntimeError):
    """A C-ABI call failed with TSDF_ERR_DEVICE / TSDF_ERR_NOMEM."""


class VolumeInfo(C.Structure):
    """struct tsdf_volume_info (include/tsdf_amd.h)."""
    _fields_ = [("size", C.c_uint32 * 3), ("z_begin", C.c_uint32), ("z_end", C.c_uint32),
                ("z_store_begin", C.c_uint32), ("z_store_end", C.c_uint32),
                ("physical_size", C.c_float * 3), ("voxel_size", C.c_float * 3), ("offset", C.c_float * 3),
                ("offset_at_clear", C.c_float * 3), ("truncation_distance", C.c_float),
                ("max_weight", C.c_float), ("global_translation", C.c_float * 3),
                ("global_rotation", C.c_float * 3), ("deformation_materialised", C.c_int32),
                ("fast_division_verified", C.c_int32)]


class CameraMatrices(C.Structure):
    """struct tsdf_camera_matrices (include/tsdf_amd.h)."""
    _fields_ = [("pose", C.c_float * 16), ("inv_pose", C.c_float * 16), ("k", C.c_float * 9), ("kinv", C.c_float * 9)]


#: tsdf_exchange_fn (include/tsdf_amd.h): int (*)(void *user, const record *mine, record *all, uint32_t n_pixels, void *stream)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "tsdf_amd: %s is missing -- build it with `make hip` (or `python -c 'import __graft_entry__ as g; "
            "g.build()'`). There is no CPU fallback." % LIB_PATH)
    # RTLD_GLOBAL on purpose: torch ships its own libamdhip64 under a different NEEDED name, so a process can hold
    # two HIP runtimes; with the first-loaded one in the global scope everything binds to that single runtime
    # whichever of torch / this library is imported first.  (Only tsdf_* and tsdf:: symbols are exported here.)
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _fp = C.c_void_p, C.POINTER(C.c_float)
_u32, _f, _i = C.c_uint32, C.c_float, C.c_int
_SIGS = {
    "tsdf_last_error": (C.c_char_p, []),
    "tsdf_build_arch": (C.c_char_p, []),
    "tsdf_device_count": (_i, [C.POINTER(_i)]),
    "tsdf_set_device": (_i, [_i]),
    "tsdf_get_device": (_i, [C.POINTER(_i)]),
    "tsdf_device_alloc": (_i, [C.c_size_t, C.POINTER(_vp)]),
    "tsdf_device_free": (_i, [_vp]),
    "tsdf_device_upload": (_i, [_vp, _vp, C.c_size_t]),
    "tsdf_device_download": (_i, [_vp, _vp, C.c_size_t]),
    "tsdf_stream_synchronize": (_i, [_vp]),
    "tsdf_volume_create": (_i, [_u32, _u32, _u32, _f, _f, _f, C.POINTER(_vp)]),
    "tsdf_volume_create_slab": (_i, [_u32, _u32, _u32, _f, _f, _f, _u32, _u32, C.POINTER(_vp)]),
    "tsdf_volume_destroy": (_i, [_vp]),
    "tsdf_volume_set_stream": (_i, [_vp, _vp]),
    "tsdf_volume_stream": (_i, [_vp, C.POINTER(_vp)]),
    "tsdf_volume_synchronize": (_i, [_vp]),
    "tsdf_volume_clear": (_i, [_vp]),
    "tsdf_volume_get_info": (_i, [_vp, C.POINTER(VolumeInfo)]),
    "tsdf_volume_set_offset": (_i, [_vp, _f, _f, _f]),
    "tsdf_volume_set_header": (_i, [_vp, _fp, _f, _f, _fp, _fp]),
    "tsdf_volume_mark_dirty": (_i, [_vp]),
    "tsdf_volume_distances": (_i, [_vp, C.POINTER(_vp)]),
    "tsdf_volume_weights": (_i, [_vp, C.POINTER(_vp)]),
    "tsdf_volume_weight_storage": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "tsdf_volume_last_raycast_kind": (_i, [_vp, C.POINTER(_i)]),
    "tsdf_volume_last_cell_list": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "tsdf_volume_set_weight_storage": (_i, [_vp, _i]),
    "tsdf_selftest_count_division": (_i, [_u32, _u32, C.POINTER(C.c_uint64)]),
    "tsdf_volume_deformation": (_i, [_vp, C.POINTER(_vp)]),
    "tsdf_volume_set_distance_data": (_i, [_vp, _vp]),
    "tsdf_volume_set_weight_data": (_i, [_vp, _vp]),
    "tsdf_volume_set_deformation": (_i, [_vp, _vp]),
    "tsdf_volume_get_distance_data": (_i, [_vp, _vp]),
    "tsdf_measure_copy_bandwidth": (_i, [C.c_size_t, _i, _vp, C.POINTER(C.c_double)]),
    "tsdf_measure_update_bandwidth": (_i, [_i, _vp, C.POINTER(C.c_double)]),
    "tsdf_volume_get_weight_data": (_i, [_vp, _vp]),
    "tsdf_volume_get_deformation_planes": (_i, [_vp, C.c_uint32, C.c_uint32, _vp]),
    "tsdf_volume_set_offset_at_clear": (_i, [_vp, _vp]),
    "tsdf_integrate": (_i, [_vp, _vp, _u32, _u32, _fp, _fp, _fp, _fp]),
    "tsdf_integrate_device": (_i, [_vp, _vp, _u32, _u32, _fp, _fp, _fp, _fp]),
    "tsdf_integrate_device_tiles": (_i, [_vp, _vp, _u32, _u32, _fp, _fp, _fp, _fp, _vp]),
    "tsdf_integrate_prepare_device_tiles": (_i, [_vp, _vp, _u32, _u32, _fp, _fp, _fp, _fp, _vp, _vp]),
    "tsdf_integrate_discard_prepared": (_i, [_vp]),
    "tsdf_volume_set_timing": (_i, [_vp, _i]),
    "tsdf_volume_kernel_time": (_i, [_vp, _i, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "tsdf_volume_set_counting": (_i, [_vp, _i]),
    "tsdf_volume_last_updated_voxels": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "tsdf_raycast": (_i, [_vp, _u32, _u32, _fp, _fp, _vp, _vp]),
    "tsdf_raycast_device": (_i, [_vp, _u32, _u32, _fp, _fp, _vp, _vp]),
    "tsdf_normals_device": (_i, [_u32, _u32, _vp, _vp, _vp]),
    "tsdf_raycast_stats": (_i, [_vp, _u32, _u32, _fp, _fp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                C.POINTER(C.c_uint64)]),
    "tsdf_raycast_evaluated_samples": (_i, [_vp, _u32, _u32, _fp, _fp, C.POINTER(C.c_uint64), _vp]),
    "tsdf_volume_deform_points": (_i, [_vp, _i, _vp]),
    "tsdf_volume_deform_points_device": (_i, [_vp, _i, _vp]),
    "tsdf_volume_occupancy": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "tsdf_volume_get_occupancy_data": (_i, [_vp, C.c_int, _vp, _vp, _vp]),
    "tsdf_raycast_slab_device": (_i, [_vp, _u32, _u32, _fp, _fp, _vp]),
    "tsdf_vertices_to_depth_device": (_i, [_u32, _u32, _vp, _vp, _vp, _vp]),
    "tsdf_raycast_depth_device": (_i, [_vp, _u32, _u32, _fp, _fp, _fp, _vp, _vp]),
    "tsdf_volume_marching_cubes": (_i, [_vp, _vp, _vp, _vp, C.c_uint64]),
    "tsdf_merge_hits_device": (_i, [_vp, _vp, _u32, _u32, _u32, _fp, _fp, _vp, _vp]),
    "tsdf_merge_hits_normals_device": (_i, [_vp, _vp, _u32, _u32, _u32, _fp, _fp, _vp, _vp, _vp]),
    "tsdf_slab_exchange_unique_id": (_i, [_vp, C.c_char_p]),
    "tsdf_slab_exchange_create": (_i, [_i, _i, _vp, C.c_char_p, C.POINTER(_vp)]),
    "tsdf_slab_exchange_create_callback": (_i, [_i, _i, EXCHANGE_FN, _vp, C.POINTER(_vp)]),
    "tsdf_slab_exchange_create_loopback": (_i, [_i, _i, C.POINTER(_vp)]),
    "tsdf_slab_exchange_world": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "tsdf_slab_exchange_ranks_seen": (_i, [_vp, C.POINTER(_i)]),
    "tsdf_slab_validate_merge": (_i, [_vp, _vp, _u32, _u32, _fp, _fp, _vp, _vp, C.POINTER(C.c_uint64)]),
    "tsdf_slab_exchange_all_gather": (_i, [_vp, _vp, _vp, _u32, _vp]),
    "tsdf_slab_exchange_destroy": (_i, [_vp]),
    "tsdf_pipeline_create": (_i, [_vp, _vp, _u32, _u32, _i, _vp, C.POINTER(_vp)]),
    "tsdf_pipeline_step": (_i, [_vp, _vp, C.POINTER(CameraMatrices), _vp, _vp, _vp, C.POINTER(CameraMatrices)]),
    "tsdf_pipeline_synchronize": (_i, [_vp]),
    "tsdf_pipeline_streams": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "tsdf_pipeline_hit_buffers": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "tsdf_pipeline_destroy": (_i, [_vp]),
    "tsdf_tracker_create": (_i, [_vp, _vp, _vp, _u32, _u32, _f, _i, C.POINTER(_vp)]),
    "tsdf_tracker_filter": (_i, [_vp, _vp]),
    "tsdf_tracker_align": (_i, [_vp, C.POINTER(CameraMatrices), _vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tsdf_tracker_integrate": (_i, [_vp, C.POINTER(CameraMatrices)]),
    "tsdf_tracker_synchronize": (_i, [_vp]),
    "tsdf_tracker_streams": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "tsdf_tracker_buffers": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "tsdf_tracker_destroy": (_i, [_vp]),
    "tsdf_icp_create": (_i, [_i, _i, _f, _f, _f, _f, _f, _f, C.POINTER(_vp)]),
    "tsdf_icp_destroy": (None, [_vp]),
    "tsdf_icp_set_stream": (_i, [_vp, _vp]),
    "tsdf_icp_stream": (_i, [_vp, C.POINTER(_vp)]),
    "tsdf_icp_init": (_i, [_vp, _i, _vp, _f]),
    "tsdf_icp_init_device": (_i, [_vp, _i, _vp, _f]),
    "tsdf_icp_estimate_step": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "tsdf_icp_get_incremental_transformation": (_i, [_vp, _vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tsdf_icp_get_map": (_i, [_vp, _i, _i, _vp]),
    "tsdf_icp_get_depth_level": (_i, [_vp, _i, _vp]),
    "tsdf_bilateral_create": (_i, [_f, _f, C.POINTER(_vp)]),
    "tsdf_bilateral_destroy": (_i, [_vp]),
    "tsdf_bilateral_filter_u8": (_i, [_vp, _vp, _i, _i]),
    "tsdf_bilateral_filter_u16": (_i, [_vp, _vp, _i, _i]),
    "tsdf_bilateral_filter_u8_device": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "tsdf_bilateral_filter_u16_device": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "tsdf_bilateral_filter_u16_device_tiles": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
}
#: every symbol include/tsdf_amd.h declares
EXPORTS = tuple(_SIGS)
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)   # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    return lib.tsdf_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a C-ABI status to the reference's error behaviour: invalid argument -> ValueError
    (std::invalid_argument in the C++ surface), anything else -> TsdfError."""
    if rc == TSDF_OK:
        return
    msg = last_error()
    if rc == TSDF_ERR_INVALID:
        raise ValueError(msg)
    if rc == TSDF_ERR_NOMEM:
        raise MemoryError(msg)
    raise TsdfError(msg)

# ---- host library (C++ class surface; Camera is exposed to Python through it) ------------------
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libtsdf_host.so")
if not os.path.exists(HOST_LIB_PATH):
    raise ImportError("tsdf_amd: %s is missing -- build it with `make host`." % HOST_LIB_PATH)
# RTLD_LOCAL: this library defines the reference's class names (BilateralFilter, Camera, ...); they must not
# interpose the same names inside other shared objects (e.g. the reference build the tests compare against).
host = C.CDLL(HOST_LIB_PATH)
_ip = C.POINTER(C.c_int)
_HOST_SIGS = {
    "tsdf_camera_create": (_vp, [_f, _f, _f, _f]),
    "tsdf_camera_destroy": (None, [_vp]),
    "tsdf_camera_get": (None, [_vp, _fp, _fp, _fp, _fp]),
    "tsdf_camera_set_pose": (None, [_vp, _fp]),
    "tsdf_camera_set_pose_tum": (None, [_vp, _fp]),
    "tsdf_camera_move_to": (None, [_vp, _f, _f, _f]),
    "tsdf_camera_look_at": (None, [_vp, _f, _f, _f]),
    "tsdf_camera_world_to_camera": (None, [_vp, _fp, _fp]),
    "tsdf_camera_camera_to_world": (None, [_vp, _fp, _fp]),
    "tsdf_camera_world_to_pixel": (None, [_vp, _fp, _ip]),
    "tsdf_camera_pixel_to_image_plane": (None, [_vp, C.c_uint16, C.c_uint16, _fp]),
    "tsdf_camera_image_plane_to_pixel": (None, [_vp, _fp, _ip]),
    "tsdf_host_marching_cubes_c": (C.c_size_t, [_vp, C.c_uint, C.c_uint, C.c_uint, _vp, _vp, _vp, C.c_size_t]),
    "tsdf_host_mc_table": (None, [_vp]),
    "tsdf_host_tum_open": (_vp, [C.c_char_p]),
    "tsdf_host_tum_next": (_i, [_vp, _vp, C.c_size_t, C.POINTER(C.c_uint), _fp]),
    "tsdf_host_tum_close": (None, [_vp]),
    "tsdf_host_block_loader_parse": (_i, [C.c_char_p, _vp, _vp, _vp, _vp, C.c_size_t]),
    "tsdf_host_write_ply": (None, [C.c_char_p, _vp, C.c_size_t, _vp, C.c_size_t]),
    "tsdf_host_read_nyu_depth_map": (C.c_size_t, [C.c_char_p, C.POINTER(C.c_uint), _vp, C.c_size_t]),
    "tsdf_host_match_file_name": (_i, [C.c_char_p, _i, C.c_char_p, C.c_char_p, C.c_char_p]),
    "tsdf_host_process_file_by_lines": (C.c_size_t, [C.c_char_p, C.POINTER(_i), C.c_char_p, C.c_size_t]),
    "tsdf_host_read_last_line": (C.c_size_t, [C.c_char_p, C.c_char_p, C.POINTER(_i), C.c_char_p, C.c_size_t]),
    "tsdf_host_files_in_directory": (C.c_size_t, [C.c_char_p, C.c_char_p, _i, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    "tsdf_host_file_exists": (_i, [C.c_char_p, C.POINTER(_i)]),
}
for _name, (_res, _args) in _HOST_SIGS.items():
    _fn = getattr(host, _name)
    _fn.restype = _res
    _fn.argtypes = _args
