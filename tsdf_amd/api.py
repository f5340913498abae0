"""Python mirror of the reference's class surface for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow src/include/TSDFVolume.hpp,
GPURaycaster.hpp and BilateralFilter.hpp of the reference so that tests read like the
reference's own.  Matrices are column-major float32 vectors (what Eigen's .data() yields).
Every call runs the HIP kernels in tsdf_amd/lib/libtsdf_hip.so; there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib


def _mat(a, n):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d matrix elements, got %d" % (n, a.size))
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _camera_matrices(camera):
    """pose, inverse_pose, k, kinv of anything shaped like the reference's Camera (src/include/Camera.hpp)."""
    return (_mat(camera.pose(), 16), _mat(camera.inverse_pose(), 16), _mat(camera.k(), 9), _mat(camera.kinv(), 9))


class TSDFVolume:
    """src/include/TSDFVolume.hpp:21-304.  `slab=(z_begin, z_end)` makes this object one Z-slab of the
    grid (multi-GPU sharding); the default is the whole volume."""

    def __init__(self, size=(64, 64, 64), physical_size=(3000.0, 3000.0, 3000.0), slab=None):
        self._h = C.c_void_p()
        sx, sy, sz = (int(s) for s in size)
        if min(sx, sy, sz) < 0:
            raise ValueError("Attempt to construct TSDFVolume with zero or negative size")
        px, py, pz = (float(p) for p in physical_size)
        if slab is None:
            check(lib.tsdf_volume_create(sx, sy, sz, px, py, pz, C.byref(self._h)))
        else:
            check(lib.tsdf_volume_create_slab(sx, sy, sz, px, py, pz, int(slab[0]), int(slab[1]), C.byref(self._h)))

    def close(self):
        if lib is not None and getattr(self, "_h", None) is not None and self._h.value:   # (lib is None during interpreter shutdown)
            for ref in getattr(self, "_dependents", ()):      # pipelines built on this volume hold its stream: they go first
                dep = ref()
                if dep is not None:
                    dep.close()
            lib.tsdf_volume_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    # ---- geometry accessors (TSDFVolume.hpp:120-153)
    def info(self):
        i = _capi.VolumeInfo()
        check(lib.tsdf_volume_get_info(self._h, C.byref(i)))
        return i

    def size(self):
        return tuple(self.info().size)

    def voxel_size(self):
        return np.array(self.info().voxel_size, np.float32)

    def physical_size(self):
        return np.array(self.info().physical_size, np.float32)

    def truncation_distance(self):
        return float(self.info().truncation_distance)

    def offset(self, *o):
        """offset() -> current offset; offset(ox, oy, oz) sets it (without re-initialising the deformation grid)."""
        if o:
            check(lib.tsdf_volume_set_offset(self._h, float(o[0]), float(o[1]), float(o[2])))
            return None
        return np.array(self.info().offset, np.float32)

    def set_global_transform(self, rotation, translation):
        """m_global_rotation (three angles) / m_global_translation, used by deform_mesh and stored in .tsdf files."""
        i = self.info()
        off = np.array(i.offset, np.float32)
        r = np.ascontiguousarray(rotation, np.float32)
        t = np.ascontiguousarray(translation, np.float32)
        check(lib.tsdf_volume_set_header(self._h, _fp(off), float(i.truncation_distance), float(i.max_weight), _fp(t), _fp(r)))

    def resident_planes(self):
        i = self.info()
        return int(i.z_store_begin), int(i.z_store_end)

    def owned_planes(self):
        i = self.info()
        return int(i.z_begin), int(i.z_end)

    def resident_voxels(self):
        i = self.info()
        return int(i.size[0]) * int(i.size[1]) * int(i.z_store_end - i.z_store_begin)

    def index(self, x, y, z):
        sx, sy, _ = self.size()
        return x + y * sx + z * sx * sy

    def clear(self):
        check(lib.tsdf_volume_clear(self._h))
        check(lib.tsdf_volume_synchronize(self._h))

    def set_stream(self, hip_stream):
        check(lib.tsdf_volume_set_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else 0)))

    def stream_ptr(self):
        """The HIP stream (as an integer, 0 = the null stream) the volume's kernels are enqueued on now."""
        p = C.c_void_p()
        check(lib.tsdf_volume_stream(self._h, C.byref(p)))
        return p.value or 0

    def synchronize(self):
        check(lib.tsdf_volume_synchronize(self._h))

    # ---- data access (TSDFVolume.hpp:165-203): device pointers, blocking uploads
    def distance_data(self):
        p = C.c_void_p()
        check(lib.tsdf_volume_distances(self._h, C.byref(p)))
        return p.value

    def weight_data(self):
        p = C.c_void_p()
        check(lib.tsdf_volume_weights(self._h, C.byref(p)))
        return p.value

    def weight_storage(self):
        """(bits per stored weight: 8 / 16 = packed counts, 32 = the reference's fp32 array; pinned to fp32 by weight_data())"""
        bits, pinned = C.c_int(), C.c_int()
        check(lib.tsdf_volume_weight_storage(self._h, C.byref(bits), C.byref(pinned)))
        return bits.value, bool(pinned.value)

    def last_raycast_cell_parallel(self):
        """True when the volume's last ray cast took the cell-parallel kernels (scheduling only: the same bits as the march)."""
        k = C.c_int()
        check(lib.tsdf_volume_last_raycast_kind(self._h, C.byref(k)))
        return bool(k.value)

    def last_cell_list(self):
        """Tasks the last cell-parallel cast listed (diagnostics; waits for the volume's stream)."""
        n = C.c_uint32()
        check(lib.tsdf_volume_last_cell_list(self._h, C.byref(n)))
        return int(n.value)

    def set_weight_storage(self, bits):
        """Widen the weight storage now (8 -> 16 -> 32 bits, values unchanged) instead of when a count is about to overflow."""
        check(lib.tsdf_volume_set_weight_storage(self._h, int(bits)))

    def deformation(self):
        p = C.c_void_p()
        check(lib.tsdf_volume_deformation(self._h, C.byref(p)))
        return p.value

    def _host(self, a, per_voxel=1):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        if a.size != self.resident_voxels() * per_voxel:
            raise ValueError("expected %d floats, got %d" % (self.resident_voxels() * per_voxel, a.size))
        return a

    def set_distance_data(self, distance_data):
        a = self._host(distance_data)
        check(lib.tsdf_volume_set_distance_data(self._h, a.ctypes.data))

    def set_weight_data(self, weight_data):
        a = self._host(weight_data)
        check(lib.tsdf_volume_set_weight_data(self._h, a.ctypes.data))

    def set_deformation(self, nodes):
        """nodes: (voxels, 6) float32 = translation xyz + rotation xyz (DeformationNode, TSDFVolume.hpp:23-26)."""
        a = self._host(nodes, 6)
        check(lib.tsdf_volume_set_deformation(self._h, a.ctypes.data))

    def extract_surface(self):
        """extract_surface on the device (tsdf_volume_marching_cubes): (3*T, 3) float32 vertices, triangle t = rows 3t,
        3t+1, 3t+2, cubes in the reference's order -- the same array as marching_cubes() on the downloaded distances."""
        table = marching_cubes_table()
        n = C.c_uint64(0)
        check(lib.tsdf_volume_marching_cubes(self._h, table.ctypes.data, C.byref(n), None, 0))
        out = np.empty((n.value, 3), np.float32)
        if n.value:
            check(lib.tsdf_volume_marching_cubes(self._h, table.ctypes.data, C.byref(n), out.ctypes.data, n.value))
        return out

    def deform_mesh(self, points):
        """TSDFVolume::deform_mesh (src/TSDF/TSDFVolume.cu:265-291): points (n,3) float32 -> deformed copy."""
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3).copy()
        check(lib.tsdf_volume_deform_points(self._h, p.shape[0], p.ctypes.data))
        return p

    def get_distance_data(self):
        a = np.empty(self.resident_voxels(), np.float32)
        check(lib.tsdf_volume_get_distance_data(self._h, a.ctypes.data))
        return a

    def get_weight_data(self):
        a = np.empty(self.resident_voxels(), np.float32)
        check(lib.tsdf_volume_get_weight_data(self._h, a.ctypes.data))
        return a

    # ---- integrate (TSDFVolume.hpp:238-246)
    def integrate(self, depth_map, width, height, camera):
        """Blocking; depth_map is a host uint16 array of width*height mm values (0 = invalid)."""
        if depth_map is None:
            raise AssertionError("depth_map")        # the reference asserts (TSDFVolume.cu:862)
        d = np.ascontiguousarray(depth_map, dtype=np.uint16).reshape(-1)
        if d.size != width * height:
            raise ValueError("depth map has %d pixels, expected %d" % (d.size, width * height))
        pose, ipose, k, kinv = _camera_matrices(camera)
        check(lib.tsdf_integrate(self._h, d.ctypes.data, width, height, _fp(pose), _fp(ipose), _fp(k), _fp(kinv)))

    def integrate_device(self, depth_ptr, width, height, camera, tile_max_ptr=None):
        """Asynchronous on the volume's stream; depth_ptr is a device pointer to width*height uint16.  tile_max_ptr: the
        16 x 16 pixel tile maxima of that image when the caller holds them (BilateralFilter.filter_device(tile_max_ptr=...))."""
        pose, ipose, k, kinv = _camera_matrices(camera)
        if tile_max_ptr:
            check(lib.tsdf_integrate_device_tiles(self._h, C.c_void_p(int(depth_ptr)), width, height, _fp(pose), _fp(ipose),
                                                  _fp(k), _fp(kinv), C.c_void_p(int(tile_max_ptr))))
        else:
            check(lib.tsdf_integrate_device(self._h, C.c_void_p(int(depth_ptr)), width, height, _fp(pose), _fp(ipose),
                                            _fp(k), _fp(kinv)))

    def integrate_prepare_device(self, depth_ptr, width, height, camera, tile_max_ptr, stream):
        """The brick culling of integrate_device(depth_ptr, ..., tile_max_ptr=...) ahead of time, on `stream` (see
        tsdf_integrate_prepare_device_tiles); the matching integrate_device call then launches the integrate kernel alone."""
        pose, ipose, k, kinv = _camera_matrices(camera)
        check(lib.tsdf_integrate_prepare_device_tiles(self._h, C.c_void_p(int(depth_ptr)), width, height, _fp(pose), _fp(ipose),
                                                      _fp(k), _fp(kinv), C.c_void_p(int(tile_max_ptr)), C.c_void_p(int(stream))))

    def occupancy(self):
        """(occupied, total) bricks of the ray caster's empty-space summary."""
        o, t = C.c_uint64(), C.c_uint64()
        check(lib.tsdf_volume_occupancy(self._h, C.byref(o), C.byref(t)))
        return int(o.value), int(t.value)

    def occupancy_data(self, force_rebuild=False):
        """(fine, cell, reach) uint8 arrays of shape (nbz, nby, nbx): the ray caster's brick flags (diagnostics)."""
        X, Y, Z = self.size()
        shape = ((Z + 3) // 4, (Y + 3) // 4, (X + 3) // 4)
        out = [np.empty(shape, np.uint8) for _ in range(3)]
        check(lib.tsdf_volume_get_occupancy_data(self._h, 1 if force_rebuild else 0, *[a.ctypes.data for a in out]))
        return tuple(out)

    def set_timing(self, enabled):
        """HIP-event timing of integrate_kernel / process_ray_kernel launches on the volume's stream: True = every
        launch, an integer n > 1 = every n-th launch, False / 0 = off."""
        check(lib.tsdf_volume_set_timing(self._h, int(enabled)))

    def kernel_time(self, which):
        """(launches, average ms) of which = 'integrate' | 'raycast' | 'raycast_tail' since set_timing(True)."""
        n, ms = C.c_uint32(), C.c_float()
        check(lib.tsdf_volume_kernel_time(self._h, {"integrate": 0, "raycast": 1, "raycast_tail": 2}[which], C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def set_counting(self, enabled):
        check(lib.tsdf_volume_set_counting(self._h, 1 if enabled else 0))

    def last_updated_voxels(self):
        c = C.c_uint64()
        check(lib.tsdf_volume_last_updated_voxels(self._h, C.byref(c)))
        return int(c.value)

    # ---- raycast (TSDFVolume.hpp:260): forwards to GPURaycaster like TSDFVolume.cu:1054-1058
    def raycast(self, width, height, camera):
        return GPURaycaster(width, height).raycast(self, camera)


class GPURaycaster:
    """src/include/GPURaycaster.hpp:19-41 (+ Raycaster.hpp:17-39)."""

    def __init__(self, width=640, height=480):
        self.m_width = int(width) & 0xFFFF    # uint16_t members in the reference
        self.m_height = int(height) & 0xFFFF

    def raycast(self, volume, camera):
        """-> (vertices, normals), each (width*height, 3) float32; misses are NaN rows."""
        pose, _, _, kinv = _camera_matrices(camera)
        n = self.m_width * self.m_height
        V = np.empty((n, 3), np.float32)
        N = np.empty((n, 3), np.float32)
        check(lib.tsdf_raycast(volume._h, self.m_width, self.m_height, _fp(pose), _fp(kinv), V.ctypes.data,
                               N.ctypes.data))
        return V, N

    def get_vertices(self, volume, camera):
        pose, _, _, kinv = _camera_matrices(camera)
        V = np.empty((self.m_width * self.m_height, 3), np.float32)
        check(lib.tsdf_raycast(volume._h, self.m_width, self.m_height, _fp(pose), _fp(kinv), V.ctypes.data, None))
        return V

    def raycast_device(self, volume, camera, vertices_ptr, normals_ptr=None):
        pose, _, _, kinv = _camera_matrices(camera)
        check(lib.tsdf_raycast_device(volume._h, self.m_width, self.m_height, _fp(pose), _fp(kinv),
                                      C.c_void_p(int(vertices_ptr)),
                                      C.c_void_p(int(normals_ptr)) if normals_ptr else None))

    def render_to_depth_device(self, volume, camera, depth_ptr, vertices_ptr=None):
        """GPURaycaster::render_to_depth_image on device buffers: uint16 mm per pixel (0 = no hit), the vertex map too when asked for."""
        pose, ipose, _, kinv = _camera_matrices(camera)
        check(lib.tsdf_raycast_depth_device(volume._h, self.m_width, self.m_height, _fp(pose), _fp(ipose), _fp(kinv),
                                            C.c_void_p(int(depth_ptr)), C.c_void_p(int(vertices_ptr)) if vertices_ptr else None))

    def raycast_slab_device(self, volume, camera, hits_ptr):
        pose, _, _, kinv = _camera_matrices(camera)
        check(lib.tsdf_raycast_slab_device(volume._h, self.m_width, self.m_height, _fp(pose), _fp(kinv),
                                           C.c_void_p(int(hits_ptr))))

    def stats(self, volume, camera, per_ray_work=False):
        """Roofline diagnostics of one raycast: samples S, distinct voxels touched T, hits."""
        pose, _, _, kinv = _camera_matrices(camera)
        s, t, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.tsdf_raycast_stats(volume._h, self.m_width, self.m_height, _fp(pose), _fp(kinv), C.byref(s),
                                     C.byref(t), C.byref(h)))
        e = C.c_uint64()
        per_ray = np.empty((self.m_width * self.m_height, 3), np.float32) if per_ray_work else None
        check(lib.tsdf_raycast_evaluated_samples(volume._h, self.m_width, self.m_height, _fp(pose), _fp(kinv), C.byref(e),
                                                 per_ray.ctypes.data if per_ray is not None else None))
        out = {"samples": int(s.value), "touched": int(t.value), "hits": int(h.value), "evaluated": int(e.value)}
        if per_ray is not None:
            out["per_ray"] = per_ray      # columns: samples evaluated, loop trips, reference sample count
        return out


def compute_normals_device(width, height, vertices_ptr, normals_ptr, stream=0):
    check(lib.tsdf_normals_device(width, height, C.c_void_p(int(vertices_ptr)), C.c_void_p(int(normals_ptr)),
                                  C.c_void_p(int(stream) if stream else 0)))


def vertices_to_depth_device(width, height, vertices_ptr, camera, depth_ptr, stream=0):
    """The per-pixel part of GPURaycaster::render_to_depth_image on device buffers (uint16 mm, 0 = no hit)."""
    ip = _mat(camera.inverse_pose(), 16)
    check(lib.tsdf_vertices_to_depth_device(width, height, C.c_void_p(int(vertices_ptr)), ip.ctypes.data,
                                            C.c_void_p(int(depth_ptr)), C.c_void_p(int(stream) if stream else 0)))


#: bytes of one slab hit record {uint32 k, float t} (struct tsdf_hit_record)
HIT_RECORD_BYTES = 8


def merge_hits_device(volume, hits_all_ptr, n_slabs, width, height, camera, vertices_ptr, stream=0):
    """Min-k select over the gathered slab records ((n_slabs, W*H) x {k, t}); the vertex of a pixel is formed from the winning
    record's refined ray parameter and the pixel's own ray (`camera`; `volume`: any slab of the grid, for its offset / size)."""
    pose, _, _, kinv = _camera_matrices(camera)
    check(lib.tsdf_merge_hits_device(volume._h, C.c_void_p(int(hits_all_ptr)), n_slabs, width, height, _fp(pose), _fp(kinv),
                                     C.c_void_p(int(vertices_ptr)), C.c_void_p(int(stream) if stream else 0)))


def merge_hits_normals_device(volume, hits_all_ptr, n_slabs, width, height, camera, vertices_ptr, normals_ptr, stream=0):
    """The same select and the normals of the merged map, one launch."""
    pose, _, _, kinv = _camera_matrices(camera)
    check(lib.tsdf_merge_hits_normals_device(volume._h, C.c_void_p(int(hits_all_ptr)), n_slabs, width, height, _fp(pose), _fp(kinv),
                                             C.c_void_p(int(vertices_ptr)), C.c_void_p(int(normals_ptr)),
                                             C.c_void_p(int(stream) if stream else 0)))


class BilateralFilter:
    """src/include/BilateralFilter.hpp:12-35.  filter() works in place on a host image, as the reference does."""

    def __init__(self, sigma_colour, sigma_space):
        self._h = C.c_void_p()
        check(lib.tsdf_bilateral_create(float(sigma_colour), float(sigma_space), C.byref(self._h)))

    def close(self):
        if lib is not None and getattr(self, "_h", None) is not None and self._h.value:   # (lib is None during interpreter shutdown)
            lib.tsdf_bilateral_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def filter(self, image, width, height):
        if not isinstance(image, np.ndarray) or image.dtype not in (np.uint8, np.uint16) or \
                not image.flags["C_CONTIGUOUS"]:
            raise ValueError("image must be a C-contiguous uint8 or uint16 numpy array")
        if image.size != width * height:
            raise ValueError("image has %d pixels, expected %d" % (image.size, width * height))
        fn = lib.tsdf_bilateral_filter_u8 if image.dtype == np.uint8 else lib.tsdf_bilateral_filter_u16
        check(fn(self._h, image.ctypes.data, width, height))

    def filter_device(self, in_ptr, out_ptr, width, height, bits=16, stream=0, tile_max_ptr=None):
        """tile_max_ptr (16 bit only): device array of ceil(width / 16) * ceil(height / 16) uint16 that receives the largest
        filtered value of every 16 x 16 pixel tile, for TSDFVolume.integrate_device(tile_max_ptr=...)."""
        if tile_max_ptr:
            if bits != 16:
                raise ValueError("tile maxima are produced by the 16-bit filter only")
            check(lib.tsdf_bilateral_filter_u16_device_tiles(self._h, C.c_void_p(int(in_ptr)), C.c_void_p(int(out_ptr)), width, height,
                                                             C.c_void_p(int(tile_max_ptr)), C.c_void_p(int(stream) if stream else 0)))
            return
        fn = lib.tsdf_bilateral_filter_u8_device if bits == 8 else lib.tsdf_bilateral_filter_u16_device
        check(fn(self._h, C.c_void_p(int(in_ptr)), C.c_void_p(int(out_ptr)), width, height,
                 C.c_void_p(int(stream) if stream else 0)))


def marching_cubes(distances, size, voxel_size, offset=(0.0, 0.0, 0.0)):
    """Host marching cubes of the class surface (MarkAndSweepMC.cpp, what extract_surface runs on a volume's distances):
    distances indexed x + y*X + z*X*Y -> (3*T, 3) float32 vertices, triangle t = rows 3t, 3t+1, 3t+2 (the reference then
    wires them (i, i+2, i+1))."""
    X, Y, Z = (int(v) for v in size)
    d = np.ascontiguousarray(distances, dtype=np.float32).reshape(-1)
    if d.size != X * Y * Z:
        raise ValueError("expected %d distances, got %d" % (X * Y * Z, d.size))
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    off = np.ascontiguousarray(offset, dtype=np.float32)
    n = _capi.host.tsdf_host_marching_cubes_c(d.ctypes.data, X, Y, Z, vs.ctypes.data, off.ctypes.data, None, 0)
    out = np.empty((n, 3), np.float32)
    _capi.host.tsdf_host_marching_cubes_c(d.ctypes.data, X, Y, Z, vs.ctypes.data, off.ctypes.data, out.ctypes.data, n)
    return out


def load_block_tsdf(file_name):
    """BlockTSDFLoader::load_from_file (text TSDF format) -> (complete, size xyz, physical size xyz, distances, weights)."""
    size = np.zeros(3, np.uint32)
    phys = np.zeros(3, np.float32)
    path = str(file_name).encode()
    ok = _capi.host.tsdf_host_block_loader_parse(path, size.ctypes.data, phys.ctypes.data, None, None, 0)
    n = int(size[0]) * int(size[1]) * int(size[2])
    d, w = np.zeros(n, np.float32), np.zeros(n, np.float32)
    if n:
        ok = _capi.host.tsdf_host_block_loader_parse(path, size.ctypes.data, phys.ctypes.data, d.ctypes.data, w.ctypes.data, n)
    return bool(ok), tuple(int(v) for v in size), tuple(float(v) for v in phys), d, w


def marching_cubes_table():
    """The generated 256 x 32 triangle table (edge numbers, -1 terminated rows)."""
    t = np.empty((256, 32), np.int8)
    _capi.host.tsdf_host_mc_table(t.ctypes.data)
    return t


class ICPOdometry:
    """third_party/ICP_CUDA/ICPOdometry.h of the reference: projective point-to-plane ICP between a model depth image
    (initICPModel) and the current one (initICP), three pyramid levels, 4/5/10 iterations."""

    def __init__(self, width, height, cx, cy, fx, fy, dist_thresh=0.10, angle_thresh=None):
        import math
        if angle_thresh is None:   # sinf(20.f * 3.14159254f / 180.f), ICPOdometry.h:27
            angle_thresh = float(np.float32(math.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))
        self.width, self.height = int(width), int(height)
        self._h = C.c_void_p()
        check(lib.tsdf_icp_create(self.width, self.height, float(cx), float(cy), float(fx), float(fy), float(dist_thresh),
                                  float(angle_thresh), C.byref(self._h)))
        self.last_error, self.last_inliers = 0.0, float(width * height)

    def close(self):
        if lib is not None and getattr(self, "_h", None) is not None and self._h.value:   # (lib is None during interpreter shutdown)
            lib.tsdf_icp_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def set_stream(self, hip_stream):
        check(lib.tsdf_icp_set_stream(self._h, C.c_void_p(int(hip_stream) if hip_stream else 0)))

    def _depth(self, depth):
        d = np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1)
        if d.size != self.width * self.height:
            raise ValueError("depth has %d pixels, expected %d" % (d.size, self.width * self.height))
        return d

    def init_icp(self, depth, depth_cutoff=20.0):
        d = self._depth(depth)
        check(lib.tsdf_icp_init(self._h, 0, d.ctypes.data, float(depth_cutoff)))

    def init_icp_model(self, depth, depth_cutoff=20.0):
        d = self._depth(depth)
        check(lib.tsdf_icp_init(self._h, 1, d.ctypes.data, float(depth_cutoff)))

    def init_icp_device(self, depth_ptr, model=False, depth_cutoff=20.0):
        check(lib.tsdf_icp_init_device(self._h, 1 if model else 0, C.c_void_p(int(depth_ptr)), float(depth_cutoff)))

    def estimate_step(self, level, R, t):
        """R 3x3 (normal indexing), t 3 -> (A 6x6, b 6, residual, inliers) of one Gauss-Newton step at `level`."""
        Rc = np.ascontiguousarray(np.asarray(R, np.float32).T.reshape(-1))   # column-major
        tc = np.ascontiguousarray(t, np.float32).reshape(-1)
        A, b, ri = np.zeros(36, np.float32), np.zeros(6, np.float32), np.zeros(2, np.float32)
        check(lib.tsdf_icp_estimate_step(self._h, int(level), Rc.ctypes.data, tc.ctypes.data, A.ctypes.data, b.ctypes.data,
                                         ri.ctypes.data))
        return A.reshape(6, 6), b, float(ri[0]), float(ri[1])

    def get_incremental_transformation(self, T=None):
        """T_prev_curr (4x4 float64, identity by default) refined in place of the reference's Sophus::SE3d argument."""
        Tc = np.ascontiguousarray((np.eye(4) if T is None else np.asarray(T, np.float64)).T.reshape(-1))
        err, inl = C.c_float(), C.c_float()
        check(lib.tsdf_icp_get_incremental_transformation(self._h, Tc.ctypes.data, C.byref(err), C.byref(inl)))
        self.last_error, self.last_inliers = float(err.value), float(inl.value)
        return Tc.reshape(4, 4).T.copy()

    def get_map(self, which, level):
        """which in vmap_prev | nmap_prev | vmap_curr | nmap_curr -> (3*rows, cols) float32 (planar)."""
        rows, cols = self.height >> level, self.width >> level
        m = np.empty((3 * rows, cols), np.float32)
        check(lib.tsdf_icp_get_map(self._h, {"vmap_prev": 0, "nmap_prev": 1, "vmap_curr": 2, "nmap_curr": 3}[which], int(level),
                                   m.ctypes.data))
        return m

    def get_depth_level(self, level):
        d = np.empty((self.height >> level, self.width >> level), np.uint16)
        check(lib.tsdf_icp_get_depth_level(self._h, int(level), d.ctypes.data))
        return d


class Camera:
    """The C++ Camera of the host library (same surface as src/include/Camera.hpp of the reference).
    Matrices come back as column-major float32 vectors, i.e. what Eigen's .data() yields."""

    def __init__(self, focal_x, focal_y, centre_x, centre_y):
        self._h = _capi.host.tsdf_camera_create(float(focal_x), float(focal_y), float(centre_x), float(centre_y))
        self._refresh()

    @staticmethod
    def default_depth_camera():
        return Camera(591.1, 590.1, 331.0, 234.6)   # Camera.hpp:41-44

    def __del__(self):
        host = getattr(_capi, "host", None) if _capi is not None else None   # (None during interpreter shutdown)
        if getattr(self, "_h", None) and host is not None:
            host.tsdf_camera_destroy(self._h)
            self._h = None

    def _refresh(self):
        self._k, self._kinv = np.zeros(9, np.float32), np.zeros(9, np.float32)
        self._pose, self._ipose = np.zeros(16, np.float32), np.zeros(16, np.float32)
        _capi.host.tsdf_camera_get(self._h, _fp(self._k), _fp(self._kinv), _fp(self._pose), _fp(self._ipose))

    def k(self):
        return self._k

    def kinv(self):
        return self._kinv

    def pose(self):
        return self._pose

    def inverse_pose(self):
        return self._ipose

    def set_pose(self, pose):
        """pose: 16 floats column-major (or a 7-vector tx ty tz qx qy qz qw in TUM order)."""
        p = np.ascontiguousarray(pose, dtype=np.float32).reshape(-1)
        if p.size == 7:
            _capi.host.tsdf_camera_set_pose_tum(self._h, _fp(p))
        else:
            _capi.host.tsdf_camera_set_pose(self._h, _fp(_mat(p, 16)))
        self._refresh()

    def set_pose_rows(self, rows):
        """pose given as a 4x4 in the usual row-major maths notation."""
        self.set_pose(np.ascontiguousarray(np.asarray(rows, np.float32).reshape(4, 4).T).reshape(-1))

    def move_to(self, wx, wy, wz):
        _capi.host.tsdf_camera_move_to(self._h, float(wx), float(wy), float(wz))
        self._refresh()

    def look_at(self, wx, wy, wz):
        _capi.host.tsdf_camera_look_at(self._h, float(wx), float(wy), float(wz))
        self._refresh()

    def position(self):
        return self._pose[12:15].copy()

    def _v3(self, fn, w):
        a = _mat(w, 3)
        out = np.zeros(3, np.float32)
        fn(self._h, _fp(a), _fp(out))
        return out

    def world_to_camera(self, w):
        return self._v3(_capi.host.tsdf_camera_world_to_camera, w)

    def camera_to_world(self, c):
        return self._v3(_capi.host.tsdf_camera_camera_to_world, c)

    def world_to_pixel(self, w):
        out = (C.c_int * 2)()
        _capi.host.tsdf_camera_world_to_pixel(self._h, _fp(_mat(w, 3)), out)
        return int(out[0]), int(out[1])

    def pixel_to_image_plane(self, x, y):
        out = np.zeros(2, np.float32)
        _capi.host.tsdf_camera_pixel_to_image_plane(self._h, int(x), int(y), _fp(out))
        return out

    def image_plane_to_pixel(self, p):
        out = (C.c_int * 2)()
        _capi.host.tsdf_camera_image_plane_to_pixel(self._h, _fp(_mat(p, 2)), out)
        return int(out[0]), int(out[1])


def load_tum_directory(directory):
    """Every frame of a TUM-layout directory through the host library's TUMDataLoader (the loader tools/kinfu_stream.cpp and
    the reference's kinfu.cpp use): [(depth uint16 (H*W,) in millimetres, Camera at the frame's ground-truth pose)], (W, H)."""
    def opened():
        h = _capi.host.tsdf_host_tum_open(str(directory).encode())
        if not h:
            raise ValueError("%s does not have the TUM layout (depth/*.png + ground_truth.txt)" % directory)
        return h

    size, pose = (C.c_uint * 2)(), np.zeros(16, np.float32)
    h = opened()       # the first frame's size (nothing is copied without a buffer)
    got = _capi.host.tsdf_host_tum_next(h, None, 0, size, _fp(pose))
    _capi.host.tsdf_host_tum_close(h)
    if got < 0:
        raise ValueError("%s: the first record's depth image is missing or unreadable" % directory)
    if not got:
        return [], (0, 0)
    w, hh = int(size[0]), int(size[1])
    buf, frames = np.zeros(w * hh, np.uint16), []
    h = opened()
    try:
        while True:
            got = _capi.host.tsdf_host_tum_next(h, buf.ctypes.data, buf.size, size, _fp(pose))
            if got == 0:
                break
            if got < 0:     # (a truncated stream would silently become another workload)
                raise ValueError("%s: record %d of ground_truth.txt has no readable depth image" % (directory, len(frames)))
            if (int(size[0]), int(size[1])) != (w, hh):
                raise ValueError("depth images of different sizes in %s" % directory)
            cam = Camera.default_depth_camera()
            cam.set_pose(pose)
            frames.append((buf.copy(), cam))
    finally:
        _capi.host.tsdf_host_tum_close(h)
    return frames, (w, hh)
