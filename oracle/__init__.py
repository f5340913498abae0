"""ctypes binding of the CPU oracle (oracle/libtsdf_oracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by anything under tsdf_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libtsdf_oracle.so")
# The pieces of the reference that compile from their own sources here (oracle/Makefile target "ref"): built into oracle/_ref/
# (git-ignored, not gpurun-ignored: the libraries travel to the GPU box, the reference's sources do not).
REF_BUILD = os.path.join(_HERE, "_ref")
_REF = os.path.join(REF_BUILD, "libref_bilateral.so")
_REF_TRANSFORMS = os.path.join(REF_BUILD, "libref_transforms.so")
_REF_IO = os.path.join(REF_BUILD, "libref_io.so")


def build(force=False):
    """(Re)build the oracle .so (and, when /root/reference is mounted, the reference build under REF_BUILD)."""
    if force or not os.path.exists(_LIB) or not os.path.exists(_FMAD) or \
            os.path.getmtime(_LIB) < max(os.path.getmtime(os.path.join(_HERE, f))
                                         for f in ("tsdf_oracle.c", "icp_oracle.c", "mc_oracle.c", "tsdf_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return _LIB


class Geom(C.Structure):
    """orc_geom (oracle/tsdf_oracle.h)."""
    _fields_ = [("dims", C.c_uint32 * 3), ("phys", C.c_float * 3), ("vs", C.c_float * 3),
                ("offset", C.c_float * 3), ("offset_at_clear", C.c_float * 3), ("trunc", C.c_float)]


class RayStats(C.Structure):
    _fields_ = [("samples", C.c_int64), ("touched", C.c_int64), ("hits", C.c_int64)]


_libs = {}
_variant = "exact"
_FMAD = os.path.join(_HERE, "libtsdf_oracle_fmad.so")


class variant:
    """with oracle.variant("fmad"): ... -- the same restatement compiled with -ffp-contract=fast -mfma, i.e. with every multiply
    feeding an add fused as nvcc's default -fmad=true may fuse them in a real CUDA build of the reference (oracle/Makefile).
    Only for the study of what that difference can flip (tests/test_fmad_sensitivity.py); never the parity oracle."""

    def __init__(self, name):
        assert name in ("exact", "fmad")
        self.name = name

    def __enter__(self):
        global _variant
        self.prev, _variant = _variant, self.name
        return self

    def __exit__(self, *a):
        global _variant
        _variant = self.prev


def lib():
    if _variant not in _libs:
        build()
        L = C.CDLL(_LIB if _variant == "exact" else _FMAD)
        fp = C.POINTER(C.c_float)
        L.orc_geom_init.argtypes = [C.POINTER(Geom), C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float]
        L.orc_clear.argtypes = [fp, fp, C.c_size_t, C.c_float]
        L.orc_voxel_centre.argtypes = [C.POINTER(Geom), C.c_int, C.c_int, C.c_int, fp]
        L.orc_integrate.restype = C.c_int64
        L.orc_integrate.argtypes = [fp, fp, C.POINTER(Geom), fp, fp, fp, C.POINTER(C.c_uint16), C.c_uint32, C.c_uint32,
                                    fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_raycast.argtypes = [fp, C.POINTER(Geom), fp, fp, C.c_uint32, C.c_uint32, fp, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_uint8), C.POINTER(RayStats), C.c_int]
        L.orc_raycast_rows.restype = C.c_int64
        L.orc_raycast_rows.argtypes = [fp, C.POINTER(Geom), fp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, fp, C.c_int]
        L.orc_raycast_slab.argtypes = [fp, C.POINTER(Geom), fp, fp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.POINTER(C.c_uint32), C.c_int]
        L.orc_merge_hits.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(Geom), fp, fp, C.c_uint32, C.c_uint32, fp]
        L.orc_normals.argtypes = [C.c_uint32, C.c_uint32, fp, fp]
        L.orc_ray_box.restype = C.c_int
        L.orc_ray_box.argtypes = [fp, fp, fp, fp, fp, fp]
        L.orc_trilinear.restype = C.c_float
        L.orc_trilinear.argtypes = [fp, C.POINTER(C.c_uint32), fp, fp]
        L.orc_ray_direction.argtypes = [C.c_uint16, C.c_uint16, fp, fp, fp]
        L.orc_world_to_pixel.argtypes = [fp, fp, fp, C.POINTER(C.c_int)]
        L.orc_bilateral_u8.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_bilateral_u16.argtypes = [C.POINTER(C.c_uint16), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        L.orc_bilateral_tables.restype = C.c_int
        L.orc_bilateral_tables.argtypes = [C.c_float, C.c_float, fp, fp, C.c_int]
        L.orc_camera_k.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, fp, fp]
        L.orc_mat3_inverse.argtypes = [fp, fp]
        L.orc_mat4_inverse.argtypes = [fp, fp]
        L.orc_look_at.argtypes = [fp, C.c_float, C.c_float, C.c_float]
        L.orc_camera_world_to_camera.argtypes = [fp, fp, fp]
        L.orc_pixel_to_image_plane.argtypes = [fp, C.c_uint16, C.c_uint16, fp]
        L.orc_image_plane_to_pixel.argtypes = [fp, fp, C.POINTER(C.c_int)]
        L.orc_max_threads.restype = C.c_int
        u16p, dp = C.POINTER(C.c_uint16), C.POINTER(C.c_double)
        L.orc_icp_pyr_down.argtypes = [u16p, C.c_int, C.c_int, u16p]
        L.orc_icp_vmap.argtypes = [u16p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, fp]
        L.orc_icp_nmap.argtypes = [fp, C.c_int, C.c_int, fp]
        L.orc_icp_step.argtypes = [fp, fp, fp, fp, fp, fp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_float, fp, fp, fp, dp]
        L.orc_ldlt_solve6.argtypes = [fp, fp, dp]
        L.orc_se3_exp.argtypes = [dp, dp]
        L.orc_mat4d_mul.argtypes = [dp, dp, dp]
        L.orc_deform_points.argtypes = [C.POINTER(C.c_uint32), fp, fp, fp, fp, fp, fp, C.c_int, fp]
        L.orc_icp_incremental_transformation.argtypes = [u16p, u16p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                                         C.c_float, C.c_float, C.c_float, C.c_float, dp, fp, fp]
        L.orc_mc_tables.argtypes = [C.POINTER(C.c_int8), C.POINTER(C.c_uint8)]
        L.orc_marching_cubes.restype = C.c_int64
        L.orc_marching_cubes.argtypes = [fp, C.c_uint32, C.c_uint32, C.c_uint32, fp, fp, fp, C.c_int64, C.c_int]
        _libs[_variant] = L
    return _libs[_variant]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def max_threads():
    return lib().orc_max_threads()


# ------------------------------------------------------------------------------ camera

def camera_k(fx=591.1, fy=590.1, cx=331.0, cy=234.6):
    """K and Kinv, column-major 9-vectors (default = Camera::default_depth_camera, Camera.hpp:41-44)."""
    k = np.zeros(9, np.float32)
    kinv = np.zeros(9, np.float32)
    lib().orc_camera_k(fx, fy, cx, cy, _fp(k), _fp(kinv))
    return k, kinv


def mat4_inverse(m):
    m = _f32(m, 16)
    o = np.zeros(16, np.float32)
    lib().orc_mat4_inverse(_fp(m), _fp(o))
    return o


def mat3_inverse(m):
    m = _f32(m, 9)
    o = np.zeros(9, np.float32)
    lib().orc_mat3_inverse(_fp(m), _fp(o))
    return o


def look_at(pose, target):
    p = _f32(pose, 16).copy()
    lib().orc_look_at(_fp(p), float(target[0]), float(target[1]), float(target[2]))
    return p


def identity_pose(position=(0, 0, 0)):
    p = np.eye(4, dtype=np.float32).T.reshape(-1).copy()  # column-major
    p[12:15] = np.asarray(position, np.float32)
    return p


def pose_from_rows(rows):
    """4x4 given in row-major maths notation -> column-major 16-vector."""
    return np.ascontiguousarray(np.asarray(rows, np.float32).reshape(4, 4).T).reshape(-1)


def world_to_camera(inv_pose, w):
    ip = _f32(inv_pose, 16)
    w = _f32(w, 3)
    c = np.zeros(3, np.float32)
    lib().orc_camera_world_to_camera(_fp(ip), _fp(w), _fp(c))
    return c


def pixel_to_image_plane(kinv, x, y):
    out = np.zeros(2, np.float32)
    lib().orc_pixel_to_image_plane(_fp(_f32(kinv, 9)), x, y, _fp(out))
    return out


def image_plane_to_pixel(k, cam):
    out = (C.c_int * 2)()
    lib().orc_image_plane_to_pixel(_fp(_f32(k, 9)), _fp(_f32(cam, 2)), out)
    return int(out[0]), int(out[1])


# ------------------------------------------------------------------------------ volume

class Volume:
    """Host-memory TSDF volume driven by the oracle; mirrors the reference's TSDFVolume state
    (src/include/TSDFVolume.hpp:269-303).  Optionally a Z-slab [z_store_begin, z_store_end)."""

    def __init__(self, size, physical_size, z_store=None):
        X, Y, Z = (int(s) for s in size)
        if min(X, Y, Z) <= 0 or min(physical_size) <= 0:
            raise ValueError("Attempt to construct TSDFVolume with zero or negative size")
        self.g = Geom()
        lib().orc_geom_init(C.byref(self.g), X, Y, Z, *[float(p) for p in physical_size])
        self.z0, self.z1 = (0, Z) if z_store is None else z_store
        n = X * Y * (self.z1 - self.z0)
        self.dist = np.empty(n, np.float32)
        self.weight = np.empty(n, np.float32)
        self.translation = None
        self.clear()

    # accessors named like the reference's
    def size(self):
        return tuple(self.g.dims)

    def voxel_size(self):
        return np.array(self.g.vs, np.float32)

    def physical_size(self):
        return np.array(self.g.phys, np.float32)

    def truncation_distance(self):
        return float(self.g.trunc)

    def offset(self, *o):
        if o:
            for i in range(3):
                self.g.offset[i] = float(o[i])   # setter does NOT re-init nodes (TSDFVolume.hpp:139-143)
            return None
        return np.array(self.g.offset, np.float32)

    def clear(self):
        lib().orc_clear(_fp(self.dist), _fp(self.weight), self.dist.size, self.g.trunc)
        for i in range(3):
            self.g.offset_at_clear[i] = self.g.offset[i]  # initialise_deformation bakes m_offset in

    def voxel_centre(self, x, y, z):
        out = np.zeros(3, np.float32)
        lib().orc_voxel_centre(C.byref(self.g), x, y, z, _fp(out))
        return out

    def set_distance_data(self, d):
        self.dist[:] = _f32(d, self.dist.size)

    def set_weight_data(self, w):
        self.weight[:] = _f32(w, self.weight.size)

    def integrate(self, depth, width, height, inv_pose, k, kinv, z_range=None, nthreads=1):
        depth = np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1)
        assert depth.size == width * height
        zb, ze = (self.z0, self.z1) if z_range is None else z_range
        tr = None if self.translation is None else _fp(self.translation)
        return lib().orc_integrate(_fp(self.dist), _fp(self.weight), C.byref(self.g), _fp(_f32(inv_pose, 16)),
                                   _fp(_f32(k, 9)), _fp(_f32(kinv, 9)),
                                   depth.ctypes.data_as(C.POINTER(C.c_uint16)), width, height, tr,
                                   self.z0, zb, ze, nthreads)

    def raycast(self, width, height, pose, kinv, nthreads=1, stats=False):
        assert self.z0 == 0 and self.z1 == self.g.dims[2]
        V = np.empty(width * height * 3, np.float32)
        sc = tm = st = None
        if stats:
            sc = np.zeros(width * height, np.int32)
            tm = np.zeros(self.dist.size, np.uint8)
            st = RayStats()
        lib().orc_raycast(_fp(self.dist), C.byref(self.g), _fp(_f32(pose, 16)), _fp(_f32(kinv, 9)), width, height,
                          _fp(V), None if sc is None else sc.ctypes.data_as(C.POINTER(C.c_int32)),
                          None if tm is None else tm.ctypes.data_as(C.POINTER(C.c_uint8)),
                          None if st is None else C.byref(st), nthreads)
        N = np.empty_like(V)
        lib().orc_normals(width, height, _fp(V), _fp(N))
        V = V.reshape(-1, 3)
        N = N.reshape(-1, 3)
        if stats:
            return V, N, {"samples": st.samples, "touched": st.touched, "hits": st.hits, "sample_count": sc}
        return V, N

    def raycast_rows(self, width, height, pose, kinv, y_begin, y_end, y_step, nthreads=1):
        """Bounded sample: only rows y_begin::y_step; returns (vertices with untouched rows = NaN, samples)."""
        V = np.full(width * height * 3, np.nan, np.float32)
        s = lib().orc_raycast_rows(_fp(self.dist), C.byref(self.g), _fp(_f32(pose, 16)), _fp(_f32(kinv, 9)), width,
                                   height, y_begin, y_end, y_step, _fp(V), nthreads)
        return V.reshape(-1, 3), int(s)

    def raycast_slab(self, width, height, pose, kinv, own, nthreads=1):
        """(W*H, 2) uint32 records {k, bits of t}: the first owned sample <= 0 (0xffffffff: none), its refined parameter."""
        hits = np.empty(width * height * 2, np.uint32)
        lib().orc_raycast_slab(_fp(self.dist), C.byref(self.g), _fp(_f32(pose, 16)), _fp(_f32(kinv, 9)), width,
                               height, self.z0, own[0], own[1], hits.ctypes.data_as(C.POINTER(C.c_uint32)), nthreads)
        return hits.reshape(-1, 2)

    def merge_hits(self, hits_all, width, height, pose, kinv):
        """Min-k select over (n_slabs, W*H, 2) records and the vertices the reference forms from the refined parameter."""
        h = np.ascontiguousarray(hits_all).view(np.uint32).reshape(-1, width * height, 2)
        V = np.empty(width * height * 3, np.float32)
        lib().orc_merge_hits(h.ctypes.data_as(C.POINTER(C.c_uint32)), h.shape[0], C.byref(self.g), _fp(_f32(pose, 16)),
                             _fp(_f32(kinv, 9)), width, height, _fp(V))
        return V.reshape(-1, 3)


def normals(width, height, V):
    V = _f32(V, width * height * 3)
    N = np.empty_like(V)
    lib().orc_normals(width, height, _fp(V), _fp(N))
    return N.reshape(-1, 3)


def ray_box(origin, direction, smin, smax):
    n = C.c_float()
    f = C.c_float()
    r = lib().orc_ray_box(_fp(_f32(origin, 3)), _fp(_f32(direction, 3)), _fp(_f32(smin, 3)), _fp(_f32(smax, 3)),
                          C.byref(n), C.byref(f))
    return bool(r), n.value, f.value


def trilinear(point, dims, vs, dist):
    d = (C.c_uint32 * 3)(*[int(x) for x in dims])
    return lib().orc_trilinear(_fp(_f32(point, 3)), d, _fp(_f32(vs, 3)), _fp(_f32(dist)))


def world_to_pixel(p, inv_pose, k):
    out = (C.c_int * 2)()
    lib().orc_world_to_pixel(_fp(_f32(p, 3)), _fp(_f32(inv_pose, 16)), _fp(_f32(k, 9)), out)
    return int(out[0]), int(out[1])


def world_to_pixel_n(points, inv_pose, k):
    p = _f32(points).reshape(-1, 3)
    out = np.empty((len(p), 2), np.int32)
    lib().orc_world_to_pixel_n(C.c_size_t(len(p)), _fp(p), _fp(_f32(inv_pose, 16)), _fp(_f32(k, 9)), out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


def world_to_camera_n(points, inv_pose):
    p = _f32(points).reshape(-1, 3)
    out = np.empty((len(p), 3), np.float32)
    lib().orc_world_to_camera_n(C.c_size_t(len(p)), _fp(p), _fp(_f32(inv_pose, 16)), _fp(out))
    return out


def pixel_to_camera_n(pixels, depth, kinv):
    px = np.ascontiguousarray(pixels, np.int32).reshape(-1, 2)
    out = np.empty((len(px), 3), np.float32)
    lib().orc_pixel_to_camera_n(C.c_size_t(len(px)), px.ctypes.data_as(C.POINTER(C.c_int)), _fp(_f32(depth, len(px))), _fp(_f32(kinv, 9)), _fp(out))
    return out


def ray_direction_n(pixels, rot, kinv):
    px = np.ascontiguousarray(pixels, np.uint16).reshape(-1, 2)
    out = np.empty((len(px), 3), np.float32)
    lib().orc_ray_direction_n(C.c_size_t(len(px)), px.ctypes.data_as(C.POINTER(C.c_uint16)), _fp(_f32(rot, 9)), _fp(_f32(kinv, 9)), _fp(out))
    return out


# ------------------------------------------------------------------------------ bilateral

def bilateral_u8(image, width, height, sigma_colour, sigma_space):
    img = np.ascontiguousarray(image, np.uint8).reshape(-1).copy()
    lib().orc_bilateral_u8(img.ctypes.data_as(C.POINTER(C.c_uint8)), width, height, sigma_colour, sigma_space)
    return img.reshape(height, width)


def bilateral_u16(image, width, height, sigma_colour, sigma_space, nthreads=1):
    img = np.ascontiguousarray(image, np.uint16).reshape(-1).copy()
    lib().orc_bilateral_u16(img.ctypes.data_as(C.POINTER(C.c_uint16)), width, height, sigma_colour, sigma_space,
                            nthreads)
    return img.reshape(height, width)


def bilateral_tables(sigma_colour, sigma_space, n_similarity=256):
    r = lib().orc_bilateral_tables(sigma_colour, sigma_space, None, None, 0)
    k = np.zeros((2 * r + 1) ** 2, np.float32)
    s = np.zeros(n_similarity, np.float32)
    lib().orc_bilateral_tables(sigma_colour, sigma_space, _fp(k), _fp(s), n_similarity)
    return r, k, s


def have_ref():
    return os.path.exists(_REF)


def ref_bilateral_u8(image, width, height, sigma_colour, sigma_space):
    """The reference's own BilateralFilter (oracle/_ref/libref_bilateral.so, built from /root/reference/src/BilateralFilter.cpp)."""
    L = C.CDLL(_REF, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_DEEPBIND", 0))
    L.ref_bilateral_u8.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_float, C.c_float]
    img = np.ascontiguousarray(image, np.uint8).reshape(-1).copy()
    L.ref_bilateral_u8(img.ctypes.data_as(C.POINTER(C.c_uint8)), width, height, sigma_colour, sigma_space)
    return img.reshape(height, width)


def have_ref_transforms():
    return os.path.exists(_REF_TRANSFORMS)


_ref_t = None


def _ref_transforms():
    """oracle/_ref/libref_transforms.so: the reference's own src/Utilities/cuda_coordinate_transforms.cu compiled with g++ against the
    CUDA toolkit headers of this image (oracle/ref_transforms_wrap.cpp has the entry points)."""
    global _ref_t
    if _ref_t is None:
        L = C.CDLL(_REF_TRANSFORMS, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_DEEPBIND", 0))
        fp, ip, up = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint16)
        L.ref_world_to_pixel.argtypes = [C.c_size_t, fp, fp, fp, ip]
        L.ref_world_to_camera.argtypes = [C.c_size_t, fp, fp, fp]
        L.ref_pixel_to_camera.argtypes = [C.c_size_t, ip, fp, fp, fp]
        L.ref_ray_direction.argtypes = [C.c_size_t, up, fp, fp, fp]
        L.ref_integrate_composed.restype = C.c_int64
        L.ref_integrate_composed.argtypes = [fp, fp, C.c_uint32, C.c_uint32, C.c_uint32, fp, fp, fp, C.c_float, fp, fp, fp, up,
                                             C.c_uint32, C.c_uint32]
        _ref_t = L
    return _ref_t


def ref_world_to_pixel(points, inv_pose, k):
    p = _f32(points).reshape(-1, 3)
    out = np.empty((len(p), 2), np.int32)
    _ref_transforms().ref_world_to_pixel(len(p), _fp(p), _fp(_f32(inv_pose, 16)), _fp(_f32(k, 9)), out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def ref_world_to_camera(points, inv_pose):
    p = _f32(points).reshape(-1, 3)
    out = np.empty((len(p), 3), np.float32)
    _ref_transforms().ref_world_to_camera(len(p), _fp(p), _fp(_f32(inv_pose, 16)), _fp(out))
    return out


def ref_pixel_to_camera(pixels, depth, kinv):
    px = np.ascontiguousarray(pixels, np.int32).reshape(-1, 2)
    d = _f32(depth, len(px))
    out = np.empty((len(px), 3), np.float32)
    _ref_transforms().ref_pixel_to_camera(len(px), px.ctypes.data_as(C.POINTER(C.c_int32)), _fp(d), _fp(_f32(kinv, 9)), _fp(out))
    return out


def ref_ray_direction(pixels, rot, kinv):
    px = np.ascontiguousarray(pixels, np.uint16).reshape(-1, 2)
    out = np.empty((len(px), 3), np.float32)
    _ref_transforms().ref_ray_direction(len(px), px.ctypes.data_as(C.POINTER(C.c_uint16)), _fp(_f32(rot, 9)), _fp(_f32(kinv, 9)), _fp(out))
    return out


def ref_integrate_composed(dist, weight, size, voxel_size, trunc, inv_pose, k, kinv, depth, width, height,
                           offset_at_clear=(0.0, 0.0, 0.0), offset_now=(0.0, 0.0, 0.0)):
    """The loop of integrate_kernel around the reference's own compiled world_to_pixel / pixel_to_camera / world_to_camera
    (oracle/ref_transforms_wrap.cpp); dist / weight (float32, C order z, y, x) are updated in place.  Returns the voxels updated."""
    assert dist.dtype == np.float32 and weight.dtype == np.float32 and dist.flags.c_contiguous and weight.flags.c_contiguous
    d = np.ascontiguousarray(depth, np.uint16).reshape(-1)
    return int(_ref_transforms().ref_integrate_composed(_fp(dist), _fp(weight), int(size[0]), int(size[1]), int(size[2]), _fp(_f32(voxel_size, 3)),
                                                        _fp(_f32(offset_at_clear, 3)), _fp(_f32(offset_now, 3)), float(trunc),
                                                        _fp(_f32(inv_pose, 16)), _fp(_f32(k, 9)), _fp(_f32(kinv, 9)),
                                                        d.ctypes.data_as(C.POINTER(C.c_uint16)), int(width), int(height)))


def have_ref_io():
    return os.path.exists(_REF_IO)


_ref_io_lib = None


def _ref_io():
    """oracle/_ref/libref_io.so: the reference's src/Utilities/ply.cpp and PgmUtilities.cpp compiled where they lie
    (oracle/ref_io_wrap.cpp has the entry points)."""
    global _ref_io_lib
    if _ref_io_lib is None:
        L = C.CDLL(_REF_IO, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_DEEPBIND", 0))
        L.ref_write_to_ply.restype = None
        L.ref_write_to_ply.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_int), C.c_size_t]
        L.ref_read_pgm.restype = C.c_size_t
        L.ref_read_pgm.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16), C.c_size_t]
        file_utilities_signatures(L, "ref")
        _ref_io_lib = L
    return _ref_io_lib


def file_utilities_signatures(L, prefix):
    """ctypes signatures of the FileUtilities entry points, the same on both sides (oracle/ref_io_wrap.cpp around the reference's
    file, tsdf_amd/host/src/host_capi.cpp around the host library's)."""
    ip = C.POINTER(C.c_int)
    f = lambda name: getattr(L, prefix + "_" + name)
    f("match_file_name").restype = C.c_int
    f("match_file_name").argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    f("process_file_by_lines").restype = C.c_size_t
    f("process_file_by_lines").argtypes = [C.c_char_p, ip, C.c_char_p, C.c_size_t]
    f("read_last_line").restype = C.c_size_t
    f("read_last_line").argtypes = [C.c_char_p, C.c_char_p, ip, C.c_char_p, C.c_size_t]
    f("files_in_directory").restype = C.c_size_t
    f("files_in_directory").argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
    f("file_exists").restype = C.c_int
    f("file_exists").argtypes = [C.c_char_p, ip]


class FileUtilities:
    """The five helpers of FileUtilities through one of the two libraries (L, prefix): strings in, Python values out."""

    def __init__(self, L, prefix):
        self.f = lambda name: getattr(L, prefix + "_" + name)

    def _text(self, call):
        n = call(None, 0)
        buf = C.create_string_buffer(n + 1)
        call(buf, n + 1)
        return buf.raw[:n]

    def match_file_name(self, prefix, num_digits, suffix, extension, test_string):
        return bool(self.f("match_file_name")(prefix, num_digits, suffix, extension, test_string))

    def process_file_by_lines(self, path):
        ok = C.c_int(-1)
        joined = self._text(lambda out, cap: self.f("process_file_by_lines")(str(path).encode(), C.byref(ok), out, cap))
        return bool(ok.value), joined.split(b"\x1e")[:-1]

    def read_last_line(self, path, preset=b"untouched"):
        ok = C.c_int(-1)
        text = self._text(lambda out, cap: self.f("read_last_line")(str(path).encode(), preset, C.byref(ok), out, cap))
        return bool(ok.value), text

    def files_in_directory(self, directory, prefix, num_digits, suffix, extension):
        joined = self._text(lambda out, cap: self.f("files_in_directory")(str(directory).encode(), prefix, num_digits, suffix, extension, out, cap))
        return joined.split(b"\x1e")[:-1]

    def file_exists(self, path, preset):
        d = C.c_int(1 if preset else 0)
        e = self.f("file_exists")(str(path).encode(), C.byref(d))
        return bool(e), bool(d.value)


def ref_file_utilities():
    """The reference's src/Utilities/FileUtilities.cpp, compiled where it lies."""
    return FileUtilities(_ref_io(), "ref")


def ref_write_to_ply(path, vertices, triangles):
    """The reference's write_to_ply (src/Utilities/ply.cpp:6-30) on (n, 3) float32 vertices and (m, 3) int32 triangles."""
    v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(triangles, np.int32).reshape(-1, 3)
    _ref_io().ref_write_to_ply(str(path).encode(), _fp(v), len(v), t.ctypes.data_as(C.POINTER(C.c_int)), len(t))


def ref_read_pgm(path):
    """The reference's read_pgm (src/Utilities/PgmUtilities.cpp:49-85) -> (height, width) uint16, as read (before the NYU byte swap of
    src/Utilities/DepthMapUtilities.cpp:20-33, which the caller applies)."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    n = _ref_io().ref_read_pgm(str(path).encode(), C.byref(w), C.byref(h), None, 0)
    out = np.empty(n, np.uint16)
    _ref_io().ref_read_pgm(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_uint16)), n)
    return out.reshape(h.value, w.value)


# ---- ICP tracking (icp_oracle.c) -------------------------------------------------------------------------------
def _u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def icp_pyr_down(depth, rows, cols):
    src = _u16(depth).reshape(rows, cols)
    dst = np.empty((rows // 2, cols // 2), np.uint16)
    lib().orc_icp_pyr_down(src.ctypes.data_as(C.POINTER(C.c_uint16)), rows, cols, dst.ctypes.data_as(C.POINTER(C.c_uint16)))
    return dst


def icp_vmap(depth, rows, cols, fx, fy, cx, cy, depth_cutoff=20.0):
    """-> (3*rows, cols) float32, planar; never-written components are 0."""
    d = _u16(depth).reshape(rows, cols)
    v = np.zeros((3 * rows, cols), np.float32)
    lib().orc_icp_vmap(d.ctypes.data_as(C.POINTER(C.c_uint16)), rows, cols, fx, fy, cx, cy, depth_cutoff, _fp(v))
    return v


def icp_nmap(vmap, rows, cols):
    v = _f32(vmap, 3 * rows * cols)
    n = np.zeros((3 * rows, cols), np.float32)
    lib().orc_icp_nmap(_fp(v), rows, cols, _fp(n))
    return n


def icp_step(R, t, vmap_curr, nmap_curr, vmap_prev, nmap_prev, rows, cols, fx, fy, cx, cy, dist_thresh, angle_thresh):
    """R: 3x3 column-major, t: 3 -> (A 6x6, b 6, residual, inliers, sums29 float64)."""
    R, t = _f32(R, 9), _f32(t, 3)
    arrs = [_f32(a, 3 * rows * cols) for a in (vmap_curr, nmap_curr, vmap_prev, nmap_prev)]
    A, b, ri, sums = np.zeros(36, np.float32), np.zeros(6, np.float32), np.zeros(2, np.float32), np.zeros(29, np.float64)
    lib().orc_icp_step(_fp(R), _fp(t), *[_fp(a) for a in arrs], rows, cols, fx, fy, cx, cy, dist_thresh, angle_thresh,
                       _fp(A), _fp(b), _fp(ri), _dp(sums))
    return A.reshape(6, 6), b, float(ri[0]), float(ri[1]), sums


def ldlt_solve6(A, b):
    A, b = _f32(A, 36), _f32(b, 6)
    x = np.zeros(6, np.float64)
    lib().orc_ldlt_solve6(_fp(A), _fp(b), _dp(x))
    return x


def se3_exp(a):
    """a = (upsilon, omega) -> 4x4 (row/col indexed as a normal matrix)."""
    a = np.ascontiguousarray(a, np.float64)
    T = np.zeros(16, np.float64)
    lib().orc_se3_exp(_dp(a), _dp(T))
    return T.reshape(4, 4).T.copy()


def icp_incremental_transformation(depth_curr, depth_model, width, height, cx, cy, fx, fy, dist_thresh=0.10,
                                   angle_thresh=None, depth_cutoff=20.0, T=None):
    """ICPOdometry::initICP(depth_curr) + initICPModel(depth_model) + getIncrementalTransformation(T).
    -> (T 4x4 float64, last_error, last_inliers)."""
    import math
    if angle_thresh is None:
        angle_thresh = float(np.float32(math.sin(np.float32(20.0) * np.float32(3.14159254) / np.float32(180.0))))
    Tc = np.ascontiguousarray((np.eye(4) if T is None else np.asarray(T, np.float64)).T.reshape(-1))   # column-major
    err, inl = C.c_float(), C.c_float()
    a, b = _u16(depth_curr).reshape(-1), _u16(depth_model).reshape(-1)
    lib().orc_icp_incremental_transformation(a.ctypes.data_as(C.POINTER(C.c_uint16)), b.ctypes.data_as(C.POINTER(C.c_uint16)),
                                             width, height, cx, cy, fx, fy, dist_thresh, angle_thresh, depth_cutoff, _dp(Tc),
                                             C.byref(err), C.byref(inl))
    return Tc.reshape(4, 4).T.copy(), float(err.value), float(inl.value)


def deform_points(dims, vs, offset, offset_at_clear, nodes, global_rotation, global_translation, points):
    """orc_deform_points: points (n,3) float32 -> deformed copy.  nodes: (N,6) float32 or None."""
    d = np.ascontiguousarray(dims, np.uint32)
    a = [_f32(x, 3) for x in (vs, offset, offset_at_clear)]
    r, t = _f32(global_rotation, 3), _f32(global_translation, 3)
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3).copy()
    nd = None if nodes is None else np.ascontiguousarray(nodes, np.float32).reshape(-1)
    lib().orc_deform_points(d.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(a[0]), _fp(a[1]), _fp(a[2]),
                            _fp(nd) if nd is not None else None, _fp(r), _fp(t), pts.shape[0], _fp(pts))
    return pts


# ------------------------------------------------------------------------------ marching cubes (mc_oracle.c)

def mc_tables():
    """(TRIANGLE_TABLE 256 x 16 int8, VERTICES_FOR_CUBE_TYPE 256 uint8) as the oracle builds them."""
    t = np.empty((256, 16), np.int8)
    c = np.empty(256, np.uint8)
    lib().orc_mc_tables(t.ctypes.data_as(C.POINTER(C.c_int8)), c.ctypes.data_as(C.POINTER(C.c_uint8)))
    return t, c


def marching_cubes(distances, size, voxel_size, offset=(0.0, 0.0, 0.0), nthreads=1):
    """extract_surface of the reference (MarkAndSweepMC.cu:506-555) on a host distance array: (n, 3) float32 vertices, three
    per triangle in emission order (the reference wires triangle t as (3t, 3t+2, 3t+1))."""
    X, Y, Z = (int(v) for v in size)
    d = _f32(distances, X * Y * Z)
    vs, off = _f32(voxel_size, 3), _f32(offset, 3)
    n = lib().orc_marching_cubes(_fp(d), X, Y, Z, _fp(vs), _fp(off), None, 0, nthreads)
    out = np.empty((max(n, 0), 3), np.float32)
    if n > 0:
        got = lib().orc_marching_cubes(_fp(d), X, Y, Z, _fp(vs), _fp(off), _fp(out), n, nthreads)
        assert got == n
    return out
