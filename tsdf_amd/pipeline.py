"""The per-frame step of BASELINE configs[2] -- bilateral filter, integrate, ray cast + normals, frame after frame as the
reference's kinfu loop integrates them (src/Tools/kinfu.cpp:32-56) -- on frames that live in HBM.

The schedule itself is C++ behind the C ABI (tsdf_amd/csrc/pipeline.hip: tsdf_pipeline_create / _step / _destroy -- two HIP
streams with priorities, four reusable events, the next frame's filter and brick culling ahead on the lower-priority stream);
this class is its ctypes mirror for the tests and bench.py, as tools/kinfu_stream.cpp is its C++ caller.  torch appears only to
wrap the pipeline's streams (torch.cuda.ExternalStream) so that the caller's tensors can be ordered against them."""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import check, lib
from .api import _camera_matrices

OVERLAP, EQUAL_PRIORITY, EXCHANGE_STREAM, NO_TIGHTEN_AHEAD = 1, 2, 4, 8     # TSDF_PIPELINE_* (include/tsdf_amd.h)


def _matrices(camera):
    m = _capi.CameraMatrices()
    pose, ipose, k, kinv = _camera_matrices(camera)
    C.memmove(m.pose, pose.ctypes.data, 64)
    C.memmove(m.inv_pose, ipose.ctypes.data, 64)
    C.memmove(m.k, k.ctypes.data, 36)
    C.memmove(m.kinv, kinv.ctypes.data, 36)
    return m


class FusionPipeline:
    """step(depth_ptr, camera, vertices_ptr, normals_ptr, next_depth_ptr=None, next_camera=None): one frame through filter ->
    integrate -> raycast (+ normals).  `next_depth_ptr`, when given, is the device pointer of the frame the next call will pass:
    its filter is queued now (and its brick culling, when `next_camera` is given).  All pointers are device pointers to
    width * height uint16 (depth) / 3 * width * height float32 (maps); the depth buffers must stay valid until the frame after
    them has been processed.  `exchange`: a tsdf_amd.multi.SlabExchange when the volume is one rank's Z-slab."""

    def __init__(self, volume, bilateral, raycaster, width, height, overlap=True, exchange=None, equal_priority=False, exchange_stream=False,
                 tighten_ahead=True):
        self.volume, self.bilateral, self.raycaster = volume, bilateral, raycaster
        self.width, self.height = int(width), int(height)
        self.overlap = bool(overlap)
        self.exchange = exchange
        flags = (OVERLAP if overlap else 0) | (EQUAL_PRIORITY if equal_priority else 0) | (EXCHANGE_STREAM if exchange_stream else 0) | \
            (0 if tighten_ahead else NO_TIGHTEN_AHEAD)
        self._h = C.c_void_p()
        check(lib.tsdf_pipeline_create(volume._h, bilateral._h, self.width, self.height, flags, exchange._h if exchange is not None else None,
                                       C.byref(self._h)))
        import weakref
        volume._dependents = list(getattr(volume, "_dependents", ())) + [weakref.ref(self)]
        m, s = C.c_void_p(), C.c_void_p()
        check(lib.tsdf_pipeline_streams(self._h, C.byref(m), C.byref(s)))
        dev = torch.device("cuda", torch.cuda.current_device())
        self.main = torch.cuda.ExternalStream(m.value, device=dev)
        self.side = torch.cuda.ExternalStream(s.value, device=dev) if s.value else None

    def hit_buffers(self):
        """Device pointers (this rank's records, all ranks' records) of a sharded pipeline."""
        a, b = C.c_void_p(), C.c_void_p()
        check(lib.tsdf_pipeline_hit_buffers(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def step(self, depth_ptr, camera, vertices_ptr, normals_ptr=None, next_depth_ptr=None, next_camera=None):
        cam = _matrices(camera)
        nxt = _matrices(next_camera) if (next_camera is not None and next_depth_ptr is not None) else None
        check(lib.tsdf_pipeline_step(self._h, C.c_void_p(int(depth_ptr)), C.byref(cam), C.c_void_p(int(vertices_ptr)),
                                     C.c_void_p(int(normals_ptr)) if normals_ptr else None,
                                     C.c_void_p(int(next_depth_ptr)) if next_depth_ptr else None,
                                     C.byref(nxt) if nxt is not None else None))

    def synchronize(self):
        check(lib.tsdf_pipeline_synchronize(self._h))

    def close(self):
        if lib is not None and getattr(self, "_h", None) is not None and self._h.value:   # (lib is None during interpreter shutdown)
            lib.tsdf_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close
