"""Frame-to-model tracking: the kinfu loop with ICP poses instead of ground truth (BASELINE configs[4]).

The reference has the two halves -- src/Tools/kinfu.cpp integrates a TUM sequence with the ground-truth poses,
src/Tools/tsdf_icp.cpp aligns one depth image to a rendering of a volume (GPURaycaster::render_to_depth_image +
ICPOdometry) -- the library composes them per frame, device resident (tsdf_amd/csrc/pipeline.hip: tsdf_tracker_*):

    filtered  = BilateralFilter(depth), initICP(filtered)          (second, lower-priority stream)
    model     = render_to_depth_image(volume, pose[i-1])           (ray cast + vertices_to_depth, beside the line above)
    T         = ICPOdometry(model = model, current = filtered)     (current-camera -> model-camera, metres)
    pose[i]   = pose[i-1] * T                                      (translation back to millimetres; here, with the Camera class)
    volume.integrate(filtered, pose[i])

This class is the ctypes mirror of those entry points plus the pose composition; torch only wraps the tracker's streams.
"""
import ctypes as C

import numpy as np

from . import api
from ._capi import check, lib
from .pipeline import OVERLAP, _matrices


class FrameToModelTracker:
    def __init__(self, volume, width=640, height=480, camera=None, sigma_colour=30.0, sigma_space=4.5, depth_cutoff=20.0, overlap=True):
        import torch
        self.torch = torch
        self.volume = volume
        self.width, self.height = int(width), int(height)
        self.camera = camera or api.Camera.default_depth_camera()
        k = self.camera.k()          # column-major 3x3: fx = k[0], fy = k[4], cx = k[6], cy = k[7]
        self.icp = api.ICPOdometry(self.width, self.height, float(k[6]), float(k[7]), float(k[0]), float(k[4]))
        if not sigma_colour:
            raise ValueError("tracking needs the bilateral filter (raw one-pixel normals fail the ICP angle gate)")
        self.bilateral = api.BilateralFilter(sigma_colour, sigma_space)
        self.depth_cutoff = float(depth_cutoff)
        self._h = C.c_void_p()
        check(lib.tsdf_tracker_create(volume._h, self.bilateral._h, self.icp._h, self.width, self.height, self.depth_cutoff,
                                      OVERLAP if overlap else 0, C.byref(self._h)))
        import weakref
        volume._dependents = list(getattr(volume, "_dependents", ())) + [weakref.ref(self)]
        m, s = C.c_void_p(), C.c_void_p()
        check(lib.tsdf_tracker_streams(self._h, C.byref(m), C.byref(s)))
        self.stream = torch.cuda.ExternalStream(m.value, device=torch.device("cuda", torch.cuda.current_device()))
        self.frames = 0
        self.last_error, self.last_inliers = 0.0, 0.0
        self.last_T = None          # the last incremental transformation (4x4, metres) as ICPOdometry returned it

    def close(self):
        if lib is not None and getattr(self, "_h", None) is not None and self._h.value:   # (lib is None during interpreter shutdown)
            lib.tsdf_tracker_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def pose(self):
        """Current camera pose, 4x4 float64 (camera -> world, millimetres)."""
        return self.camera.pose().astype(np.float64).reshape(4, 4).T.copy()

    def process_device(self, depth_ptr, initial_pose=None):
        """One frame (uint16 millimetres on the device; it must stay valid until the call returns ... and until the frame has been
        filtered: synchronize() or the next call).  The first frame is placed at `initial_pose` (4x4, camera -> world, mm; default:
        the camera's current pose) and only integrated.  Returns the pose used for the frame."""
        check(lib.tsdf_tracker_filter(self._h, C.c_void_p(int(depth_ptr))))
        if self.frames == 0:
            if initial_pose is not None:
                self.camera.set_pose_rows(np.asarray(initial_pose, np.float64))
        else:
            prev = _matrices(self.camera)
            Tc = np.ascontiguousarray(np.eye(4).T.reshape(-1))      # column-major double, identity start
            err, inl = C.c_float(), C.c_float()
            check(lib.tsdf_tracker_align(self._h, C.byref(prev), Tc.ctypes.data, C.byref(err), C.byref(inl)))
            T = Tc.reshape(4, 4).T.copy()                            # current camera -> previous camera, metres
            self.last_T = T.copy()
            T[:3, 3] *= 1000.0
            self.last_error, self.last_inliers = float(err.value), float(inl.value)
            self.icp.last_error, self.icp.last_inliers = self.last_error, self.last_inliers
            self.camera.set_pose_rows(self.pose() @ T)
        cam = _matrices(self.camera)
        check(lib.tsdf_tracker_integrate(self._h, C.byref(cam)))
        self.frames += 1
        return self.pose()

    def synchronize(self):
        check(lib.tsdf_tracker_synchronize(self._h))

    def last_icp_inputs(self):
        """(model depth, filtered current depth) of the last tracked frame as host uint16 arrays: the pair ICPOdometry was
        given, for checking its answer elsewhere."""
        self.synchronize()
        m, f = C.c_void_p(), C.c_void_p()
        check(lib.tsdf_tracker_buffers(self._h, C.byref(m), C.byref(f)))
        n = self.width * self.height
        out = []
        for p in (m, f):
            a = np.empty(n, np.uint16)
            check(lib.tsdf_device_download(a.ctypes.data, p, n * 2))
            out.append(a)
        return tuple(out)

    def process(self, depth, initial_pose=None):
        """Host depth image (uint16 mm)."""
        d = self.torch.from_numpy(np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1).view(np.int16)).cuda()
        self.torch.cuda.current_stream().synchronize()
        pose = self.process_device(d.data_ptr(), initial_pose)
        self.synchronize()           # (the temporary upload is released on return)
        return pose
