"""Frame-to-model tracking: the kinfu loop with ICP poses instead of ground truth (BASELINE configs[4]).

The reference has the two halves -- src/Tools/kinfu.cpp integrates a TUM sequence with the ground-truth poses,
src/Tools/tsdf_icp.cpp aligns one depth image to a rendering of a volume (GPURaycaster::render_to_depth_image +
ICPOdometry) -- this module composes them per frame, device resident:

    filtered  = BilateralFilter(depth)
    model     = render_to_depth_image(volume, pose[i-1])           (ray cast + vertices_to_depth)
    T         = ICPOdometry(model = model, current = filtered)     (current-camera -> model-camera, metres)
    pose[i]   = pose[i-1] * T                                      (translation back to millimetres)
    volume.integrate(filtered, pose[i])

Everything numeric is done by the HIP library through its C ABI; this file is host glue (torch only owns the device
buffers and the stream).
"""
import numpy as np

from . import api


class FrameToModelTracker:
    def __init__(self, volume, width=640, height=480, camera=None, sigma_colour=30.0, sigma_space=4.5, depth_cutoff=20.0):
        import torch
        self.torch = torch
        self.volume = volume
        self.width, self.height = int(width), int(height)
        self.camera = camera or api.Camera.default_depth_camera()
        k = self.camera.k()          # column-major 3x3: fx = k[0], fy = k[4], cx = k[6], cy = k[7]
        self.icp = api.ICPOdometry(self.width, self.height, float(k[6]), float(k[7]), float(k[0]), float(k[4]))
        self.bilateral = api.BilateralFilter(sigma_colour, sigma_space) if sigma_colour else None
        self.raycaster = api.GPURaycaster(self.width, self.height)
        self.depth_cutoff = float(depth_cutoff)
        n = self.width * self.height
        self._filtered = torch.empty((n,), dtype=torch.int16, device="cuda")
        # the filter leaves the 16 x 16 tile maxima of its output for integrate's brick culling (one launch fewer per frame)
        self._tile_max = torch.empty((((self.width + 15) // 16) * ((self.height + 15) // 16),), dtype=torch.int16, device="cuda")
        self._model = torch.empty((n,), dtype=torch.int16, device="cuda")
        self._vertices = torch.empty((n, 3), dtype=torch.float32, device="cuda")
        self.stream = torch.cuda.current_stream()
        self.volume.set_stream(self.stream.cuda_stream)
        self.icp.set_stream(self.stream.cuda_stream)
        self.frames = 0
        self.last_error, self.last_inliers = 0.0, 0.0
        self.last_T = None          # the last incremental transformation (4x4, metres) as ICPOdometry returned it

    def pose(self):
        """Current camera pose, 4x4 float64 (camera -> world, millimetres)."""
        return self.camera.pose().astype(np.float64).reshape(4, 4).T.copy()

    def _filter(self, depth_ptr):
        s = self.stream.cuda_stream
        if self.bilateral is None:
            self.torch.cuda.synchronize()
            raise ValueError("tracking needs the bilateral filter (raw one-pixel normals fail the ICP angle gate)")
        self.bilateral.filter_device(depth_ptr, self._filtered.data_ptr(), self.width, self.height, bits=16, stream=s,
                                     tile_max_ptr=self._tile_max.data_ptr())

    def process_device(self, depth_ptr, initial_pose=None):
        """One frame (uint16 millimetres on the device).  The first frame is placed at `initial_pose` (4x4, camera ->
        world, mm; default: the camera's current pose) and only integrated.  Returns the pose used for the frame."""
        W, H, s = self.width, self.height, self.stream.cuda_stream
        self._filter(depth_ptr)
        if self.frames == 0:
            if initial_pose is not None:
                self.camera.set_pose_rows(np.asarray(initial_pose, np.float64))
        else:
            # model image: the volume rendered from the previous pose
            self.raycaster.raycast_device(self.volume, self.camera, self._vertices.data_ptr(), None)
            api.vertices_to_depth_device(W, H, self._vertices.data_ptr(), self.camera, self._model.data_ptr(), s)
            self.icp.init_icp_device(self._model.data_ptr(), model=True, depth_cutoff=self.depth_cutoff)
            self.icp.init_icp_device(self._filtered.data_ptr(), depth_cutoff=self.depth_cutoff)
            T = self.icp.get_incremental_transformation()      # current camera -> previous camera, metres
            self.last_T = T.copy()
            T[:3, 3] *= 1000.0
            self.last_error, self.last_inliers = self.icp.last_error, self.icp.last_inliers
            self.camera.set_pose_rows(self.pose() @ T)
        self.volume.integrate_device(self._filtered.data_ptr(), W, H, self.camera, tile_max_ptr=self._tile_max.data_ptr())
        self.frames += 1
        return self.pose()

    def last_icp_inputs(self):
        """(model depth, filtered current depth) of the last tracked frame as host uint16 arrays: the pair ICPOdometry was
        given, for checking its answer elsewhere."""
        self.torch.cuda.synchronize()
        return (self._model.cpu().numpy().view(np.uint16).copy(), self._filtered.cpu().numpy().view(np.uint16).copy())

    def process(self, depth, initial_pose=None):
        """Host depth image (uint16 mm)."""
        d = self.torch.from_numpy(np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1).view(np.int16)).cuda()
        return self.process_device(d.data_ptr(), initial_pose)
