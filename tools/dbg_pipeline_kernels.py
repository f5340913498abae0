"""Where a pipelined step goes: the kernels' own durations (dispatch timestamps, tsdf_volume_set_timing) INSIDE the two-stream run
against the wall time per step -- what is left is launch gaps, events and the small kernels.   python tools/dbg_pipeline_kernels.py [steps]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, tsdf_amd
from tsdf_amd import synth
from tsdf_amd.pipeline import FusionPipeline
W, H, n = 640, 480, 512
K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Wu = 8
frames, cams = [], []
for i in range(K + Wu + 1):
    d, cam = synth.depth_frame(i % 200, 200, seed=0x5EED0003); frames.append(d); cams.append(cam)
depth = torch.from_numpy(np.stack(frames).view(np.int16)).cuda()
vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda"); norm = torch.empty_like(vert)
bil = tsdf_amd.BilateralFilter(30.0, 4.5); rc = tsdf_amd.GPURaycaster(W, H)
for name, overlap, timing, equal in (("two streams", True, False, False), ("two streams, kernels timed", True, True, False), ("one stream, kernels timed", False, True, False),
                                     ("two streams of equal priority, kernels timed", True, True, True), ("two streams of equal priority", True, False, True), ("two streams", True, False, False)):
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    p = FusionPipeline(vol, bil, rc, W, H, overlap=overlap, equal_priority=equal)
    for i in range(Wu):
        p.step(depth[i].data_ptr(), cams[i], vert.data_ptr(), norm.data_ptr(), depth[i + 1].data_ptr(), cams[i + 1])
    torch.cuda.synchronize()
    if timing: vol.set_timing(True)
    t0 = time.perf_counter()
    for i in range(Wu, Wu + K):
        p.step(depth[i].data_ptr(), cams[i], vert.data_ptr(), norm.data_ptr(), depth[i + 1].data_ptr(), cams[i + 1])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    msg = "%-46s %.4f ms per step" % (name, dt)
    if timing:
        ki, kr, kt = vol.kernel_time("integrate")[1], vol.kernel_time("raycast")[1], vol.kernel_time("raycast_tail")[1]
        msg += "; integrate %.4f + bulk %.4f + tail %.4f = %.4f; rest %.4f" % (ki, kr, kt, ki + kr + kt, dt - ki - kr - kt)
    print(msg, flush=True)
    p.close(); vol.close()
