"""FusionPipeline (next frame's filter on a lower-priority stream) against the sequential step, same box:
    python tools/dbg_pipeline.py [steps]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, tsdf_amd
from tsdf_amd import synth
from tsdf_amd.pipeline import FusionPipeline
W, H, n = 640, 480, 512
K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
Wu = 10
frames, cams = [], []
for i in range(K + Wu + 1):
    d, cam = synth.depth_frame(i % 200, 200, seed=0x5EED0003); frames.append(d); cams.append(cam)
depth = torch.from_numpy(np.stack(frames).view(np.int16)).cuda()
vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda"); norm = torch.empty_like(vert)
bil = tsdf_amd.BilateralFilter(30.0, 4.5); rc = tsdf_amd.GPURaycaster(W, H)
def run(overlap, gate, prepare=False):
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    p = FusionPipeline(vol, bil, rc, W, H, overlap=overlap, equal_priority=not gate)
    def step(i, last):
        p.step(depth[i].data_ptr(), cams[i], vert.data_ptr(), norm.data_ptr(), None if last else depth[i + 1].data_ptr(),
               cams[i + 1] if (prepare and not last) else None)
    for i in range(Wu): step(i, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Wu, Wu + K): step(i, False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    pic = vert.clone(); p.close(); vol.close()
    return dt, pic
ref = None
for rep in range(3):
    for name, ov, gate, prep in (("sequential", False, True, False), ("filter ahead", True, True, False), ("filter ahead, equal priorities", True, False, False),
                                 ("filter + brick culling ahead", True, True, True)):
        dt, pic = run(ov, gate, prep)
        if ref is None: ref = pic
        same = bool(((pic.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(pic) & torch.isnan(ref))).all().item())
        print("%-42s %.4f ms per step, picture identical: %s" % (name, dt, same), flush=True)
