"""One-GPU experiment: does the NEXT frame's bilateral filter, queued on a second stream, hide behind this frame's ray cast?
    python tools/dbg_overlap.py [steps]
Prints ms per step of the strictly sequential step and of the overlapped variants, and checks that the last picture is the
same bits in every variant."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import numpy as np, torch, tsdf_amd
from tsdf_amd import synth

W, H, n = 640, 480, 512
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
Wu = 10
frames, cams = [], []
for i in range(K + Wu + 1):
    d, cam = synth.depth_frame(i % 200, 200, seed=0x5EED0003)
    frames.append(d); cams.append(cam)
depth_dev = torch.from_numpy(np.stack(frames).view(np.int16)).cuda()
vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
norm = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
filt = [torch.empty((H * W,), dtype=torch.int16, device="cuda") for _ in range(2)]
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
rc = tsdf_amd.GPURaycaster(W, H)


def run(mode):
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    main = torch.cuda.Stream(priority=-1) if mode in ("mainhigh", "seqhigh") else torch.cuda.current_stream()
    vol.set_stream(main.cuda_stream)
    rc_stream = main
    if mode in ("seq", "seqhigh"):
        side = None
    elif mode == "mainhigh":
        side = torch.cuda.Stream(priority=0)      # (lower than the main stream: fills what the main stream leaves idle)
    elif mode == "low":
        side = torch.cuda.Stream(priority=0)     # (torch: lower number = higher priority; 0 is the default / lowest)
    elif mode == "high":
        side = torch.cuda.Stream(priority=-1)
    else:
        side = torch.cuda.Stream()
    int_done = {}
    bil_done = {}

    def filt_on(i, s):
        bil.filter_device(depth_dev[i].data_ptr(), filt[i % 2].data_ptr(), W, H, bits=16, stream=s.cuda_stream)

    def step(i, last):
        if side is None:
            filt_on(i, main)
        else:
            if i in bil_done:
                main.wait_event(bil_done.pop(i))
            else:
                filt_on(i, main)
        vol.integrate_device(filt[i % 2].data_ptr(), W, H, cams[i])
        if side is not None and not last:
            if mode == "early":      # the filter of frame i+1 goes in beside integrate of frame i (its buffer was last read by integrate i-1)
                pass
            e = torch.cuda.Event(); e.record(main); int_done[i] = e
            # buffer (i+1) % 2 was last read by integrate(i-1): queued on `main` before this point
            if i - 1 in int_done:
                side.wait_event(int_done.pop(i - 1))
            filt_on(i + 1, side)
            e2 = torch.cuda.Event(); e2.record(side); bil_done[i + 1] = e2
        rc.raycast_device(vol, cams[i], vert.data_ptr(), norm.data_ptr())

    for i in range(Wu):
        step(i, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Wu, Wu + K):
        step(i, i == Wu + K - 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    pic = vert.clone()
    vol.close()
    return dt, pic


ref = None
print("stream priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
for rep in range(2):
    for mode in ("seq", "seqhigh", "side", "mainhigh"):
        dt, pic = run(mode)
        if ref is None:
            ref = pic
        same = bool(((pic.view(torch.int32) == ref.view(torch.int32)) | (torch.isnan(pic) & torch.isnan(ref))).all().item())
        print("%-5s %.4f ms per step, picture identical: %s" % (mode, dt, same), flush=True)
