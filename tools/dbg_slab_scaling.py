"""Per-rank stage times of the Z-slab decomposition, emulated on ONE GPU: for P = 1, 2, 4, 8 every slab is built, fed the
same frames and timed on its own (integrate, slab ray cast); the slowest slab is what a P-GPU step would wait for.
python tools/dbg_slab_scaling.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, time, torch
from tsdf_amd import synth, multi
n, W, H = 512, 640, 480
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
frames = []
for i in range(12):
    d, cam = synth.depth_frame(i, 200, seed=0x5EED0003)
    f = d.copy(); bil.filter(f, W, H)
    frames.append((torch.from_numpy(f.view(np.int16)).cuda(), cam))
rc = tsdf_amd.GPURaycaster(W, H)
hits = torch.empty((W * H, 2), dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream()
def timed(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(reps): fn()
    b.record(stream); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
raw = [synth.depth_frame(i, 200, seed=0x5EED0003) for i in (0, 6, 11)]
wi, wr, wc = (float(x) for x in os.environ.get("PLAN_WEIGHTS", "0.4,0.6,0.02").split(","))
costs = multi.plane_costs(lambda g: tsdf_amd.TSDFVolume(g, (3000.0,) * 3), [d for d, _ in raw], [c for _, c in raw], (n, n, n),
                          integrate_weight=wi, raycast_weight=wr, constant=wc)
measured = {}
for P in (1, 2, 4, 8):
  for plan in (("uniform", "balanced", "refined1", "refined2", "refined3") if P > 1 else ("uniform",)):
    if plan == "uniform":
        ranges = [multi.slab_range(n, P, r) for r in range(P)]
    elif plan == "balanced":
        ranges = multi.balanced_slab_ranges(costs, P, min_planes=8)
    else:   # measured rebalancing, starting from the uniform split
        prev_ranges, prev_t = measured[P]
        ranges = multi.refine_slab_ranges(prev_ranges, prev_t, n, min_planes=8)
    worst_i = worst_r = 0.0
    rows = []
    for r in range(P):
        zb, ze = ranges[r]
        v = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3, slab=(zb, ze)) if P > 1 else tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
        v.set_stream(stream.cuda_stream)
        for fr, cam in frames[:8]:
            v.integrate_device(fr.data_ptr(), W, H, cam)
        k = [8]
        def integ():
            fr, cam = frames[k[0] % 12]; k[0] += 1
            v.integrate_device(fr.data_ptr(), W, H, cam)
        ti = timed(integ, 4)
        cam = frames[11][1]
        if P > 1:
            rc.raycast_slab_device(v, cam, hits.data_ptr())
            tr = timed(lambda: rc.raycast_slab_device(v, cam, hits.data_ptr()), 10)
        else:
            vert = torch.empty((W * H, 3), dtype=torch.float32, device="cuda")
            rc.raycast_device(v, cam, vert.data_ptr(), None)
            tr = timed(lambda: rc.raycast_device(v, cam, vert.data_ptr(), None), 10)
        rows.append((round(ti, 3), round(tr, 3)))
        worst_i, worst_r = max(worst_i, ti), max(worst_r, tr)
        v.close()
    tot = [a + b for a, b in rows]
    if plan != "balanced":
        measured[P] = (ranges, tot)
    print("P=%d %-8s slowest slab: integrate %.3f ms, raycast %.3f ms, integrate+raycast %.3f (mean %.3f, x%.2f)  planes %s  per slab: %s"
          % (P, plan, worst_i, worst_r, max(tot), sum(tot) / len(tot), max(tot) / (sum(tot) / len(tot)), [b - a for a, b in ranges], rows))
