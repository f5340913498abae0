"""Per-rank stage times of the Z-slab decomposition, emulated on ONE GPU: for P = 1, 2, 4, 8 every slab is built, fed the
same frames and timed on its own (integrate, slab ray cast); the slowest slab is what a P-GPU step would wait for.  The last
lines price the step a node would see: slowest slab + the replicated filter + merge + the all-gather of the 8-byte records at
one xGMI link's rate (direct: every peer over its own link at once; ring: P - 1 hops), nothing of which has run on P > 1 GPUs.
python tools/dbg_slab_scaling.py [config3|config4]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, time, torch
from tsdf_amd import synth, multi
workload = sys.argv[1] if len(sys.argv) > 1 else "config3"
inside = workload == "config4"
n, W, H = (1024 if inside else 512), 640, 480
n_stream, seed = (100, 0x5EED0004) if inside else (200, 0x5EED0003)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
frames = []
for i in range(12):
    d, cam = synth.depth_frame(i, n_stream, seed=seed, inside=inside)
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
raw = [synth.depth_frame(i, n_stream, seed=seed, inside=inside) for i in (0, 6, 11)]
wi, wr, wc = (float(x) for x in os.environ.get("PLAN_WEIGHTS", "0.4,0.6,0.02").split(","))
costs = multi.plane_costs(lambda g: tsdf_amd.TSDFVolume(g, (3000.0,) * 3), [d for d, _ in raw], [c for _, c in raw], (n, n, n),
                          integrate_weight=wi, raycast_weight=wr, constant=wc)
measured = {}
best = {}
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
    if P not in best or max(tot) < best[P][0]:
        best[P] = (max(tot), plan, worst_i, worst_r)
    print("P=%d %-8s slowest slab: integrate %.3f ms, raycast %.3f ms, integrate+raycast %.3f (mean %.3f, x%.2f)  planes %s  per slab: %s"
          % (P, plan, worst_i, worst_r, max(tot), sum(tot) / len(tot), max(tot) / (sum(tot) / len(tot)), [b - a for a, b in ranges], rows))

# What a node's step would be made of (nothing here has run on more than one GPU): the slowest slab of the best plan, the filter
# every rank repeats, merge + normals, and the all-gather of W * H 8-byte records per rank.  xGMI: 7 links of ~153 GB/s per GPU,
# point to point -- "direct" has every rank send its records to each peer over that peer's own link at the same time (one record
# buffer at one link's rate), "ring" passes them round (P - 1 hops); 20 us stand for the collective's launch and synchronisation
# (the one-rank RCCL run's exchange stage: 0.019 ms).
record_mb = W * H * 8 / 1e6
link_gbs, launch_ms, filter_ms, merge_ms = 153.0, 0.020, 0.036, 0.010
print("%s: record buffer %.2f MB per rank; one link %.0f GB/s -> %.1f us per buffer" % (workload, record_mb, link_gbs, record_mb / link_gbs * 1e3))
t1 = best[1][0] + filter_ms
for P in (1, 2, 4, 8):
    slab = best[P][0]
    hop = record_mb / link_gbs           # ms per record buffer over one link
    direct = 0.0 if P == 1 else launch_ms + hop
    ring = 0.0 if P == 1 else launch_ms + (P - 1) * hop
    extra = filter_ms + (merge_ms if P > 1 else 0.0)
    print("P=%d  plan %-8s slowest slab %.3f ms (integrate %.3f, ray cast %.3f)  predicted step: %.3f ms direct / %.3f ms ring   speed-up x%.2f / x%.2f"
          % (P, best[P][1], slab, best[P][2], best[P][3], slab + extra + direct, slab + extra + ring, t1 / (slab + extra + direct), t1 / (slab + extra + ring)))
