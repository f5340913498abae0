"""One rank's PIPELINED step of a P-GPU run, emulated on one GPU, for every rank of P = 1, 2, 4, 8: the rank's Z-slab through
tsdf_pipeline_step with the loopback exchange (tsdf_slab_exchange_create_loopback: the all-gather's launches and the bytes of P record
buffers landing in this GPU's memory, the merge over P buffers -- everything a rank does except the wire), the next frame's filter and
culling on the second stream, and (xs=1) the exchange + merge on a third stream beside the next frame's integrate.  The step a node
would see is the slowest rank's (+ the wire where it is not hidden: 2.46 MB per peer over that peer's own xGMI link, 16 us).
Nothing here has run on more than one GPU.      python tools/dbg_slab_pipeline.py [config3|config4] [steps]"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, torch
from tsdf_amd import synth, multi
from tsdf_amd.pipeline import FusionPipeline
workload = sys.argv[1] if len(sys.argv) > 1 else "config3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
inside = workload == "config4"
n, W, H, Wu = (1024 if inside else 512), 640, 480, 5
n_stream, seed = (100, 0x5EED0004) if inside else (200, 0x5EED0003)
raw = [synth.depth_frame(i, n_stream, seed=seed, inside=inside) for i in range(Wu + K + 1)]
depth = torch.from_numpy(np.stack([d for d, _ in raw]).view(np.int16)).cuda()
cams = [c for _, c in raw]
vert = torch.empty((H * W, 3), dtype=torch.float32, device="cuda"); norm = torch.empty_like(vert)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
rc = tsdf_amd.GPURaycaster(W, H)
wi, wr, wc = 0.4, 0.6, 0.02
costs = multi.plane_costs(lambda g: tsdf_amd.TSDFVolume(g, (3000.0,) * 3), [raw[i][0] for i in (0, Wu + K // 2, Wu + K - 1)], [cams[i] for i in (0, Wu + K // 2, Wu + K - 1)],
                          (n, n, n), integrate_weight=wi, raycast_weight=wr, constant=wc)

def rank_step(ranges, r, P, xs):
    """ms per pipelined step of rank r (median of 3 timed regions of K steps)"""
    zb, ze = ranges[r]
    v = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3, slab=(zb, ze)) if P > 1 else tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3)
    ex = multi.LoopbackExchange(r, P) if P > 1 else None
    pipe = FusionPipeline(v, bil, rc, W, H, overlap=True, exchange=ex, exchange_stream=bool(xs) and P > 1)
    runs = []
    for rep in range(3):
        v.clear()
        for i in range(Wu):
            pipe.step(depth[i].data_ptr(), cams[i], vert.data_ptr(), norm.data_ptr(), depth[i + 1].data_ptr(), cams[i + 1])
        pipe.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(Wu, Wu + K):
            pipe.step(depth[i].data_ptr(), cams[i], vert.data_ptr(), norm.data_ptr(), depth[i + 1].data_ptr(), cams[i + 1])
        pipe.synchronize(); torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) / K * 1e3)
    pipe.close()
    if ex is not None: ex.close()
    v.close()
    return sorted(runs)[1]

hop_ms = W * H * 8 / 1e6 / 153.0
base = None
for P in (1, 2, 4, 8):
    plans = {"uniform": [multi.slab_range(n, P, r) for r in range(P)]}
    if P > 1:
        plans["balanced"] = multi.balanced_slab_ranges(costs, P, min_planes=8)
    for xs in ((0,) if P == 1 else (0, 1)):
        for name, ranges in plans.items():
            t = [rank_step(ranges, r, P, xs) for r in range(P)]
            if name != "uniform" or P == 1:
                pass
            else:   # one round of the measured rebalancing, from the uniform split
                plans_ref = multi.refine_slab_ranges(ranges, t, n, min_planes=8)
                t_ref = [rank_step(plans_ref, r, P, xs) for r in range(P)]
                print("P=%d xs=%d %-8s per rank ms %s planes %s -> slowest %.4f" % (P, xs, "refined", [round(x, 4) for x in t_ref], [b - a for a, b in plans_ref], max(t_ref)))
                if base: print("      speed-up x%.2f (wire hidden) / x%.2f (+ %.0f us of wire)" % (base / max(t_ref), base / (max(t_ref) + hop_ms), hop_ms * 1e3))
            print("P=%d xs=%d %-8s per rank ms %s planes %s -> slowest %.4f" % (P, xs, name, [round(x, 4) for x in t], [b - a for a, b in ranges], max(t)))
            if P == 1: base = max(t)
            else: print("      speed-up x%.2f (wire hidden) / x%.2f (+ %.0f us of wire)" % (base / max(t), base / (max(t) + hop_ms), hop_ms * 1e3))
