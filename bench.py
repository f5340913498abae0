#!/usr/bin/env python3
"""bench.py -- the TSDF hot path on MI355X: bilateral filter -> integrate -> raycast -> normals.

    python bench.py --gpus N --steps K --warmup W

A "step" is one 640x480 depth frame of the synthetic TUM surrogate (tsdf_amd/synth.py) pushed through the
whole path on a 512^3 / 3000 mm volume (BASELINE.json configs[2]); inputs are resident in HBM before the
timed region starts.  N > 1: the volume is split into N Z-slabs, one process per GPU (torch.distributed over
RCCL); every rank integrates its slab (+1 halo plane), ray casts the samples it owns, and one all-gather of
16-byte hit records per pixel is merged by a min-k select.  Total work is fixed => "strong".

Rank 0 prints ONE JSON line.  `value` = voxels of the grid pushed through the whole step per second (whole
job); per-stage figures (integrate Mvoxels/s, raycast Mrays/s), the roofline of the dominant kernel and the
CPU baseline (the oracle, timed here on the host cores on a bounded sample) ride along in the same object.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 640, 480
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
SEED = 0x5EED0003            # config 3 stream


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--physical", type=float, default=3000.0)
    ap.add_argument("--stream-frames", type=int, default=200, help="length of the synthetic trajectory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--event-period", type=int, default=16,
                    help="per-stage and per-kernel HIP events are recorded on every n-th timed step (0 = on the first one only): each record costs "
                         "a few microseconds of stream time, 9 %% of the step when every launch is bracketed")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: filter frame i+1 after, not during, the exchange of frame i")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--path-only", action="store_true",
                    help="only the timed hot path and its roofline (no ICP / tracking / host-buffer legs): what the profiling passes run")
    return ap.parse_args()


def main():
    args = parse()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (before the HIP runtime starts: RCCL's IPC needs it on this host driver)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # launched without torchrun for N>1 is a usage error; N=1 runs standalone
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
    # Debug aid for a 1-GPU box: TSDF_BENCH_SHARE_GPU=1 puts every rank on device 0 and uses gloo for the
    # collective, so the N>1 code path can be exercised (the numbers mean nothing then).
    share = os.environ.get("TSDF_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import tsdf_amd
    from tsdf_amd import synth
    from tsdf_amd._capi import check, lib

    check(lib.tsdf_set_device(local_rank))
    n = args.grid
    N_vox = n * n * n
    K, Wu = args.steps, args.warmup
    n_frames = K + Wu + 1

    # ---- inputs: synthetic stream, resident in HBM before timing ---------------------------------
    frames, cams = [], []
    for i in range(n_frames):
        d, cam = synth.depth_frame(i % args.stream_frames, args.stream_frames, seed=SEED)
        frames.append(d)
        cams.append(cam)
    depth_dev = torch.from_numpy(np.stack(frames).view(np.int16)).cuda()           # (F, H*W) uint16 bits
    filt_dev = torch.empty((H * W,), dtype=torch.int16, device="cuda")
    vert_dev = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    norm_dev = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")

    # ---- volume (whole, or this rank's Z-slab) -----------------------------------------------------
    if world == 1:
        vol = tsdf_amd.TSDFVolume((n, n, n), (args.physical,) * 3)
    else:
        from tsdf_amd.multi import slab_range
        zb, ze = slab_range(n, world, rank)
        vol = tsdf_amd.TSDFVolume((n, n, n), (args.physical,) * 3, slab=(zb, ze))
        hits_mine = torch.empty((H * W, 4), dtype=torch.float32, device="cuda")
        hits_all = torch.empty((world, H * W, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    vol.set_stream(stream.cuda_stream)
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    rc = tsdf_amd.GPURaycaster(W, H)

    stage_names = ["bilateral", "integrate", "raycast", "exchange", "normals"]
    ev = {s: [] for s in stage_names}

    # N > 1: while the hit records are exchanged the compute units idle, and the next frame's bilateral filter does not
    # depend on the volume -- it is queued on a second stream at that point (two filtered-frame buffers).  Every timed
    # step still contains one filter, one integrate, one ray cast, one exchange, one normal map.  (On one GPU there is no
    # idle phase to fill: tried, no gain, so the single-GPU step stays strictly sequential.)
    overlap = world > 1 and not args.no_overlap
    side = torch.cuda.Stream() if overlap else None
    filt2 = [filt_dev, torch.empty_like(filt_dev)] if overlap else [filt_dev, filt_dev]
    prefiltered = {}        # frame index -> (event on the side stream, timing pair or None)

    def step(i, timed, timed_next=False):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if timed else None
        cam = cams[i]
        fbuf = filt2[i % 2]
        if i in prefiltered:
            done, pair = prefiltered.pop(i)
            stream.wait_event(done)
            if timed and pair is not None:
                ev["bilateral"].append(pair)
            if timed: e[1].record(stream)
        else:
            if timed: e[0].record(stream)
            bil.filter_device(depth_dev[i].data_ptr(), fbuf.data_ptr(), W, H, bits=16, stream=stream.cuda_stream)
            if timed:
                e[1].record(stream)
                ev["bilateral"].append((e[0], e[1]))
        vol.integrate_device(fbuf.data_ptr(), W, H, cam)
        if timed: e[2].record(stream)
        if world == 1:
            rc.raycast_device(vol, cam, vert_dev.data_ptr(), None if os.environ.get('BENCH_SPLIT_NORMALS') else norm_dev.data_ptr())   # vertices and normals in one go
            if timed: e[3].record(stream)
        else:
            rc.raycast_slab_device(vol, cam, hits_mine.data_ptr())
            if timed: e[3].record(stream)
            if overlap and i + 1 < n_frames:
                cast = torch.cuda.Event()
                cast.record(stream)            # integrate(i) and the slab cast are done: the other buffer is free, the CUs too
                side.wait_event(cast)
                pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed_next else None   # (it is the next step's filter)
                if pair: pair[0].record(side)
                bil.filter_device(depth_dev[i + 1].data_ptr(), filt2[(i + 1) % 2].data_ptr(), W, H, bits=16, stream=side.cuda_stream)
                if pair: pair[1].record(side)
                done = torch.cuda.Event()
                done.record(side)
                prefiltered[i + 1] = (done, pair)
            if share:   # gloo: stage through the host
                h_all = torch.empty(hits_all.shape, dtype=hits_all.dtype)
                dist.all_gather_into_tensor(h_all.view(-1), hits_mine.cpu().view(-1))
                hits_all.copy_(h_all)
            else:
                dist.all_gather_into_tensor(hits_all.view(-1), hits_mine.view(-1))
            tsdf_amd.merge_hits_normals_device(hits_all.data_ptr(), world, W, H, vert_dev.data_ptr(), norm_dev.data_ptr(), stream.cuda_stream)   # merged vertices and their normals in one go
        if timed: e[4].record(stream)
        if os.environ.get('BENCH_SPLIT_NORMALS'):    # (the ray cast, or the merge of the slabs' records, has formed the normals with the vertices)
            tsdf_amd.compute_normals_device(W, H, vert_dev.data_ptr(), norm_dev.data_ptr(), stream.cuda_stream)
        if timed:
            e[5].record(stream)
            for j, s in enumerate(stage_names):
                if s != "bilateral":
                    ev[s].append((e[j], e[j + 1]))

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warmup, then the timed region -------------------------------------------------------------
    for i in range(Wu):
        if i == Wu - 1:
            vol.set_counting(True)            # voxels updated by the frame just before the timed region (byte model below)
        step(i, False)
    torch.cuda.synchronize()
    U_before = vol.last_updated_voxels() if Wu > 0 else None
    vol.set_counting(False)
    period = args.event_period if args.event_period > 0 else K + 1
    vol.set_timing(period)      # HIP events around the dominant kernels, on the stream they are launched on
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Wu, Wu + K):
        step(i, (i - Wu) % period == 0, (i + 1 - Wu) % period == 0 and i + 1 < Wu + K)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage_ms = {s: (float(np.mean([a.elapsed_time(b) for a, b in ev[s]])) if ev[s] else None) for s in stage_names}   # (None: not sampled)
    kern = {w: vol.kernel_time(w) for w in ("integrate", "raycast", "raycast_tail")}     # (launches, avg ms), kernel only
    vol.set_timing(False)
    ms_per_step = elapsed * 1e3 / K
    value = N_vox * K / elapsed / 1e6

    out = {
        "metric": "Mvoxels/s integrate + Mrays/s raycast, 512^3 grid, 640x480 depth; 1/2/4/8 GPU",
        "value": round(value, 3),
        "unit": "Mvoxels/s",
        "value_definition": "grid voxels pushed through the whole step (bilateral+integrate+raycast+normals) per second",
        "n_gpus": world,
        "steps": K,
        "warmup": Wu,
        "event_period": period,     # per-stage / per-kernel HIP events on every n-th timed step (each record costs stream time)
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "configs[2]: %d^3 TSDF over %.0f mm, synthetic TUM-surrogate stream (%d-frame "
                               "trajectory, seed 0x%X), 640x480 uint16 depth, bilateral(30,4.5) + integrate + "
                               "raycast + normals per frame" % (n, args.physical, args.stream_frames, SEED),
                   "grid": [n, n, n], "image": [W, H], "parallelism": "zslab%d" % world,
                   "overlap": "bilateral(i+1) on a second stream during the exchange of frame i" if overlap else "none"},
        "integrate_mvoxels_per_s": round(N_vox / (stage_ms["integrate"] * 1e-3) / 1e6, 1),
        "raycast_mrays_per_s": round(W * H / ((stage_ms["raycast"] + stage_ms["exchange"] + stage_ms["normals"]) * 1e-3) / 1e6, 2),
        "stage_ms": {s: (round(v, 4) if v is not None else None) for s, v in stage_ms.items()},
        # sum of the finite vertex coordinates of the last frame's picture: equal between runs that differ only in schedule
        "last_frame_vertex_checksum": float(torch.nan_to_num(vert_dev.double(), nan=0.0).sum().item()),
    }

    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel (by time): HIP-event durations measured above --------------
        last = Wu + K                                  # one more frame, untimed, for the byte counts
        st = rc.stats(vol, cams[last - 1])             # S samples, T distinct voxels touched at the end state
        vol.set_counting(True)
        bil.filter_device(depth_dev[last].data_ptr(), filt_dev.data_ptr(), W, H, bits=16, stream=stream.cuda_stream)
        vol.integrate_device(filt_dev.data_ptr(), W, H, cams[last])
        U_after = vol.last_updated_voxels()
        vol.set_counting(False)
        # the kernel time is an average over the timed launches; the updated-voxel count drifts as the camera moves, so the
        # bytes are priced at the mean of the frame before and the frame after the timed region
        U = (U_before + U_after) // 2 if U_before is not None else U_after
        ray_bytes = 4 * st["touched"] + 12 * W * H     # 4*T + vertex store (normals kernel: +12*W*H)
        int_bytes = 16 * U + 2 * W * H                 # 16*U + depth frame
        # the march is two kernels (bulk + tail queue); its bytes are priced against their summed duration.  The
        # stage times also hold the occupancy refresh / merge / cull kernels.
        ray_main_ms, ray_tail_ms = kern["raycast"][1], kern["raycast_tail"][1]
        ray_ms = (ray_main_ms + ray_tail_ms) or stage_ms["raycast"]
        int_ms = kern["integrate"][1] or stage_ms["integrate"]
        ray_gbs = ray_bytes / (ray_ms * 1e-3) / 1e9
        int_gbs = int_bytes / (int_ms * 1e-3) / 1e9
        traffic = load_traffic()
        # dominant = the single kernel with the longest average launch
        dominant = "raycast" if max(ray_main_ms, ray_tail_ms) >= int_ms else "integrate"
        roof_ray = {"kernel": "process_ray_kernel + process_ray_tail_kernel", "bound": "hbm", "achieved": round(ray_gbs, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ray_gbs / HBM_PEAK_GBS, 5), "traffic": (traffic.get("process_ray_kernel", 0) + traffic.get("process_ray_tail_kernel", 0)) or None,
                    "algorithmic_bytes": ray_bytes, "avg_launch_ms": round(ray_ms, 4), "launches_timed": kern["raycast"][0],
                    "avg_launch_ms_by_kernel": {"process_ray_kernel": round(ray_main_ms, 4), "process_ray_tail_kernel": round(ray_tail_ms, 4)},
                    "T_voxels_touched": st["touched"], "S_samples": st["samples"],
                    "samples_evaluated_after_exact_skipping": st["evaluated"],
                    "msamples_per_s": round(st["samples"] / (ray_ms * 1e-3) / 1e6, 1),
                    "l2_level_gbs": round(32 * st["samples"] / (ray_ms * 1e-3) / 1e9, 1)}
        roof_int = {"kernel": "integrate_kernel", "bound": "hbm", "achieved": round(int_gbs, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(int_gbs / HBM_PEAK_GBS, 5), "traffic": traffic.get("integrate_kernel"),
                    "algorithmic_bytes": int_bytes, "avg_launch_ms": round(int_ms, 4), "launches_timed": kern["integrate"][0],
                    "U_voxels_updated": U, "U_first_last": [U_before, U_after], "dense_bytes": 16 * N_vox}
        # the fraction against the measured device-to-device copy rate as well as the nominal peak
        copy_gbs = measured_copy_gbs(torch)
        for r_ in (roof_ray, roof_int):
            r_["measured_copy_gbs"] = round(copy_gbs, 1)
            r_["frac_of_measured_copy"] = round(r_["achieved"] / copy_gbs, 5)
        out["roofline"] = roof_ray if dominant == "raycast" else roof_int
        out["roofline_other"] = roof_int if dominant == "raycast" else roof_ray
        if not args.path_only:
            out["host_buffer_api"] = host_api_time(vol, bil, frames, cams, last)
            out["icp"] = icp_tracking(tsdf_amd, synth, depth_dev, frames, stream, not args.no_cpu_baseline)
            out["tracking"] = tracking_loop(tsdf_amd, synth, n, args.physical, args.stream_frames)
        if not args.no_parity:
            out["parity"] = parity_gate(tsdf_amd, synth, n_small=96)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(vol, frames[last], cams[last], n, args.physical, args.cpu_budget_s)

    if rank == 0:
        def finite(o):      # strict JSON: a non-finite number (an unsampled average, an empty ratio) becomes null
            if isinstance(o, float):
                return o if math.isfinite(o) else None
            if isinstance(o, dict):
                return {k: finite(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [finite(v) for v in o]
            return o
        print(json.dumps(finite(out), allow_nan=False))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_copy_gbs(torch):
    """Device-to-device copy of 1 GiB (read + write = 2 GiB of traffic), best of 5: the practical HBM ceiling."""
    a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def host_api_time(vol, bil, frames, cams, last):
    """End-to-end time of the reference-shaped blocking calls on HOST buffers (PCIe inclusive): BilateralFilter::filter,
    TSDFVolume::integrate (614 KB up), TSDFVolume::raycast (2 x 3.7 MB down).  Reported beside the resident-in-HBM value,
    never as `value`."""
    reps = 5
    t = {"bilateral": 0.0, "integrate": 0.0, "raycast": 0.0}
    for r in range(reps + 1):
        f = frames[last].copy()
        t0 = time.perf_counter()
        bil.filter(f, W, H)
        t1 = time.perf_counter()
        vol.integrate(f, W, H, cams[last])
        t2 = time.perf_counter()
        vol.raycast(W, H, cams[last])
        t3 = time.perf_counter()
        if r > 0:   # first round warms the staging buffers
            t["bilateral"] += t1 - t0
            t["integrate"] += t2 - t1
            t["raycast"] += t3 - t2
    res = {k: round(v * 1e3 / reps, 4) for k, v in t.items()}
    res["ms_per_frame"] = round(sum(res.values()), 4)
    return res


def icp_tracking(tsdf_amd, synth, depth_dev, frames, stream, with_cpu):
    """Next-row measurement (BASELINE config 5): ICP tracking between two consecutive bilateral-filtered
    frames of the stream -- both pyramids (initICPModel + initICP) and the 19 Gauss-Newton iterations of
    getIncrementalTransformation, device resident.  Not part of `value`.  The oracle's ICP on one host thread beside it."""
    import torch
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    a = torch.empty((H * W,), dtype=torch.int16, device="cuda")
    b = torch.empty((H * W,), dtype=torch.int16, device="cuda")
    bil.filter_device(depth_dev[0].data_ptr(), a.data_ptr(), W, H, bits=16, stream=stream.cuda_stream)
    bil.filter_device(depth_dev[1].data_ptr(), b.data_ptr(), W, H, bits=16, stream=stream.cuda_stream)
    icp = tsdf_amd.ICPOdometry(W, H, 331.0, 234.6, 591.1, 590.1)
    icp.set_stream(stream.cuda_stream)
    reps = 20
    for r in range(reps + 3):
        if r == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        icp.init_icp_device(a.data_ptr(), model=True)
        icp.init_icp_device(b.data_ptr())
        T = icp.get_incremental_transformation()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    res = {"ms_per_frame": round(ms, 4), "iterations": [4, 5, 10], "inliers": icp.last_inliers,
           "what": "initICPModel + initICP + getIncrementalTransformation, 640x480, 3 levels"}
    if with_cpu:
        import oracle as O
        fa = a.cpu().numpy().view(np.uint16)
        fb = b.cpu().numpy().view(np.uint16)
        t0 = time.perf_counter()
        To, _, inl = O.icp_incremental_transformation(fb, fa, W, H, 331.0, 234.6, 591.1, 590.1)
        res["cpu_oracle_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        res["cpu_cores"] = 1
        res["max_abs_pose_difference_vs_oracle"] = float(np.max(np.abs(T - To)))
    return res


def tracking_loop(tsdf_amd, synth, n, physical, stream_frames, n_frames=24):
    """BASELINE configs[4] without the mesh: the closed loop (filter, render the model from the previous pose, ICP,
    integrate) on a fresh volume of the bench size, poses from tracking instead of ground truth.  Reported beside `value`:
    time per frame and how far the tracked trajectory strays from the true one."""
    import torch
    from tsdf_amd.tracking import FrameToModelTracker
    vol = tsdf_amd.TSDFVolume((n, n, n), (physical,) * 3)
    tracker = FrameToModelTracker(vol, W, H)
    frames = [synth.depth_frame(i, stream_frames, seed=SEED) for i in range(n_frames)]
    dev = [torch.from_numpy(d.view(np.int16)).cuda() for d, _ in frames]
    worst_t = worst_r = 0.0
    t0 = None
    for i, (d, cam) in enumerate(frames):
        truth = cam.pose().astype(np.float64).reshape(4, 4).T
        if i == 4:                      # the first frames also build the occupancy flags / allocate scratch
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        pose = tracker.process_device(dev[i].data_ptr(), initial_pose=truth if i == 0 else None)
        worst_t = max(worst_t, float(np.linalg.norm(pose[:3, 3] - truth[:3, 3])))
        c = (np.trace(pose[:3, :3].T @ truth[:3, :3]) - 1.0) / 2.0
        worst_r = max(worst_r, float(np.arccos(np.clip(c, -1.0, 1.0))))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / (n_frames - 4)
    # ... and the mesh of the tracked model: extract_surface = marching cubes on the device (vertices to the host), beside
    # the host implementation on the downloaded distances (the same vertices, bit for bit: tests/test_parity_marching_cubes.py)
    t1 = time.perf_counter()
    mesh_dev = vol.extract_surface()
    t1b = time.perf_counter()
    dist_host = vol.get_distance_data()
    t2 = time.perf_counter()
    mesh = tsdf_amd.marching_cubes(dist_host, (n, n, n), (physical / n,) * 3)
    t3 = time.perf_counter()
    return {"ms_per_frame": round(ms, 4),
            "mesh": {"extract_surface_device_s": round(t1b - t1, 4), "same_as_host": bool(mesh_dev.shape == mesh.shape and np.array_equal(mesh_dev.view(np.uint32), mesh.view(np.uint32))), "host_download_s": round(t2 - t1b, 3), "host_marching_cubes_s": round(t3 - t2, 3), "triangles": int(mesh.shape[0] // 3)}, "frames": n_frames, "max_translation_error_mm": round(worst_t, 3),
            "max_rotation_error_rad": round(worst_r, 6),
            "what": "bilateral + raycast(prev pose) + render depth + ICP (3 levels, 19 iterations) + integrate per frame"}


def load_traffic():
    """HBM bytes per launch from the PMC passes (profiles/traffic.json, written by tools/pmc_traffic.py from
    rocprofv3 --pmc runs of this same command); empty when no counter run has been committed."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("bytes_per_launch", {})
        except Exception:
            return {}
    return {}


def parity_gate(tsdf_amd, synth, n_small):
    """GPU vs CPU oracle on a reduced grid (first and last frame of a short stream), as SURVEY.md 8d asks.
    The oracle is the checker here, nothing it computes is timed or reported as a result."""
    import oracle as O
    gv = tsdf_amd.TSDFVolume((n_small,) * 3, (3000.0,) * 3)
    ov = O.Volume((n_small,) * 3, (3000.0,) * 3)
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    res = {"grid": n_small, "frames": 3}
    for i in (0, 1, 2):
        d, cam = synth.depth_frame(i, 200, seed=SEED)
        f = d.copy()
        bil.filter(f, W, H)
        fo = O.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=O.max_threads()).reshape(-1)
        res["bilateral_mismatch"] = res.get("bilateral_mismatch", 0) + int((f != fo).sum())
        gv.integrate(f, W, H, cam)
        ov.integrate(fo, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=O.max_threads())
    gd, gw = gv.get_distance_data(), gv.get_weight_data()
    res["weight_mismatch"] = int((gw != ov.weight).sum())
    m = ov.weight > 0
    rel = np.abs(gd[m] - ov.dist[m]) / np.maximum(np.abs(ov.dist[m]), 1e-6)
    res["dist_max_rel_err"] = float(rel.max()) if m.any() else 0.0
    res["dist_bit_mismatch"] = int((gd.view(np.uint32) != ov.dist.view(np.uint32)).sum())
    d, cam = synth.depth_frame(0, 200, seed=SEED)
    V, Nn = gv.raycast(W, H, cam)
    Vo, No = ov.raycast(W, H, cam.pose(), cam.kinv(), nthreads=O.max_threads())
    res["nan_mask_mismatch"] = int((np.isnan(V) != np.isnan(Vo)).sum())
    ok = ~np.isnan(Vo)
    res["vertex_max_rel_err"] = float((np.abs(V[ok] - Vo[ok]) / np.maximum(np.abs(Vo[ok]), 1e-6)).max()) if ok.any() else 0.0
    okn = np.isfinite(No)
    res["normal_max_abs_err"] = float(np.abs(Nn[okn] - No[okn]).max()) if okn.any() else 0.0
    res["pass"] = bool(res["weight_mismatch"] == 0 and res["dist_max_rel_err"] <= 1e-4 and res["nan_mask_mismatch"] == 0
                       and res["vertex_max_rel_err"] <= 1e-4 and res["normal_max_abs_err"] <= 1e-4
                       and res["bilateral_mismatch"] == 0)
    return res


def cpu_baseline(vol, depth, cam, n, physical, budget_s):
    """The oracle (a port of the reference arithmetic, oracle/tsdf_oracle.c) timed on this box's host cores on a
    bounded sample of the same workload: one frame of bilateral + integrate over the full 512^3 grid and a ray
    cast of every 16th image row against the GPU-built volume state.  Reported only."""
    import oracle as O
    cores = O.max_threads()
    t = time.perf_counter()
    f = O.bilateral_u16(depth, W, H, 30.0, 4.5, nthreads=cores).reshape(-1)
    t_bil = time.perf_counter() - t
    ov = O.Volume((n, n, n), (physical,) * 3)
    ov.set_distance_data(vol.get_distance_data())
    ov.set_weight_data(vol.get_weight_data())
    t = time.perf_counter()
    U = ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=cores)
    t_int = time.perf_counter() - t
    # single-thread figure on a slab of planes so it stays bounded
    zs = max(1, n // 16)
    t = time.perf_counter()
    ov.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), z_range=(n // 2, n // 2 + zs), nthreads=1)
    t_int1 = (time.perf_counter() - t) * (n / zs)
    # probe with every 64th row, then size the sample to the remaining CPU budget (at least every 16th row)
    t = time.perf_counter()
    ov.raycast_rows(W, H, cam.pose(), cam.kinv(), 0, H, 64, nthreads=cores)
    probe = time.perf_counter() - t
    remaining = max(1.0, budget_s - (t_bil + t_int + t_int1 * zs / n + probe))
    row_step = int(min(16, max(1, math.ceil(64 * probe / remaining))))
    t = time.perf_counter()
    Vs, samples = ov.raycast_rows(W, H, cam.pose(), cam.kinv(), 0, H, row_step, nthreads=cores)
    t_ray = time.perf_counter() - t
    rays = W * len(range(0, H, row_step))
    t_ray_full = t_ray * (W * H / rays)
    step_s = t_bil + t_int + t_ray_full
    return {"value": round(n ** 3 / step_s / 1e6, 3), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": "1 frame: bilateral 640x480 + integrate over the full %d^3 grid (%d voxels updated) + ray cast of "
                      "every %dth row (%d rays, %d samples) scaled to 307200 rays; oracle/tsdf_oracle.c, OpenMP over "
                      "z planes / rows, all %d host threads" % (n, U, row_step, rays, samples, cores),
            "integrate_mvoxels_per_s": round(n ** 3 / t_int / 1e6, 2),
            "integrate_mvoxels_per_s_1thread": round(n ** 3 / t_int1 / 1e6, 2),
            "raycast_mrays_per_s": round(rays / t_ray / 1e6, 4),
            "bilateral_ms": round(t_bil * 1e3, 2),
            "wall_seconds_spent": round(t_bil + t_int + t_int1 * zs / n + probe + t_ray, 2),
            "cpu_seconds_spent": round((t_bil + t_int + probe + t_ray) * cores + t_int1 * zs / n, 1)}


if __name__ == "__main__":
    main()
