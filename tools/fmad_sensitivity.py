"""What a real CUDA build of the reference may differ by: nvcc's default -fmad=true contracts a multiply feeding an add into one
fused multiply-add, the oracle (and the HIP kernels, built -ffp-contract=off) round each operation on its own.  The reference holds
no vectors for integrate / ray cast (parity unpinned), so this is the only further evidence this container can give:
the SAME restatement compiled with gcc -ffp-contract=fast -mfma (oracle/libtsdf_oracle_fmad.so: 146 fused operations -- the
projection sums of world_to_pixel / world_to_camera, (D w + tsdf) / w', start + t dir, the trilinear blend) run beside the exact
one on BASELINE configs[0] and configs[1], counting what the decisions downstream (round(), the frustum test, sdf >= -trunc,
tsdf <= 0: SURVEY.md H2) turn a last-bit difference into.  (The reference's own makefiles build with -G, which switches the
contraction off in nvcc: its shipped binaries are expected to match the UNFUSED arithmetic.)

    python tools/fmad_sensitivity.py            # both configurations, the table of DESIGN.md 4 / INTEGRATION.md
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

W, H = 640, 480


def compare(n, frames, cast_from, threads=None, filtered=False):
    """frames: [(depth, camera)] integrated into two n^3 / 3000 mm oracle volumes (exact, fused); cast_from: camera of the ray cast.
    -> dict of counts."""
    import oracle as O
    threads = threads or O.max_threads()
    vols = {}
    for name in ("exact", "fmad"):
        with O.variant(name):
            v = O.Volume((n, n, n), (3000.0,) * 3)
            for d, cam in frames:
                f = O.bilateral_u16(d, W, H, 30.0, 4.5, nthreads=threads).reshape(-1) if filtered else d
                v.integrate(f, W, H, cam.inverse_pose(), cam.k(), cam.kinv(), nthreads=threads)
            V, N = v.raycast(W, H, cast_from.pose(), cast_from.kinv(), nthreads=threads)
            vols[name] = (v, V, N)
    (a, Va, Na), (b, Vb, Nb) = vols["exact"], vols["fmad"]
    upd = (a.weight > 0) | (b.weight > 0)
    wdiff = a.weight != b.weight
    same_w = upd & ~wdiff
    rel = np.zeros(a.dist.shape, np.float64)
    rel[same_w] = np.abs(a.dist[same_w].astype(np.float64) - b.dist[same_w]) / np.maximum(np.abs(a.dist[same_w]), 1e-6)
    hit_a, hit_b = ~np.isnan(Va[:, 0]), ~np.isnan(Vb[:, 0])
    both = hit_a & hit_b
    vrel = np.zeros(Va.shape[0])
    if both.any():
        vrel[both] = (np.abs(Va[both].astype(np.float64) - Vb[both]) / np.maximum(np.abs(Va[both]), 1e-6)).max(axis=1)
    vabs = np.zeros(Va.shape[0])
    if both.any():
        vabs[both] = np.linalg.norm(Va[both].astype(np.float64) - Vb[both], axis=1)
    step = float(np.float32(np.float64(np.float32(a.truncation_distance())) * 0.05))
    return {
        "grid": n, "frames": len(frames), "voxels_updated": int(upd.sum()),
        "weight_mismatches": int(wdiff.sum()),                               # a voxel updated in one build and not in the other, some frame
        "distance_bits_differ": int((same_w & (a.dist.view(np.uint32) != b.dist.view(np.uint32))).sum()),
        "distance_beyond_1e-4_relative": int((rel > 1e-4).sum()), "distance_max_relative": float(rel.max()),
        "rays": int(Va.shape[0]), "hits": int(hit_a.sum()), "nan_mask_flips": int((hit_a != hit_b).sum()),
        "vertex_bits_differ": int((both & (Va.view(np.uint32) != Vb.view(np.uint32)).any(axis=1)).sum()),
        "vertices_beyond_1e-4_relative": int((vrel > 1e-4).sum()), "vertex_max_relative": float(vrel.max()),
        "vertices_a_sample_or_more_apart": int((vabs > 0.9 * step).sum()), "vertex_max_mm": float(vabs.max()), "step_mm": step,
    }


def config1():
    """BASELINE configs[0]: 128^3, one synthetic frame, ground-truth pose."""
    from tests.helpers import camera_at
    from tsdf_amd import synth
    cam = camera_at((1500, 1500, -1000))
    return compare(128, [(synth.config1_depth(), cam)], cam)


def config2(n_frames=50):
    """BASELINE configs[1]: 256^3, the first 50 frames of the surrogate stream, ray cast from pose 0."""
    from tsdf_amd import synth
    frames = [synth.depth_frame(i, 50, seed=0x5EED0002) for i in range(n_frames)]
    return compare(256, frames, frames[0][1])


if __name__ == "__main__":
    for name, r in (("configs[0] (128^3, 1 frame)", config1()), ("configs[1] (256^3, 50 frames)", config2())):
        print(name)
        for k, v in r.items():
            print("    %-36s %s" % (k, ("%.3g" % v) if isinstance(v, float) else v))
