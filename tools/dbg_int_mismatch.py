"""Debug aid: where do two settings of the integrate kernels differ?  (GPU box)  python tools/dbg_int_mismatch.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, subprocess, json
if len(sys.argv) > 1:   # child: integrate and dump
    import tsdf_amd
    from tsdf_amd import synth
    planes = 32
    size, phys = (72, 20, planes), (2700.0, 750.0, planes * 37.5)
    frames = [synth.depth_frame(i, 8, seed=5) for i in (1, 4)]
    v = tsdf_amd.TSDFVolume(size, phys)
    if os.environ.get('CNT'): v.set_counting(True)
    for d, cam in frames:
        v.integrate(d, 640, 480, cam)
    np.save(sys.argv[1], v.get_weight_data())
    np.save(sys.argv[1] + "_depth", frames[1][0])
    np.save(sys.argv[1] + "_ip", np.asarray(frames[1][1].inverse_pose(), np.float32))
    np.save(sys.argv[1] + "_k", np.asarray(frames[1][1].k(), np.float32))
    sys.exit(0)
for fast in ("1", "0"):
    env = dict(os.environ, TSDF_INT_FAST=fast)
    subprocess.check_call([sys.executable, __file__, "/tmp/w%s.npy" % fast], env=env)
a, b = np.load("/tmp/w1.npy"), np.load("/tmp/w0.npy")
depth = np.load("/tmp/w1.npy_depth.npy").reshape(480, 640)
ip = np.load("/tmp/w1.npy_ip.npy").reshape(-1); k = np.load("/tmp/w1.npy_k.npy").reshape(-1)
bad = np.flatnonzero(a != b)
print("differ:", bad.size, "of", a.size)
X, Y = 72, 20
vs = np.float32(37.5)
M = ip.reshape(4, 4).T if ip.size == 16 else None   # column-major float[16]
K = k.reshape(3, 3).T
for i in bad[:40]:
    z, r = divmod(int(i), X * Y); y, x = divmod(r, X)
    c = np.array([(x + 0.5) * 37.5, (y + 0.5) * 37.5, (z + 0.5) * 37.5, 1.0])
    cam = M @ c
    im = K @ cam[:3]
    px, py = im[0] / im[2], im[1] / im[2]
    rx, ry = int(np.round(px)), int(np.round(py))
    d = depth[ry, rx] if 0 <= rx < 640 and 0 <= ry < 480 else -1
    nb = depth[max(ry-1,0):ry+2, max(rx-2,0):rx+3] if d >= 0 else None
    print("voxel", (x, y, z), "fast w", a[i], "old w", b[i], "pixel %.3f %.3f ->" % (px, py), (rx, ry), "depth", d, "camz %.1f" % cam[2])
    if nb is not None: print(nb)
