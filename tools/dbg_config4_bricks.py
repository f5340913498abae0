"""Integrate's brick statistics on the config-4 stream (1024^3, camera inside the volume), with a diagnostics build of the library:
    make DIAG=1 into another directory (or tools/ab_variants.sh-style), then
    TSDF_HIP_LIB=<that libtsdf_hip.so> TSDF_DEBUG_BRICKS=3 python tools/dbg_config4_bricks.py"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tsdf_amd, torch
from tsdf_amd import synth
n = 1024
v = tsdf_amd.TSDFVolume((n, n, n), (3000.,) * 3)
bil = tsdf_amd.BilateralFilter(30.0, 4.5)
s = torch.cuda.current_stream(); v.set_stream(s.cuda_stream)
for i in range(10):
    d, cam = synth.depth_frame(i, 100, seed=0x5EED0004, inside=True)
    f = d.copy(); bil.filter(f, 640, 480)
    fd = torch.from_numpy(f.astype(np.int16)).cuda()
    if i >= 8: print("frame", i, file=sys.stderr)
    v.integrate_device(fd.data_ptr(), 640, 480, cam)
    torch.cuda.synchronize()
