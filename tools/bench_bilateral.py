#!/usr/bin/env python3
"""Time the bilateral filter alone (HIP events on the launch stream) and check it against the oracle on the same frames.
    python tools/bench_bilateral.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import oracle as O
import tsdf_amd
from tsdf_amd import synth

W, H = 640, 480
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
frames = [synth.depth_frame(i, 200, seed=0x5EED0003)[0] for i in range(4)]
dev = [torch.from_numpy(f.view(np.int16)).cuda() for f in frames]
out = torch.empty_like(dev[0])
f = tsdf_amd.BilateralFilter(30.0, 4.5)
s = torch.cuda.current_stream()
ok = True
for i, d in enumerate(dev):
    f.filter_device(d.data_ptr(), out.data_ptr(), W, H, bits=16, stream=s.cuda_stream)
    torch.cuda.synchronize()
    exp = O.bilateral_u16(frames[i], W, H, 30.0, 4.5, nthreads=O.max_threads()).reshape(-1)
    ok = ok and np.array_equal(out.cpu().numpy().view(np.uint16), exp)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for r in range(reps):
    f.filter_device(dev[r % 4].data_ptr(), out.data_ptr(), W, H, bits=16, stream=s.cuda_stream)
e1.record(s)
torch.cuda.synchronize()
print("%.2f us per 640x480 frame, parity %s" % (e0.elapsed_time(e1) * 1e3 / reps, ok))
