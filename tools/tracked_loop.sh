#!/bin/bash
# On the GPU box: the tracked loop of configs[4] at 512^3 from the C++ driver and from the Python mirror, three times each
out=${1:-gpurun_out/r05v}; mkdir -p $out
d=$(mktemp -d)
python -c "
import sys; sys.path.insert(0, '.')
from tsdf_amd import synth
synth.write_tum_directory('$d', 28, seed=0x5EED0003, stream_frames=200)"
{ echo "tracked loop of configs[4] at 512^3, 640x480, 24 timed frames after 4: C++ driver (tools/kinfu_stream.cpp --track) and Python mirror (tools/dbg_tracking.py), one box"
for r in 1 2 3; do build/kinfu_stream -d $d -n 512 -k 24 -w 4 --track 2>&1 | tail -1; python tools/dbg_tracking.py 28 | tail -1; done; } | tee $out/tracked_loop.txt
rm -rf $d
