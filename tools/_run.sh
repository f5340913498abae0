cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04trk
dir=/tmp/tsdf_tum_trk
python - <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
from tsdf_amd import synth
synth.write_tum_directory("$dir", 26, seed=0x5EED0003, stream_frames=200)
PY
for i in 1 2 3; do build/kinfu_stream -d $dir -n 512 -k 24 --track | tail -1; python tools/dbg_tracking.py 2>&1 | grep "ms per frame"; done | tee gpurun_out/r04trk/track_cpp_vs_python.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_trk && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_trk -o run -- $GRAFT_REPO_ROOT/build/kinfu_stream -d $dir -n 512 -k 24 --track > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; db=$(find /tmp/prof_trk -name "*.db" | head -1); python tools/rocprof_summary.py timeline $db 380 70 | grep -B2 -A3 "brick_cull" | head -40 | cut -c1-105
