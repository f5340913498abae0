#!/bin/bash
# The C++ driver (tools/kinfu_stream.cpp, no Python) and bench.py on the SAME frames, same box, alternating:
#   bash tools/compare_drivers.sh [steps=20] [warmup=5] [grid=512] [reps=3]
# Writes a synthetic TUM-layout directory (the bench stream, 200-frame trajectory, as many frames as the run needs), then per repetition
# one line of each driver: ms per step and the checksum of the last picture (sum of the 32-bit patterns of the vertex map).
steps=${1:-20}; warmup=${2:-5}; grid=${3:-512}; reps=${4:-3}
root=$(cd "$(dirname "$0")/.." && pwd)
dir=${TMPDIR:-/tmp}/tsdf_tum_$$
python - <<PY
import sys; sys.path.insert(0, "$root")
from tsdf_amd import synth
synth.write_tum_directory("$dir", $steps + $warmup + 1, seed=0x5EED0003, stream_frames=200)
PY
for r in $(seq $reps); do
    "$root/build/kinfu_stream" -d "$dir" -n $grid -k $steps -w $warmup | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kinfu_stream (C++)  %.4f ms/step  vertex bits %d  hits %d' % (d['ms_per_step'], d['last_frame_vertex_bits'], d['last_frame_hits']))"
    python "$root/bench.py" --tum-dir "$dir" --grid $grid --steps $steps --warmup $warmup --path-only --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench.py (Python)    %.4f ms/step  vertex bits %d' % (d['ms_per_step'], d['last_frame_vertex_bits']))"
done
rm -rf "$dir"
