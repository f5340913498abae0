#!/bin/bash
# On the GPU box: march kernels against the cell-parallel cast over grid sizes (a voxel's footprint in pixels doubles as the grid halves):
# where TSDF_RAY_CELLS_FOOTPRINT belongs, and the pairs a listed part should hold.  bench.py --path-only lines.
out=${1:-gpurun_out/r05q}; mkdir -p $out
run() { echo -n "$1 $2: "; env $1 timeout 300 python bench.py --steps 20 --warmup 5 --path-only --no-parity --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'raycast stage', d['stage_ms']['raycast'], 'integrate stage', d['stage_ms']['integrate'])"; }
for g in 64 96 128 192 256 384; do for m in "TSDF_RAY_CELLS=0" "TSDF_RAY_CELLS=2" "TSDF_RAY_CELLS=2 TSDF_RAY_CELLS_PAIRS=256" "TSDF_RAY_CELLS=2 TSDF_RAY_CELLS_PAIRS=4096"; do run "$m" "--grid $g"; done; done 2>&1 | tee $out/cells_footprint_sweep.txt
