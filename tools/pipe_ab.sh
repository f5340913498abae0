#!/bin/bash
# On the GPU box: the step under the pipeline's schedule knobs and the cell-parallel cast's (bench.py --path-only lines)
out=${1:-gpurun_out/r05p}; mkdir -p $out
run() { echo -n "$1 $2: "; env $1 timeout 300 python bench.py --steps 20 --warmup 5 --path-only --no-parity --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], d['ms_per_step_runs'], 'stages', d['stage_ms'], d['roofline']['kernel'][:20], d['roofline']['avg_launch_ms'], d.get('roofline_other',{}).get('avg_launch_ms'))"; }
for e in "TSDF_PIPE_RELEASE=0" "TSDF_PIPE_RELEASE=1" "TSDF_PIPE_RELEASE=2" "TSDF_PIPE_HOST_WAIT=1" "TSDF_RAY_CELLS_GRID=4096" "TSDF_RAY_CELLS_GRID=16384" "TSDF_RAY_CELLS=0" "TSDF_PIPE_RELEASE=0"; do run "$e" ""; done 2>&1 | tee $out/pipe_ab.txt
run "TSDF_RAY_CELLS=1" "--workload config4" | tee -a $out/pipe_ab.txt
run "TSDF_RAY_CELLS=1" "--grid 256" | tee -a $out/pipe_ab.txt
run "TSDF_RAY_CELLS=2" "--grid 256" | tee -a $out/pipe_ab.txt
