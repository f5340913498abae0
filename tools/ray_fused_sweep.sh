#!/bin/bash
# On the GPU box: the ray cast alone (tools/dbg_ray_only.py), two launches against the fused launch over pass budgets.
out=${1:-gpurun_out/r05c}; mkdir -p $out
for cfg in "0 22" "1 22" "1 16" "1 12" "1 8" "1 6" "1 4" "0 22" "1 12"; do
  set -- $cfg
  echo -n "fused $1 budget $2: "
  TSDF_RAY_FUSED=$1 TSDF_RAY_TRIP_BUDGET=$2 timeout 120 python tools/dbg_ray_only.py 40 2>&1 | tail -1
done | tee $out/fused_sweep.txt
