#!/bin/bash
# On the GPU box: the ray cast alone under library variants and pass budgets:  bash tools/ray_ab.sh <out dir> "<variants>" "<budgets>" [extra env]
out=$1; mkdir -p $out
for rep in 1 2; do
for v in $2; do for b in $3; do
  echo -n "$v budget $b: "
  env $4 TSDF_HIP_LIB=$PWD/build/variants/$v/libtsdf_hip.so TSDF_RAY_TRIP_BUDGET=$b timeout 120 python tools/dbg_ray_only.py 40 2>&1 | tail -1
done; done; done | tee -a $out/ray_ab.txt
