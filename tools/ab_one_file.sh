#!/bin/bash
# A variant library that differs from the working tree's build in ONE source file (fast: one hipcc call + a link):
#   bash tools/ab_one_file.sh <name> <file without .hip> "<extra compiler flags>"   -> build/variants/<name>/libtsdf_hip.so
# (the other objects are the tree's own tsdf_amd/csrc/*.o: run `make hip` first).  Run side by side with tools/ab_variants.sh run.
name=$1; file=$2; extra=$3
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd); dst=$root/build/variants/$name; mkdir -p $dst
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$root/include -I$root/tsdf_amd/csrc -Wall -Wno-unused-function $extra \
    $( [ $file = integrate_packed ] && echo "-mllvm -amdgpu-sched-strategy=max-ilp" ) -c $root/tsdf_amd/csrc/$file.hip -o $dst/$file.o || exit 1
objs=""
for f in $(cd $root/tsdf_amd/csrc && ls *.hip | sed "s/.hip//"); do
  if [ $f = $file ]; then objs="$objs $dst/$f.o"; else objs="$objs $root/tsdf_amd/csrc/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $dst/libtsdf_hip.so $objs -ldl && rm -f $dst/$file.o && ls -la $dst/libtsdf_hip.so
