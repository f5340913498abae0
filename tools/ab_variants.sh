#!/bin/bash
# A/B of kernel variants on ONE box (boxes differ by 10-15 % in memory throughput, so timings from different gpurun calls do not
# compare).  Here (build container):   bash tools/ab_variants.sh build <name> [git-rev]   compiles tsdf_amd/csrc of the working tree
# (or of a revision) into build/variants/<name>/libtsdf_hip.so.   On the GPU box:   bash tools/ab_variants.sh run "<command>" <name> ...
# (EXTRA="-DTSDF_X=0" adds compiler flags to a build.)  run: the command once per variant (TSDF_HIP_LIB picks the library, a debug hook of tsdf_amd/_capi.py) under rocprofv3 --stats.
mode=$1; shift
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)
if [ "$mode" = build ]; then
  name=$1; rev=$2; dst=$root/build/variants/$name; mkdir -p $dst
  src=$root
  if [ -n "$rev" ]; then src=$(mktemp -d); git -C $root archive $rev tsdf_amd/csrc include | tar -x -C $src; fi
  objs=""
  for f in $(cd $src/tsdf_amd/csrc && ls *.hip | sed "s/.hip//"); do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$src/include -I$src/tsdf_amd/csrc -Wall -Wno-unused-function $EXTRA $( [ $f = integrate_packed ] && echo "-mllvm -amdgpu-sched-strategy=max-ilp" ) \
        -c $src/tsdf_amd/csrc/$f.hip -o $dst/$f.o 2>&1 | grep -E "error" ; objs="$objs $dst/$f.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $dst/libtsdf_hip.so $objs -ldl && rm -f $dst/*.o && ls -la $dst/libtsdf_hip.so
else
  cmd=$1; shift
  for name in "$@" "$@"; do
    export TSDF_HIP_LIB=$root/build/variants/$name/libtsdf_hip.so
    echo "== $name"; bash $root/tools/stats_cmd.sh ab_$name "$cmd" 12 | grep -E "integrate_k|cull|bilateral_k|process_ray|tile_max|reach|resolve|cast_cells|cell_cast"
  done
fi
