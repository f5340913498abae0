#!/bin/bash
# Drop-in check of the C++ class surface: compile the REFERENCE's unchanged src/Tools/kinfu.cpp against this
# repo's headers (tsdf_amd/host/include) and link it with libtsdf_host.so / libtsdf_hip.so.
#
# kinfu.cpp includes "../include/<Name>.hpp", resolved relative to the including file, so a scratch tree is
# built where src/Tools/kinfu.cpp is a symlink to the reference file and src/include a symlink to our headers.
# The reference source is never copied into the repo.  Runs only where /root/reference is mounted.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
REF="${REF:-/root/reference}"
SRC="$REF/src/Tools/kinfu.cpp"
if [ ! -f "$SRC" ]; then echo "linkcheck: $SRC not present, skipped"; exit 0; fi
# Everything compiled from reference sources lands OUTSIDE this tree (TSDF_REF_BUILD, default /tmp/tsdf_ref_build): it is a check of
# this container's build and never travels to the GPU box.  tests/cpp/test_surface.cpp and tools/kinfu_stream.cpp drive
# the same call sequence on the GPU.
REFBUILD="${TSDF_REF_BUILD:-/tmp/tsdf_ref_build}"
W="$REFBUILD/linkcheck"
OUT="$W/bin"
rm -rf "$W"; mkdir -p "$W/src/Tools"
ln -s "$SRC" "$W/src/Tools/kinfu.cpp"
ln -s "$ROOT/tsdf_amd/host/include" "$W/src/include"
EIGEN=""
for d in /usr/include/eigen3 /usr/local/include/eigen3; do [ -f "$d/Eigen/Core" ] && EIGEN="-I$d" && break; done
[ -z "$EIGEN" ] && EIGEN="-I$ROOT/tsdf_amd/host/eigen_compat"
g++ -std=c++11 -O1 -w $EIGEN -I"$ROOT/include" -c "$W/src/Tools/kinfu.cpp" -o "$W/kinfu.o"
mkdir -p "$OUT"
g++ -o "$OUT/kinfu" "$W/kinfu.o" -L"$ROOT/tsdf_amd/lib" -ltsdf_host -ltsdf_hip -Wl,-rpath,"$ROOT/tsdf_amd/lib"
echo "linkcheck: reference kinfu.cpp compiled unchanged and linked -> $OUT/kinfu"
# the same for src/Tools/tsdf_icp.cpp (ICP between a saved volume and a depth image): it includes
# "ICP_CUDA/ICPOdometry.h" (ours, under tsdf_amd/host/third_party) and <sophus/se3.hpp> (real Sophus or the bundled subset)
ICPSRC="$REF/src/Tools/tsdf_icp.cpp"
if [ -f "$ICPSRC" ]; then
  ln -s "$ICPSRC" "$W/src/Tools/tsdf_icp.cpp"
  SOPHUS=""
  for d in /usr/include /usr/local/include; do [ -f "$d/sophus/se3.hpp" ] && SOPHUS="-I$d" && break; done
  [ -z "$SOPHUS" ] && SOPHUS="-I$ROOT/tsdf_amd/host/sophus_compat"
  g++ -std=c++11 -O1 -w $EIGEN $SOPHUS -I"$ROOT/include" -I"$ROOT/tsdf_amd/host/third_party" -c "$W/src/Tools/tsdf_icp.cpp" -o "$W/tsdf_icp.o"
  g++ -o "$OUT/tsdf_icp" "$W/tsdf_icp.o" -L"$ROOT/tsdf_amd/lib" -ltsdf_host -ltsdf_hip -Wl,-rpath,"$ROOT/tsdf_amd/lib"
  echo "linkcheck: reference tsdf_icp.cpp compiled unchanged and linked -> $OUT/tsdf_icp"
fi
# usage line only (no GPU needed): the binary must start and reject a bad command line like the reference
"$OUT/kinfu" 2>&1 | head -2 || true
