#!/bin/bash
# Registers, scratch and LDS of every kernel in a HIP object:  bash tools/kernel_regs.sh tsdf_amd/csrc/raycast.o [name filter]
obj=$1; filt=${2:-.}
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $obj $tmp/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | awk '
/\.name:/ {name=$2}
/\.vgpr_count:/ {v=$2}
/\.sgpr_count:/ {s=$2}
/\.private_segment_fixed_size:/ {p=$2}
/\.group_segment_fixed_size:/ {l=$2}
/\.vgpr_spill_count:/ {sp=$2}
/\.wavefront_size:/ {print name, "vgpr", v, "sgpr", s, "scratch", p, "lds", l, "spill", sp}' | grep -E "$filt" | sed 's/_ZN4tsdf//' | cut -c1-200
[ -n "$KEEP" ] && cp $tmp/dev.co $KEEP
rm -rf $tmp
