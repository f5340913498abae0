#!/bin/bash
# Counter passes over an arbitrary command (on the GPU box, via gpurun):
#   bash tools/pmc_cmd.sh <tag> "<command>" "<kernel name filter (grep -i)>" [pass definitions "name:C1 C2 C3" ...]
# Each pass is its own rocprofv3 run (--kernel-trace + --pmc only).  Summaries: gpurun_out/profiles_<tag>/<tag>_pmc_<name>.txt
tag=$1; cmd=$2; filt=$3; shift 3
root=$(pwd); out=$root/gpurun_out; mkdir -p $out/profiles_$tag
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  set -- "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
         "active:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
         "busy:GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
         "mem:TA_TA_BUSY_sum TA_BUSY_avr SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
fi
for p in "$@"; do
  name=${p%%:*}; ctrs=${p#*:}
  rm -rf $out/prof_$name
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs -d $out/prof_$name -o run -- $cmd > $out/prof_$name.log 2>&1
  db=$(find $out/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then (cd $root; python tools/rocprof_summary.py pmc $db > $out/profiles_$tag/${tag}_pmc_$name.txt); echo "== $name"; grep -i "$filt" $out/profiles_$tag/${tag}_pmc_$name.txt | cut -c1-175
  else echo "pass $name produced no database"; tail -5 $out/prof_$name.log; fi
  rm -rf $out/prof_$name
done
