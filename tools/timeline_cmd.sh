#!/bin/bash
# Kernel timeline of a command (start order, durations, gaps):  bash tools/timeline_cmd.sh <tag> "<command>" [first] [count]   (on the GPU box)
tag=$1; cmd=$2; first=${3:-0}; count=${4:-80}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out/profiles_$tag
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_tl_$tag
timeout 300 rocprofv3 --kernel-trace -d $out/prof_tl_$tag -o run -- $cmd > $out/prof_tl_$tag.log 2>&1
db=$(find $out/prof_tl_$tag -name "*.db" | head -1)
cd $root
if [ -n "$db" ]; then python tools/rocprof_summary.py timeline $db $first $count > $out/profiles_$tag/${tag}_timeline.txt; cat $out/profiles_$tag/${tag}_timeline.txt | cut -c1-110
else echo "no database"; tail -5 $out/prof_tl_$tag.log; fi
rm -rf $out/prof_tl_$tag
