#!/bin/bash
# Per-kernel times of an arbitrary command (on the GPU box, via gpurun):  bash tools/stats_cmd.sh <tag> "<command>" [rows]
tag=$1; cmd=$2; rows=${3:-14}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out/profiles_$tag
cd /tmp && export TMPDIR=/tmp
rm -rf $out/prof_stats_$tag
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_stats_$tag -o run -- $cmd > $out/prof_stats_$tag.log 2>&1
db=$(find $out/prof_stats_$tag -name "*.db" | head -1)
cd $root
if [ -n "$db" ]; then python tools/rocprof_summary.py stats $db > $out/profiles_$tag/${tag}_kernel_stats.txt; head -$rows $out/profiles_$tag/${tag}_kernel_stats.txt | cut -c1-150
else echo "no database"; tail -5 $out/prof_stats_$tag.log; fi
rm -rf $out/prof_stats_$tag
