#!/bin/bash
# Memory-side counters of the in-place / out-of-place walks of tools/ubench_rmw.hip (why an in-place read-modify-write stops
# near 5 TB/s where a copy reaches 6.2):  bash tools/pmc_rmw.sh r03c   (on the GPU box, via gpurun)
# Separate rocprofv3 passes (--kernel-trace + --pmc only), 4 TCC counters each; summaries in gpurun_out/profiles_<tag>/.
tag=${1:-r03x}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out/profiles_$tag
$root/build/ubench_rmw > $out/profiles_$tag/${tag}_ubench_rmw.txt 2>&1
cat $out/profiles_$tag/${tag}_ubench_rmw.txt
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $out/prof_$name -o rmw -- $root/build/ubench_rmw > $out/prof_$name.log 2>&1
  db=$(find $out/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then (cd $root; python tools/rocprof_summary.py pmc $db > $out/profiles_$tag/${tag}_rmw_pmc_$name.txt); else echo "pass $name produced no database"; tail -5 $out/prof_$name.log; fi
  rm -rf $out/prof_$name
}
run rd TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_32B
run wr TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL
run l2 TCC_HIT TCC_MISS TCC_WRITEBACK TCC_TAG_STALL
cd $root
for g in rd wr l2; do echo "== $g"; sort -k1,1 -k2,2 $out/profiles_$tag/${tag}_rmw_pmc_$g.txt | cut -c1-150 | head -70; done
