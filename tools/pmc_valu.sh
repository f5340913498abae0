#!/bin/bash
root=$(pwd); out=$root/gpurun_out; mkdir -p $out/profiles_valu
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --no-cpu-baseline --no-parity --steps 40 --warmup 4"
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $out/prof_$c -o bench -- $cmd > $out/prof_$c.log 2>&1
  cd $root; python tools/rocprof_summary.py pmc $(find $out/prof_$c -name "*.db" | head -1) > $out/profiles_valu/r01o_pmc_$c.txt; cd /tmp
done
cd $root; for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES; do echo == $c; grep -i "integrate_kernel<false, false\|process_ray_kernel<false, false\|process_ray_tail\|bilateral" $out/profiles_valu/r01o_pmc_$c.txt | cut -c1-160; done
