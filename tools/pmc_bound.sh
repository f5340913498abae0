#!/bin/bash
# What binds the hot kernels, from counters instead of inference:  bash tools/pmc_bound.sh r02a   (on the GPU box, via gpurun)
# Separate rocprofv3 passes (--kernel-trace + --pmc only; never mixed with other trace domains), each under its own timeout.
# Summaries land in gpurun_out/profiles_<tag>/<tag>_pmc_<group>.txt; copy the ones to be judged into profiles/.
#   clock  : GRBM_GUI_ACTIVE (shader-engine cycles while busy -> sustained clock = cycles / duration), SQ_BUSY_CYCLES, SQ_WAVES
#   insts  : SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS (wave-instructions issued)
#   active : SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES (quad-cycles, per the microarch guide)
tag=${1:-r02x}
root=$(pwd); out=$root/gpurun_out; mkdir -p $out/profiles_$tag
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --no-cpu-baseline --no-parity --path-only --repeats 1 --steps 40 --warmup 8"
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $out/prof_$name -o bench -- $cmd > $out/prof_$name.log 2>&1
  db=$(find $out/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then (cd $root; python tools/rocprof_summary.py pmc $db > $out/profiles_$tag/${tag}_pmc_$name.txt); else echo "pass $name produced no database"; tail -5 $out/prof_$name.log; fi
  rm -rf $out/prof_$name   # raw database: scratch (gpurun copies back at most 64 MiB)
}
run clock GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS
run active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
cd $root
for g in clock insts active; do echo "== $g"; grep -i "integrate_kernel<false, false\|process_ray_kernel<false, false\|process_ray_tail\|bilateral_kernel" $out/profiles_$tag/${tag}_pmc_$g.txt | cut -c1-170; done
