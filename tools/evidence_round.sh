#!/bin/bash
# Everything the round's DESIGN / README figures rest on, in one go on the GPU box:   bash tools/evidence_round.sh r03h
# Summaries land in gpurun_out/profiles_<tag>/ (copy them into profiles/).
tag=${1:-r03x}
root=$(pwd); out=$root/gpurun_out/profiles_$tag; mkdir -p $out
bash tools/profile_round.sh $tag 20 5 > $out/${tag}_profile_round.log 2>&1            # kernel stats, FETCH / WRITE traffic, the driver-shaped bench line
bash tools/pmc_bound.sh $tag > $out/${tag}_pmc_bound.log 2>&1                          # clock / instruction / activity counters
python bench.py > $out/${tag}_bench_default.json 2> /dev/null                         # the default run (100 steps)
python bench.py --workload config4 --steps 20 --warmup 5 > $out/${tag}_config4_n1_bench.json 2> /dev/null
python tools/dbg_ray_only.py 40 > $out/${tag}_ray_only.txt 2>&1
python tools/dbg_ray_work.py 40 > $out/${tag}_ray_work.txt 2>&1
python tools/dbg_pipeline_kernels.py 100 > $out/${tag}_pipeline_kernels.txt 2>&1
python tools/dbg_slab_scaling.py config3 > $out/${tag}_slab_scaling_config3.txt 2>&1
python tools/dbg_slab_scaling.py config4 > $out/${tag}_slab_scaling_config4.txt 2>&1
bash tools/compare_drivers.sh > $out/${tag}_cpp_driver_vs_bench.txt 2>&1
ls $out
