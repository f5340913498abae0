#!/bin/bash
# Profiles of one round, run on the GPU box through gpurun:  bash tools/profile_round.sh r02b [steps] [warmup]
# Writes the rocprofv3 databases under gpurun_out/ (scratch) and the judged summaries under gpurun_out/profiles_<tag>/
# (copy them into profiles/ afterwards; <tag>_traffic.json also becomes profiles/traffic.json, which bench.py reports as
# `roofline.traffic` when its kernel-source digest and bench arguments match the run).  Counter passes are separate runs with
# --kernel-trace only (no --stats mixed with tracing domains), each under its own timeout.  Default arguments = the driver's
# round-end run (--steps 20 --warmup 5), so the per-launch PMC averages cover the launches bench.py prices.
# Optional 4th / 5th argument: extra bench arguments and the grid they imply, e.g.  bash tools/profile_round.sh r04b_config4 20 5 "--workload config4" 1024
tag=${1:-rXX}; steps=${2:-20}; warmup=${3:-5}; extra=${4:-}; grid=${5:-512}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out/profiles_$tag
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --no-cpu-baseline --no-parity --path-only --repeats 1 --spread-steps 0 --steps $steps --warmup $warmup $extra"
rm -rf $out/prof_stats $out/prof_fetch $out/prof_write $out/prof_activity
timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_stats -o bench -- $cmd > $out/prof_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/prof_fetch -o bench -- $cmd > $out/prof_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/prof_write -o bench -- $cmd > $out/prof_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/prof_activity -o bench -- $cmd > $out/prof_activity.log 2>&1
cd $root
f() { find $out/$1 -name "*.db" | head -1; }
sha=$(python -c "import bench; print(bench.kernel_source_sha())")
python tools/rocprof_summary.py stats $(f prof_stats) > $out/profiles_$tag/${tag}_kernel_stats.txt
python tools/rocprof_summary.py pmc $(f prof_fetch) > $out/profiles_$tag/${tag}_pmc_FETCH_SIZE.txt
python tools/rocprof_summary.py pmc $(f prof_write) > $out/profiles_$tag/${tag}_pmc_WRITE_SIZE.txt
python tools/rocprof_summary.py pmc $(f prof_activity) > $out/profiles_$tag/${tag}_pmc_activity.txt
python tools/rocprof_summary.py traffic+activity $(f prof_fetch) $(f prof_write) $(f prof_activity) $warmup $steps tag=$tag kernel_source_sha=$sha \
    "bench_args={\"steps\": $steps, \"warmup\": $warmup, \"grid\": $grid, \"gpus\": 1}" > $out/profiles_$tag/${tag}_traffic.json
cp $out/profiles_$tag/${tag}_traffic.json profiles/traffic_${tag}.json   # (so that the bench line below carries `traffic`; commit it with the summaries)
timeout 900 python bench.py --steps $steps --warmup $warmup $extra > $out/profiles_$tag/${tag}_bench.json 2> $out/bench.err
tail -c 600 $out/profiles_$tag/${tag}_kernel_stats.txt
# the raw databases are scratch (gpurun copies back at most 64 MiB): keep only the summaries
rm -rf $out/prof_stats $out/prof_fetch $out/prof_write $out/prof_activity
