cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04h 20 5 > gpurun_out/r04h.log 2>&1
bash tools/profile_round.sh r04h_config4 20 5 "--workload config4 --no-parity --no-cpu-baseline" 1024 > gpurun_out/r04h_config4.log 2>&1
bash tools/profile_round.sh r04h_grid256 20 5 "--grid 256 --stream-frames 50 --no-cpu-baseline" 256 > gpurun_out/r04h_grid256.log 2>&1
ls gpurun_out/profiles_r04h*; tail -3 gpurun_out/r04h*.log
