cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04s 20 5 2>&1 | tail -12
bash tools/profile_round.sh r04s_config4 20 5 "--workload config4" 1024 2>&1 | tail -8
bash tools/profile_round.sh r04s_grid256 20 5 "--grid 256" 256 2>&1 | tail -8
ls gpurun_out/profiles_r04s gpurun_out/profiles_r04s_config4 gpurun_out/profiles_r04s_grid256
