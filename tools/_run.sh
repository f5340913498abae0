cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04final
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r04final/pytest_gpu.log
cat gpurun_out/r04final/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r04final/smoke.log
bash tools/profile_round.sh r04zz 20 5 2>&1 | tail -2
bash tools/profile_round.sh r04zz_config4 20 5 "--workload config4" 1024 2>&1 | tail -2
bash tools/profile_round.sh r04zz_grid256 20 5 "--grid 256" 256 2>&1 | tail -2
cp profiles/traffic_r04zz*.json gpurun_out/r04final/ 2>/dev/null
