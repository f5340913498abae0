cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/final/pytest_gpu.txt 2>&1; tail -3 gpurun_out/final/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
