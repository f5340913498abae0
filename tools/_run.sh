set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_parity_integrate.py tests/test_fuzz_parity.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04a/pytest.log
for v in new r03; do
  if [ $v = r03 ]; then export TSDF_HIP_LIB=$PWD/build/variants/r03/libtsdf_hip.so; else unset TSDF_HIP_LIB; fi
  timeout 600 python bench.py --workload config4 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only > gpurun_out/r04a/config4_$v.json 2> gpurun_out/r04a/config4_$v.err
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only > gpurun_out/r04a/config3_$v.json 2> gpurun_out/r04a/config3_$v.err
done
export TSDF_HIP_LIB=$PWD/build/variants/diag/libtsdf_hip.so
TSDF_DEBUG_BRICKS=3 timeout 600 python tools/dbg_config4_bricks.py > gpurun_out/r04a/bricks.log 2>&1
cat gpurun_out/r04a/pytest.log
