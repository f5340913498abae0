cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_bilateral.py tests/test_fuzz_parity.py -k "bilateral" -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do
echo new $(python tools/bench_bilateral.py 2>&1 | tail -1)
echo old $(TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/bilold/libtsdf_hip.so python tools/bench_bilateral.py 2>&1 | tail -1)
done
for rep in 1 2 3; do for n in new old; do
if [ $n = new ]; then unset TSDF_HIP_LIB; else export TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/bilold/libtsdf_hip.so; fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d['ms_per_step_runs'], d['last_frame_vertex_bits'])"
done; done
