cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_parity_raycast.py tests/test_fuzz_parity.py tests/test_pipeline.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do for n in new old; do
if [ $n = new ]; then unset TSDF_HIP_LIB; else export TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/volold/libtsdf_hip.so; fi
timeout 600 python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ro=d['roofline_other'] if 'integrate' in d['roofline']['kernel'] else d['roofline']; print('c4 $n', d['ms_per_step'], d['ms_per_step_runs'], ro.get('avg_launch_ms_by_kernel'), d['last_frame_vertex_bits'])"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $n', d['ms_per_step'], d['ms_per_step_runs'], d['last_frame_vertex_bits'])"
done; done
