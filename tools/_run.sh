cd $GRAFT_REPO_ROOT
python -m pytest tests/test_pipeline.py tests/test_long_stream.py tests/test_weight_storage.py tests/test_cpp_stream.py -x -q 2>&1 | tail -2
for r in 1 2 3; do
for n in 0 1; do
echo stop_event $n "step" $(TSDF_PIPE_STOP_EVENT=$n python bench.py --no-cpu-baseline --no-parity --steps 20 --warmup 5 --repeats 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_runs'], d['last_frame_vertex_bits'])")
done; done
