cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04y
for rep in 1 2 3; do
for n in 0 1; do
TSDF_PIPE_HOST_WAIT=$n timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04y/c3_hw${n}_$rep.json 2>/dev/null
TSDF_PIPE_HOST_WAIT=$n timeout 600 python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04y/c4_hw${n}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04y/*hw*.json")):
    try:
        d=json.load(open(f)); print(f, d["ms_per_step"], d.get("ms_per_step_runs"), d.get("last_frame_vertex_bits"))
    except Exception as e: print(f, "ERR", e)
PY
