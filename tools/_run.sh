cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v
for rep in 1 2; do for n in 0 2; do
TSDF_PIPE_RELEASE=$n timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04v/c3_r${n}_$rep.json 2>/dev/null
TSDF_PIPE_RELEASE=$n timeout 600 python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04v/c4_r${n}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04v/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"] if "integrate" in d["roofline"]["kernel"] else d["roofline_other"]
        print(f, d["ms_per_step"], d.get("ms_per_step_runs"), r["avg_launch_ms"], d.get("last_frame_vertex_bits"))
    except Exception as e: print(f, "ERR", e)
PY
