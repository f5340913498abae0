cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04p/bench_c3.json 2> gpurun_out/r04p/bench_c3.err
for rep in 1 2; do for n in 0 8 16; do
TSDF_WEIGHT_PACK=$n timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04p/c3_p${n}_$rep.json 2>/dev/null
TSDF_WEIGHT_PACK=$n timeout 600 python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04p/c4_p${n}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04p/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"] if "integrate" in d["roofline"]["kernel"] else d["roofline_other"]
        print(f, d["ms_per_step"], d.get("ms_per_step_runs"), r["kernel"], r["avg_launch_ms"], r["frac"], d["stage_ms"]["integrate"], d.get("parity"))
    except Exception as e: print(f, "ERR", e)
PY
