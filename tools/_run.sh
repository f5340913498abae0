cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04x
for i in 1 2 3; do for pr in 1 0; do
echo prio$pr $(TSDF_RAY_LONG_PRIO=$pr python tools/dbg_ray_only.py 40 2>&1 | tail -1)
done; done | tee gpurun_out/r04x/ray_prio.txt
for rep in 1 2; do for n in 1 0; do
TSDF_RAY_LONG_PRIO=$n timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04x/c3_p${n}_$rep.json 2>/dev/null
TSDF_RAY_LONG_PRIO=$n timeout 600 python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04x/c4_p${n}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04x/*.json")):
    try:
        d=json.load(open(f)); ro=d["roofline_other"] if "integrate" in d["roofline"]["kernel"] else d["roofline"]
        print(f, d["ms_per_step"], d.get("ms_per_step_runs"), ro.get("avg_launch_ms_by_kernel"), d.get("last_frame_vertex_bits"))
    except Exception as e: print(f, "ERR", e)
PY
