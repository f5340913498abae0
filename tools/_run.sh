cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
for rep in 1 2 3; do for n in base rayprio; do
if [ $n = base ]; then unset TSDF_HIP_LIB; else export TSDF_HIP_LIB=$GRAFT_REPO_ROOT/build/variants/$n/libtsdf_hip.so; fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r05a/c3_${n}_$rep.json 2>/dev/null
timeout 600 python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r05a/c4_${n}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05a/*.json")):
    try:
        d=json.load(open(f)); print(f, d["ms_per_step"], d.get("ms_per_step_runs"), d.get("last_frame_vertex_bits"))
    except Exception as e: print(f, "ERR", e)
PY
