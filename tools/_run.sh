cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
for rep in 1 2 3; do for k in 1 8; do
  TSDF_PLACE_VERBOSE=1 TSDF_PLACE_TRIES=$k timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04g/config3_k${k}_$rep.json 2> gpurun_out/r04g/config3_k${k}_$rep.err
  grep "placement" gpurun_out/r04g/config3_k${k}_$rep.err
done; done
for k in 1 6; do
  TSDF_PLACE_VERBOSE=1 TSDF_PLACE_TRIES=$k timeout 600 python bench.py --workload config4 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --path-only --repeats 3 > gpurun_out/r04g/config4_k${k}.json 2> gpurun_out/r04g/config4_k${k}.err
  grep "placement" gpurun_out/r04g/config4_k${k}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04g/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"] if d["roofline"]["kernel"]=="integrate_kernel" else d["roofline_other"]
        print(f, d["ms_per_step"], r["avg_launch_ms"], r["frac"], r.get("measured_inplace_update_gbs"))
    except Exception as e: print(f, "ERR", e)
PY
