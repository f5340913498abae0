cd $GRAFT_REPO_ROOT
for r in 1 2; do for s in 0 1 2; do
TSDF_EVENT_SCOPE=$s python bench.py --no-cpu-baseline --steps 20 --warmup 5 --repeats 3 > gpurun_out/ev_s${s}_$r.json 2>gpurun_out/ev_err.txt || tail -3 gpurun_out/ev_err.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/ev_s${s}_$r.json").read().strip().splitlines()[-1])
print("scope $s run $r ms_per_step", d["ms_per_step"], d.get("ms_per_step_runs"), "parity", d.get("parity",{}).get("status") if isinstance(d.get("parity"),dict) else d.get("parity"), "bits", d.get("last_frame_vertex_bits"))
PY
done; done
for s in 0 1 2; do echo scope $s $(TSDF_EVENT_SCOPE=$s python tools/dbg_tracking.py 2>&1 | tail -1); done
