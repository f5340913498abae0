cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/final/pytest_gpu.txt 2>&1; tail -3 gpurun_out/final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
bash tools/profile_round.sh r04zz 20 5 > gpurun_out/final/pr1.log 2>&1
bash tools/profile_round.sh r04zz_config4 20 5 "--workload config4" 1024 > gpurun_out/final/pr2.log 2>&1
bash tools/profile_round.sh r04zz_grid256 20 5 "--grid 256" 256 > gpurun_out/final/pr3.log 2>&1
for t in r04zz r04zz_config4 r04zz_grid256; do python - <<PY
import json
d=json.loads(open("gpurun_out/profiles_$t/${t}_bench.json").read().strip().splitlines()[-1])
print("$t", d["ms_per_step"], d.get("ms_per_step_runs"), d["value"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("tracking",{}).get("ms_per_frame"))
PY
done
