#!/bin/bash
# On the GPU box: everything the round's judged set holds, on the sources as they are (profile sets of the three workloads -- kernel
# stats, FETCH / WRITE counters, the activity counters behind valu_busy / lanes_active, the bench line of the same arguments --, the GPU
# suite, smoke, the bench line with the driver's arguments and with default arguments).
# Copy gpurun_out/profiles_r06*/ and gpurun_out/r06z_* into profiles/ afterwards (profile_round.sh leaves profiles/traffic_<tag>.json itself).
bash tools/profile_round.sh r06 20 5 > gpurun_out/profile_round_r06.log 2>&1
bash tools/profile_round.sh r06_config4 20 5 "--workload config4" 1024 > gpurun_out/profile_round_r06_config4.log 2>&1
bash tools/profile_round.sh r06_grid256 20 5 "--grid 256" 256 > gpurun_out/profile_round_r06_grid256.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r06z_pytest_gpu.txt 2>&1; tail -1 gpurun_out/r06z_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06z_smoke.txt 2>&1; tail -1 gpurun_out/r06z_smoke.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06z_bench_driver_args.json 2> gpurun_out/b.err
python bench.py > gpurun_out/r06z_bench_default_args.json 2> gpurun_out/b2.err
python - <<'P'
import json
for f in ("gpurun_out/r06z_bench_driver_args.json", "gpurun_out/r06z_bench_default_args.json", "gpurun_out/profiles_r06/r06_bench.json", "gpurun_out/profiles_r06_config4/r06_config4_bench.json", "gpurun_out/profiles_r06_grid256/r06_grid256_bench.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r, o = d["roofline"], d["roofline_other"]
    print(f.split("/")[-1], d["ms_per_step"], d["parity"]["pass"], "|", r["kernel"][:24], r["frac"], r["traffic"], r.get("valu_busy"), r.get("lanes_active"), "|", o["kernel"][:24], o["frac"], o["traffic"], o.get("valu_busy"), o.get("lanes_active"), "| spread", d["step_ms_spread"]["max_over_median"], d["step_ms_spread"].get("max_over_median_without_trials"), d["step_ms_spread"].get("chooser_trial_steps"))
P
