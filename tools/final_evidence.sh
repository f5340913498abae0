#!/bin/bash
# On the GPU box: everything the round's judged set holds, on the sources as they are (profiles of the three workloads, counters,
# the GPU suite by default and with the cell-parallel cast forced on, smoke, the bench line with default arguments).
# Copy gpurun_out/profiles_r05*/ and gpurun_out/r05z_* into profiles/ afterwards.
bash tools/profile_round.sh r05 20 5 > gpurun_out/profile_round_r05.log 2>&1
bash tools/pmc_cmd.sh r05 "python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --path-only --repeats 1 --steps 20 --warmup 5" "kernel" \
    "insts:SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES" \
    "active:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" > gpurun_out/pmc_r05.log 2>&1
bash tools/profile_round.sh r05_config4 20 5 "--workload config4" 1024 > gpurun_out/profile_round_r05_config4.log 2>&1
bash tools/profile_round.sh r05_grid256 20 5 "--grid 256" 256 > gpurun_out/profile_round_r05_grid256.log 2>&1
( timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r05z_pytest_gpu.txt 2>&1; tail -1 gpurun_out/r05z_pytest_gpu.txt
( TSDF_RAY_CELLS=2 timeout 1500 python -m pytest tests/test_parity_raycast.py tests/test_fuzz_parity.py tests/test_multi_slab.py tests/test_pipeline.py -m gpu -x -q ) > gpurun_out/r05z_pytest_cells_forced.txt 2>&1; tail -1 gpurun_out/r05z_pytest_cells_forced.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z_smoke.txt 2>&1; tail -1 gpurun_out/r05z_smoke.txt
python bench.py > gpurun_out/r05z_bench_default_args.json 2> gpurun_out/b.err
python - <<'P'
import json
for f in ("gpurun_out/r05z_bench_default_args.json", "gpurun_out/profiles_r05/r05_bench.json", "gpurun_out/profiles_r05_config4/r05_config4_bench.json", "gpurun_out/profiles_r05_grid256/r05_grid256_bench.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["ms_per_step"], d["parity"]["pass"], d["roofline"]["kernel"][:24], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline_other"]["kernel"][:24], d["roofline_other"]["frac"], d["roofline_other"]["traffic"])
P
