#!/bin/bash
# The 1/2/4/8-GPU curves of bench.py in one go, on a node with 8 MI355X (the driver's SCALE run does the same for the default
# workload):   bash tools/scale_run.sh [steps] [warmup] [extra bench.py flags ...]
# Both workloads are run: config3 (BASELINE configs[2], 512^3 -- the metric's configuration) and config4 (configs[3], 1024^3 --
# the one that motivates sharding); WORKLOADS="config4" restricts it.  One JSON line per (workload, N) lands in
# gpurun_out/scale_<workload>_N<N>.json.  TSDF_PIPE_EXCHANGE_STREAM=1 puts the all-gather + merge on a third stream (the first
# thing to A/B on a real node).  HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC
# (RCCL / tensor sharing across processes fails without it).
steps=${1:-20}; warmup=${2:-5}; shift 2 2>/dev/null
export HSA_ENABLE_IPC_MODE_LEGACY=0
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd); out=$root/gpurun_out; mkdir -p $out
have=$(python -c "import torch; print(torch.cuda.device_count())")
for wl in ${WORKLOADS:-config3 config4}; do
  for n in 1 2 4 8; do
    if [ "$n" -gt "$have" ] && [ -z "$TSDF_BENCH_SHARE_GPU" ]; then echo "$wl N=$n skipped: $have GPU(s) visible"; continue; fi
    port=$((29500 + n)); f=$out/scale_${wl}_N$n
    # (bench.py --gpus N started plainly re-executes itself under torch.distributed.run, one rank per GPU, 127.0.0.1 rendezvous)
    timeout 1200 python $root/bench.py --gpus $n --steps $steps --warmup $warmup --workload $wl "$@" > $f.json 2> $f.err
    echo "$wl N=$n rc=$? $(tail -c 300 $f.json | head -c 300)"
  done
done
python - <<PY
import json, os
for wl in "${WORKLOADS:-config3 config4}".split():
    rows = []
    for n in (1, 2, 4, 8):
        p = os.path.join("$out", "scale_%s_N%d.json" % (wl, n))
        try:
            d = json.loads(open(p).read().strip().splitlines()[-1])
            rows.append((n, d["value"], d["ms_per_step"], d.get("parity", {}).get("pass"), (d.get("stage_ms") or {}).get("exchange")))
        except Exception:
            pass
    for n, v, ms, ok, ex in rows:
        print("%s N=%d  %.0f Mvoxels/s  %.4f ms/step  x%.2f vs N=1  exchange stage %s ms  parity %s" % (wl, n, v, ms, v / rows[0][1], ex, ok))
PY
