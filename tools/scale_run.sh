#!/bin/bash
# The 1/2/4/8-GPU curve of bench.py in one go, on a node with 8 MI355X (the driver's SCALE run does the same):
#   bash tools/scale_run.sh [steps] [warmup] [extra bench.py flags ...]     e.g.  bash tools/scale_run.sh 20 5 --workload config4
# One JSON line per N lands in gpurun_out/scale_N<N>.json.  HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only
# supports dmabuf IPC (RCCL / tensor sharing across processes fails without it).
steps=${1:-20}; warmup=${2:-5}; shift 2 2>/dev/null
export HSA_ENABLE_IPC_MODE_LEGACY=0
root=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd); out=$root/gpurun_out; mkdir -p $out
have=$(python -c "import torch; print(torch.cuda.device_count())")
for n in 1 2 4 8; do
  if [ "$n" -gt "$have" ] && [ -z "$TSDF_BENCH_SHARE_GPU" ]; then echo "N=$n skipped: $have GPU(s) visible"; continue; fi
  port=$((29500 + n))
  if [ "$n" -eq 1 ]; then
    timeout 900 python $root/bench.py --gpus 1 --steps $steps --warmup $warmup "$@" > $out/scale_N$n.json 2> $out/scale_N$n.err
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port \
        $root/bench.py --gpus $n --steps $steps --warmup $warmup "$@" > $out/scale_N$n.json 2> $out/scale_N$n.err
  fi
  echo "N=$n rc=$? $(tail -c 400 $out/scale_N$n.json | head -c 400)"
done
python - <<PY
import json, glob, os
rows = []
for n in (1, 2, 4, 8):
    p = os.path.join("$out", "scale_N%d.json" % n)
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
        rows.append((n, d["value"], d["ms_per_step"], d.get("parity", {}).get("pass")))
    except Exception as e:
        pass
for n, v, ms, ok in rows:
    print("N=%d  %.0f Mvoxels/s  %.4f ms/step  x%.2f vs N=1  parity %s" % (n, v, ms, v / rows[0][1], ok))
PY
