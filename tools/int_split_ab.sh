#!/bin/bash
# On the GPU box: integrate_packed_kernel's workgroups per brick (TSDF_INT_SPLIT) at 256^3 and 512^3: bench.py --path-only lines
out=${1:-gpurun_out/r05i}; mkdir -p $out
for cfg in "256 1" "256 2" "256 4" "256 1" "256 4" "512 1" "512 2" "512 1"; do
  set -- $cfg
  echo -n "grid $1 split $2: "
  TSDF_INT_SPLIT=$2 timeout 300 python bench.py --grid $1 --steps 20 --warmup 5 --path-only --no-parity --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline'] if d['roofline']['kernel'].startswith('integrate') else d['roofline_other']
print('ms_per_step', d['ms_per_step'], 'integrate avg_launch_ms', r['avg_launch_ms'], 'frac', r['frac'], 'stage integrate', d['stage_ms']['integrate'])"
done | tee $out/int_split_ab.txt
